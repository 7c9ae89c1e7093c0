"""Several GPUs from one process: the `-@` of the reference (segment.py:144-146, a Pool of chunk processes) as a group of
GPU shares (include/wgbsseg.h, wgbsseg_group_*).

The chunk grid (segment.py:124-135) is cut into contiguous runs of chunks, one per share, balanced by the work the chunks
hold (scored blocks, counted from the loci); a share uploads only its own window of every beta file; the chunk DPs and
junction patches of a batch run on all shares side by side (one host thread each, no device-to-device traffic); the
reference's pairwise stitching tree (segment.py:157-165,199-252) runs once on the host over all shares' results, so
the borders do not depend on the number of shares.
"""
import numpy as np

from . import _lib


class GroupEngine:
    """Chunk engine of SegmentByChunks over a group of shares; same `segment_regions` interface as HipEngine."""

    def __init__(self, betas, genome, devices):
        self.betas = list(betas)
        self.genome = genome
        self.devices = list(devices)
        self.group = _lib.SegmenterGroup(self.devices)
        self._maps = None
        self.base = 0
        self.last_stats = None
        self.windows = None

    def _run(self, regions, chunk_size, params, copy):
        loci = self.genome.loci()
        self.windows = self.group.plan(loci, regions, chunk_size, params['pcount'], params['max_cpg'], params['max_bp'])
        if self._maps is None:
            self._maps = [np.memmap(b, dtype=np.uint8, mode='r') for b in self.betas]
        self.group.load_host(self._maps, wait=False)        # the shares segment what has arrived while the rest is uploading
        res, self.last_stats = self.group.segment_regions(copy=copy)
        return res

    def segment_regions(self, regions, chunk_size, params):
        """regions: [(startCpG, endCpG)] 1-based half-open, ascending and disjoint (whole chromosomes, a -s/-r range or the
        rows of a sorted -L file).  -> merged absolute border list of each region."""
        return self._run(regions, chunk_size, params, True)

    def segment_regions_csr(self, regions, chunk_size, params):
        """The same call, the result left as ONE CSR: (flat int32 absolute 1-based borders, off int64 [regions + 1]) — views into the
        group's result buffer, valid until its next call; what wgbsseg_add_loci_borders prints the BED from without any (start, end)
        arrays in between."""
        self._run(regions, chunk_size, params, False)
        return self.group.last_csr

    def segment_region_slices(self, regions, chunk_size, params, n_slices=4):
        """Generator over slices of the regions, cut where the cumulative sites pass k / n_slices of the total: yields (first, end, flat, off)
        — regions [first, end) as one CSR of absolute borders in a buffer of its own — as soon as the slice is segmented, while the beta bytes
        of the later slices are still uploading (the upload streams site-major) and before the next slice is computed: the caller's BED
        writer works on slice k during slice k + 1 (round 6; regions never interact: segment.py:84-86,129-134)."""
        loci = self.genome.loci()
        self.windows = self.group.plan(loci, regions, chunk_size, params['pcount'], params['max_cpg'], params['max_bp'])
        if self._maps is None:
            self._maps = [np.memmap(b, dtype=np.uint8, mode='r') for b in self.betas]
        self.group.load_host(self._maps, wait=False)
        sizes = np.array([b - a for a, b in regions], dtype=np.int64)
        cum = np.cumsum(sizes)
        cuts = sorted(set([0, len(regions)] + [int(np.searchsorted(cum, cum[-1] * k / n_slices, side='left')) + 1 for k in range(1, n_slices)]))
        cuts = [c for c in cuts if 0 <= c <= len(regions)]
        stats = []
        for first, end in zip(cuts[:-1], cuts[1:]):
            flat, off, st = self.group.segment_region_range(first, end, int(sizes[first:end].sum()) + (end - first))
            stats.append(st)
            yield first, end, flat, off
        self.last_stats = {k: (sum(s[k] for s in stats)) for k in stats[0]} if stats else None
        if self.last_stats:
            self.last_stats['slices'] = len(stats)

    def timings(self):
        return [self.group.timings(d) for d in range(self.group.n_shares)]

    def close(self):
        self.group.close()
        self._maps = None


def regions_fit_a_group(regions):
    """A group plans over ascending, disjoint regions (every whole-genome / -r / -s run, and sorted -L files)."""
    return all(b > a for a, b in regions) and all(regions[i][0] >= regions[i - 1][1] for i in range(1, len(regions)))


def segment_regions_on_shares(data_ptr, n_samples, pitch, n_sites, loci, regions, chunk_size, pcount, max_cpg, max_bp, devices,
                              keepalive=None, group=None, want_group=False):
    """The group path over ONE whole-genome device buffer [n_samples][pitch] that every share can see (all `devices` equal
    to the buffer's device: tests and the one-GPU bench): share d borrows the view of its window.
    -> list of border arrays per region (and the group, stats when want_group)."""
    own = group is None
    g = group or _lib.SegmenterGroup(devices)
    try:
        if own or g.n_regions == 0:
            w = g.plan(loci, regions, chunk_size, pcount, max_cpg, max_bp)
            for d in range(g.n_shares):
                g.share_set_device(d, int(data_ptr) + 2 * int(w['win_lo'][d]), n_samples, pitch, keepalive=keepalive)
        res, stats = g.segment_regions()
        if want_group:
            return res, stats, g
        return res
    finally:
        if own and not want_group:
            g.close()
