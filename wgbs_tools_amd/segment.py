#!/usr/bin/python3 -u
"""`wgbstools segment` on MI355X: the driver side of the hot path.

Mirrors the interface of the reference's src/python/segment.py (same flags, defaults, stderr text, exceptions,
block-BED output) so that it is a drop-in for that command, but replaces its process fan-out
(`Pool.starmap(segment_process)` launching one `tabix | cut | segmentor` pipeline per chunk, segment.py:41-59,
144-146) by ONE batched call through the C ABI into the HIP kernels (wgbs_tools_amd/_lib.py ->
csrc/libwgbsseg.so).  Chunk grid (segment.py:124-135) and junction stitching (segment.py:199-252) are kept
exactly, because the running double sums of the DP make the chunk grid part of the bit-exact answer.

There is no CPU fallback: without the HIP library and a gfx950 device `segment` fails.
"""
import argparse
import multiprocessing
import os
import sys
import threading
import time

import numpy as np

from .genome import (GenomeRefPaths, GenomicRegion, IllegalArgumentError, beta_sanity_check, eprint, write_bed)

DEF_CHUNK = 60000
MAX_CPG_CAP = 8000            # WGBSSEG_MAX_CPG (include/wgbsseg.h): the one limit the reference does not have


# ------------------------------------------------------------------------------------------------------------
# chunk engine: (start, end) 1-based half-open CpG ranges -> absolute border arrays
# ------------------------------------------------------------------------------------------------------------
class HipEngine:
    """Resident betas + loci on one GPU; `segment_many` is the batched form of segment_process (segment.py:41-59)."""

    def __init__(self, betas, genome, device=0, site_range=None):
        from . import _lib                     # raises NativeLibraryError if libwgbsseg.so is not built
        self._seg = _lib.Segmenter(device)
        nr = genome.get_nr_sites()
        lo, hi = (0, nr) if site_range is None else (max(0, site_range[0]), min(nr, site_range[1]))
        self.base = lo                                           # 0-based site index of the first resident site
        maps = [np.memmap(b, dtype=np.uint8, mode='r') for b in betas]
        self._seg.set_betas([m[2 * lo:2 * hi] for m in maps])
        self._seg.set_loci(genome.loci()[lo:hi])
        self._seg.set_site_base(lo)                              # error messages name absolute sites

    def segment_many(self, sites_list, params):
        """sites_list: [(start, end), ...] 1-based half-open; returns [np.int64 array of absolute borders, ...]."""
        out = [None] * len(sites_list)
        idx, st0, ln = [], [], []
        for i, (start, end) in enumerate(sites_list):
            assert end - start > 0, f'trying to segment an empty interval {(start, end)}'
            if end - start == 1:                                 # segment.py:45-46
                out[i] = np.array([start, end])
            else:
                idx.append(i)
                st0.append(start - 1 - self.base)
                ln.append(end - start)
        if idx:
            try:
                res = self._seg.segment_chunks(st0, ln, params['pcount'], params['max_cpg'], params['max_bp'])
            except Exception as e:
                eprint(f'Failed in sites {sites_list[idx[0]]} .. {sites_list[idx[-1]]}')   # segment.py:57-59
                raise e
            for i, r in zip(idx, res):
                out[i] = r.astype(np.int64) + sites_list[i][0]
        return out

    def segment_csr(self, starts, ends, params, off=None, out=None):
        """Array form of segment_many: 1-based half-open ranges -> (offsets int64 [n + 1], int32 borders RELATIVE to each start),
        written into `off` / `out` when given (a slot of parallel.NodeSlots)."""
        st0 = np.asarray(starts, dtype=np.int64) - 1 - self.base
        ln = (np.asarray(ends, dtype=np.int64) - np.asarray(starts, dtype=np.int64)).astype(np.int32)
        flat, off = self._seg.segment_chunks_csr(st0, ln, params['pcount'], params['max_cpg'], params['max_bp'], out=out, off=off)
        return off, flat

    def segment_regions(self, regions, chunk_size, params):
        """The whole of SegmentByChunks.run's chunk fan-out + merge_df_list, native (wgbsseg_segment_regions):
        regions = [(startCpG, endCpG), ...] 1-based half-open; returns the merged absolute border list of each."""
        st = np.array([r[0] for r in regions], dtype=np.int64) - self.base
        en = np.array([r[1] for r in regions], dtype=np.int64) - self.base
        try:
            res, self.last_stats = self._seg.segment_regions(st, en, chunk_size, params['pcount'], params['max_cpg'],
                                                             params['max_bp'])
        except Exception as e:
            eprint(f'Failed in sites {regions[0]} .. {regions[-1]}')
            raise e
        return [r + self.base for r in res]

    def timings(self):
        return self._seg.timings()

    def close(self):
        self._seg.close()


class GatherEngine:
    """Junction patches anywhere in the genome on one GPU without the genome being resident there: the sites of the
    requested ranges are gathered from the memory-mapped beta files into one compact buffer (a chunk DP reads nothing
    outside its own range: segmentor.cpp:164-177 seeks to -s and reads -n sites).  Used by rank 0 of a multi-process run for
    the patches between the ranks' chunk results: a few hundred sites per junction."""

    def __init__(self, betas, genome, device=0):
        from . import _lib
        self._seg = _lib.Segmenter(device)
        self._maps = [np.memmap(b, dtype=np.uint8, mode='r') for b in betas]
        self._loci = genome.loci()

    def segment_many(self, sites_list, params):
        out = [None] * len(sites_list)
        idx = [i for i, (a, b) in enumerate(sites_list) if b - a > 1]
        for i, (a, b) in enumerate(sites_list):
            assert b - a > 0, f'trying to segment an empty interval {(a, b)}'
            if b - a == 1:                                       # segment.py:45-46
                out[i] = np.array([a, b])
        if idx:
            lens = np.array([sites_list[i][1] - sites_list[i][0] for i in idx], dtype=np.int64)
            offs = np.concatenate([[0], np.cumsum((lens + 7) // 8 * 8)])      # ranges start on 16-byte boundaries
            total = int(offs[-1])
            buf = np.zeros((len(self._maps), 2 * total), dtype=np.uint8)
            loci = np.zeros(total, dtype=np.uint32)
            for k, i in enumerate(idx):
                a, b = sites_list[i]
                o = int(offs[k])
                for s, m in enumerate(self._maps):
                    buf[s, 2 * o:2 * (o + b - a)] = m[2 * (a - 1):2 * (b - 1)]
                loci[o:o + b - a] = self._loci[a - 1:b - 1]
            self._seg.set_betas(list(buf))
            self._seg.set_loci(loci)
            res = self._seg.segment_chunks(offs[:-1], lens, params['pcount'], params['max_cpg'], params['max_bp'])
            for i, r in zip(idx, res):
                out[i] = r.astype(np.int64) + sites_list[i][0]
        return out

    def segment_csr(self, starts, ends, params, off=None, out=None):
        res = self.segment_many(list(zip(np.asarray(starts).tolist(), np.asarray(ends).tolist())), params)
        n = len(res)
        off = np.empty(n + 1, dtype=np.int64) if off is None else off
        off[0] = 0
        np.cumsum([len(r) for r in res], out=off[1:n + 1])
        out = np.empty(max(1, int(off[n])), dtype=np.int32) if out is None else out
        for i, r in enumerate(res):
            out[off[i]:off[i + 1]] = r - int(starts[i])
        return off, out

    def close(self):
        self._seg.close()


def segment_process(params):
    """segment.py:41-59 for a single chunk (kept for interface parity; the driver itself batches)."""
    return params['engine'].segment_many([params['sites']], params)[0]


# ------------------------------------------------------------------------------------------------------------
# stitching (segment.py:199-252), restated with numpy set operations instead of pandas
# ------------------------------------------------------------------------------------------------------------
def find_dups(b1, b2):
    """segment.py:239-240: mask over concatenate([b1, b2]) of the values that occur more than once."""
    cat = np.concatenate([b1, b2])
    _, inv, cnt = np.unique(cat, return_inverse=True, return_counts=True)
    return cnt[inv] > 1


def is_2_overlap(b1, b2):
    return np.sum(find_dups(b1, b2))


def merge2(b1, b2):
    """segment.py:243-246"""
    nr_from_df1 = np.argmax(find_dups(b1, b2))
    skip_from_df2 = np.searchsorted(b2, b1[nr_from_df1])
    return np.concatenate([b1[:nr_from_df1 + 1], b2[skip_from_df2 + 1:]]).copy()


def increase_patch(pre_size, maxval):
    """segment.py:249-252"""
    if pre_size == maxval:
        return maxval + 1
    return int(min(pre_size * 2, maxval))


STITCH_FAIL_MSG = '[wt segment] Patch stitching Failed! ' \
                  '             Try increasing chunk size (--chunk_size flag)'


class _Stitch:
    """State machine of one stitch_2_dfs call (segment.py:199-232): `want()` names the patch it needs next,
    `feed(patch)` consumes it; lets many junctions share one GPU batch per attempt."""

    def __init__(self, b1, b2):
        if b1[-1] != b2[0]:
            msg = '[wt segment] Patch stitching Failed! ' \
                  '             patches are not supposed to be merged'
            raise IllegalArgumentError(msg)
        self.b1, self.b2 = b1, b2
        self.n1 = b1[-1] - b1[0]
        self.n2 = b2[-1] - b2[0]
        self.p1 = min(50, self.n1)
        self.p2 = min(50, self.n2)
        self.result = None

    def want(self):
        if self.result is not None:
            return None
        if not (self.p1 <= self.n1 and self.p2 <= self.n2):
            raise IllegalArgumentError(STITCH_FAIL_MSG)
        return (int(self.b1[-1] - self.p1), int(self.b1[-1] + self.p2))

    def feed(self, patch):
        o1 = is_2_overlap(self.b1, patch)
        o2 = is_2_overlap(patch, self.b2)
        if o1 and o2:
            self.result = merge2(merge2(self.b1, patch), self.b2)
        else:
            if not o1:
                self.p1 = increase_patch(self.p1, self.n1)
            if not o2:
                self.p2 = increase_patch(self.p2, self.n2)


def stitch_2_dfs(b1, b2, params):
    """segment.py:199-232 (one junction; patches come from params['engine'])."""
    st = _Stitch(b1, b2)
    while st.result is None:
        sites = st.want()
        st.feed(params['engine'].segment_many([sites], params)[0])
    return st.result


def stitch_round(pairs, params, cache):
    """All junctions of one pairwise-reduce round (segment.py:159-161) together: every attempt's patches go to
    the GPU as one batch.  The patch DP is a pure function of (start, end), so results are cached."""
    states = [_Stitch(b1, b2) for b1, b2 in pairs]
    while True:
        need = {}
        for st in states:
            w = st.want()
            if w is not None and w not in cache:
                need[w] = None
        if need:
            keys = list(need)
            for k, r in zip(keys, params['engine'].segment_many(keys, params)):
                cache[k] = r
        pending = False
        for st in states:
            w = st.want()
            if w is not None:
                st.feed(cache[w])
                pending = pending or st.result is None
        if not pending:
            return [st.result for st in states]


class SegmentByChunks:
    def __init__(self, args, betas, engine=None):
        self.betas = betas
        max_cpg = min(args.max_cpg, args.max_bp // 2)            # segment.py:65
        assert max_cpg > 1
        if max_cpg > MAX_CPG_CAP:
            # the one limit the reference does not have (include/wgbsseg.h, WGBSSEG_MAX_CPG): say so here, before any upload
            raise IllegalArgumentError(f'[wt segment] ERROR: blocks of up to min(max_cpg, max_bp/2) = {max_cpg} sites requested; '
                                       f'this implementation supports at most {MAX_CPG_CAP} (default 1000).')
        self._regions = None
        self.genome = GenomeRefPaths(args.genome)
        self.param_dict = {'betas': betas,
                           'pcount': args.pcount,
                           'max_cpg': max_cpg,
                           'max_bp': args.max_bp,
                           'genome': self.genome,
                           'engine': engine}
        self.args = args
        self.validate_genome()

    def validate_genome(self):                                   # segment.py:78-82
        for beta in self.betas:
            if not beta_sanity_check(beta, self.genome):
                msg = f'[wt segment] ERROR: current genome reference ({self.genome.genome}) does not match the input beta file ({beta}).'
                raise IllegalArgumentError(msg)

    def regions(self):
        """The (startCpG, endCpG) rows break_to_chunks iterates over (segment.py:95-122), read once.  Rows with
        endCpG == startCpG yield no chunk in the reference (`range(start, end, step) + [end]` has a single element):
        they are dropped here."""
        if self._regions is None:
            self._regions = [(s, e) for s, e in self._read_regions() if e > s]
        return self._regions

    def _read_regions(self):
        if self.args.bed_file:
            df = load_blocks_file(self.args.bed_file)
            is_nice, msg = is_block_file_nice(df)
            if not is_nice:
                msg = '[wt segment] ERROR: invalid bed file.\n' \
                      f'                    {msg}\n' \
                      f'                    Try: sort -k1,1 -k2,2n {self.args.bed_file} | ' \
                      'bedtools merge -i - | wgbstools convert --drop_empty -p -L -'
                eprint(msg)
                raise IllegalArgumentError('Invalid bed file')
            if df.shape[0] > 2 * 1e4:
                msg = '[wt segment] WARNING: bed file contains many regions.\n' \
                      '                      Segmentation will take a long time.\n' \
                      '                      Consider running w/o -L flag and intersect the results\n'
                eprint(msg)
            return [(int(s), int(e)) for s, e in df]
        gr = GenomicRegion(self.args, genome=self.genome)
        if gr.is_whole():
            _, sizes = self.genome.get_chrom_cpg_sizes()
            ends = np.cumsum(sizes) + 1
            starts = ends - sizes
            return [(int(s), int(e)) for s, e in zip(starts, ends)]
        return [tuple(int(x) for x in gr.sites)]

    def break_to_chunks(self):
        """ Break range of sites to chunks of size 'step',
            while keeping chromosomes separated  (segment.py:84-135)"""
        step = self.args.chunk_size
        if step < self.args.max_cpg:
            msg = '[wt segment] WARNING: chunk_size is small compared to max_cpg and/or max_bp.\n' \
                  '                      It may cause wt segment to fail. It\'s best setting\n' \
                  '                      chunk_size > min{max_cpg, max_bp/2}'
            eprint(msg)
        tags, starts, ends = [], [], []
        for start, end in self.regions():
            bords = list(range(start, end, step)) + [end]
            tags += [f'{start}-{end}'] * (len(bords) - 1)
            starts += bords[:-1]
            ends += bords[1:]
        return tags, starts, ends

    def run(self):
        tags, starts, ends = self.break_to_chunks()
        from . import parallel
        rank, world, local = parallel.env_rank_world()
        own_engine = self.param_dict['engine'] is None
        if world > 1:
            return self.run_sharded(rank, world, local)
        want_stats = bool(getattr(self.args, 'stats', None))
        prof = [('start', time.perf_counter())] if (os.environ.get('WGBSSEG_PROFILE') or want_stats) else None
        self.report = {'regions': len(self.regions()), 'chunks': len(starts), 'sites': int(sum(e - s for s, e in zip(starts, ends)))}
        if not starts:                                           # nothing to segment (e.g. an -L file of empty rows)
            self.dump_result(np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
            self.write_stats(prof)
            return
        if own_engine:
            self.param_dict['engine'] = self.make_engine(starts, ends)
            if prof: prof.append(('engine (one GPU: loci + betas to the device)', time.perf_counter()))
        try:
            eng = self.param_dict['engine']
            if hasattr(eng, 'segment_regions'):
                # native chunk grid + batched patches + stitching around the GPU batches (csrc/stitch.h)
                regs = self.regions()
                try:
                    res = eng.segment_regions(regs, self.args.chunk_size, self.param_dict)
                except Exception as e:
                    if not (own_engine and getattr(e, 'code', 0) == -7 and 'not resident on any single share' in str(e)):
                        raise
                    # a junction patch outgrew the halo between two shares (patch doubling past a chunk): one GPU, whole range
                    eprint('[wt segment] a junction patch outgrew the share halo; rerunning on one GPU')
                    eng.close()
                    eng = self.param_dict['engine'] = self.make_engine(starts, ends, gpus=1)
                    res = eng.segment_regions(regs, self.args.chunk_size, self.param_dict)
                merged = dict(zip([f'{a}-{b}' for a, b in regs], res))
                if want_stats:                                   # (before the engine goes away)
                    self.report['engine'] = type(eng).__name__
                    self.report['stitching'] = getattr(eng, 'last_stats', None)
                    tm = eng.timings() if hasattr(eng, 'timings') else None
                    self.report['device'] = tm if isinstance(tm, list) else ([tm] if tm else None)
            else:
                arr = eng.segment_many(list(zip(starts, ends)), self.param_dict)
                # merge chunks from the same "tag" group (segment.py:148-154); all groups advance round by round together
                groups = {}
                for i, t in enumerate(tags):
                    groups.setdefault(t, []).append(arr[i])
                merged = self.merge_groups(groups)
        finally:
            closer = None
            if own_engine:
                # releasing ~10 GB of device buffers takes as long as writing the BED: do both at once
                closer = threading.Thread(target=self.param_dict['engine'].close)
                closer.start()
                self.param_dict['engine'] = None
        try:
            if prof: prof.append(('segmentation (device + stitching)', time.perf_counter()))
            s = np.concatenate([m[:-1] for m in merged.values()]) if merged else np.zeros(0, dtype=np.int64)
            e = np.concatenate([m[1:] for m in merged.values()]) if merged else np.zeros(0, dtype=np.int64)
            self.dump_result(s, e)
        finally:
            if closer is not None:
                closer.join()
        if prof:
            prof.append(('blocks to BED', time.perf_counter()))
            if os.environ.get('WGBSSEG_PROFILE'):
                eprint('[wt segment] phases: ' + ', '.join('%s %.3f s' % (n, t - prof[i][1]) for i, (n, t) in enumerate(prof[1:])))
        self.write_stats(prof)

    def write_stats(self, prof):
        """--stats PATH: one JSON object about the run (the reference only has its stderr lines)."""
        path = getattr(self.args, 'stats', None)
        if not path:
            return
        import json
        from . import wgbs_tools
        a = self.args
        rep = {'tool': 'wgbstools segment', 'version': wgbs_tools.VERSION,
               'parameters': {'betas': len(self.betas), 'genome': self.genome.genome, 'chunk_size': a.chunk_size, 'pcount': a.pcount,
                              'min_cpg': a.min_cpg, 'max_cpg': self.param_dict['max_cpg'], 'max_bp': a.max_bp,
                              'region': a.region, 'sites': a.sites, 'bed_file': a.bed_file},
               'out_path': None if a.out_path is sys.stdout else str(a.out_path)}
        rep.update(getattr(self, 'report', {}))
        if prof and len(prof) > 1:
            rep['phases_s'] = {n: round(t - prof[i][1], 6) for i, (n, t) in enumerate(prof[1:])}
            rep['wall_s'] = round(prof[-1][1] - prof[0][1], 6)
        with open(path, 'w') as f:
            json.dump(rep, f, indent=1, default=lambda o: int(o) if isinstance(o, np.integer) else float(o) if isinstance(o, np.floating) else str(o))
            f.write('\n')

    def make_engine(self, starts, ends, gpus=None):
        """--gpus N (default: every visible GPU): a region list a share group can plan over (ascending, disjoint: every
        whole-genome / -r / -s run and sorted -L files) -> one share per GPU (wgbs_tools_amd/multi.py), each holding only its
        window; else one GPU holding the site range the run needs."""
        from . import _lib, multi
        want = getattr(self.args, 'gpus', 0) if gpus is None else gpus
        have = max(1, _lib.device_count())
        n = want or have            # more shares than GPUs is allowed (they wrap around the devices): only useful for tests
        first = getattr(self.args, 'device', 0)
        if multi.regions_fit_a_group(self.regions()):
            # one share per GPU (also for a single GPU: the share group streams the upload and segments what has arrived)
            return multi.GroupEngine(self.betas, self.genome, [(first + d) % have for d in range(n)])
        return HipEngine(self.betas, self.genome, device=first, site_range=(min(starts) - 1, max(ends) - 1))

    def run_sharded(self, rank, world, local, engine_factory=None, patch_engine_factory=None):
        """One process per GPU (python -m torch.distributed.run ... wgbstools segment ...): the chunk grid is cut into
        `world` contiguous, work-balanced runs of chunks (the planner of the share groups, include/wgbsseg.h); every rank
        uploads its own window of the beta files and runs its chunk DPs on its own GPU; the per-chunk border lists are
        gathered on the host of rank 0 (no collective on the data path), which walks the reference's pairwise tree
        (segment.py:157-165) over ALL chunks with the native stitcher — the junction patches, a few hundred sites each,
        run on rank 0's GPU from a gathered buffer — and writes the BED.  The result is the one-GPU result whatever the
        number of ranks.  The other ranks produce no output."""
        from . import parallel, _lib
        regs = self.regions()
        pd = self.param_dict
        dist = parallel.init_host_group()
        if engine_factory is None:
            def engine_factory(site_range):
                ndev = max(1, _lib.device_count())
                return HipEngine(self.betas, self.genome, device=local % ndev, site_range=site_range)
        if patch_engine_factory is None:
            def patch_engine_factory():
                return GatherEngine(self.betas, self.genome, device=local % max(1, _lib.device_count()))
        if not regs:
            if rank == 0:
                self.dump_result(np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
            dist.barrier()
            return
        run = parallel.ShardedRun(dist, regs, self.args.chunk_size, self.genome.loci(), pd, rank, world)
        eng, peng = None, []
        try:
            if run.my_starts.size:
                eng = engine_factory(run.window())

            def patches(starts, ends):                           # the few patches the rehearsal still misses: rank 0's GPU
                if not peng:
                    peng.append(patch_engine_factory())
                return parallel.csr_engine(peng[0], pd)(starts, ends)
            merged = run.step(parallel.csr_engine(eng, pd), patches)
            self.last_stats = run.last_stats
            if rank == 0:
                s_ = np.concatenate([m[:-1] for m in merged])
                e_ = np.concatenate([m[1:] for m in merged])
                self.dump_result(s_, e_)
        finally:
            for e in [eng] + peng:
                if e is not None and hasattr(e, 'close'):
                    e.close()
            self.param_dict['engine'] = None
            dist.barrier()
            run.close()

    def merge_groups(self, groups):
        """merge_df_list (segment.py:157-165) for every tag at once: the pairing order inside a tag is the
        reference's ((0,1),(2,3),.. then again on the merged list); rounds of different tags share GPU batches."""
        lists = {t: list(v) for t, v in groups.items()}
        cache = {}
        while any(len(v) > 1 for v in lists.values()):
            pairs, owner = [], []
            for t, dflist in lists.items():
                if len(dflist) > 1:
                    for i in range(1, len(dflist), 2):
                        pairs.append((dflist[i - 1], dflist[i]))
                        owner.append(t)
            res = stitch_round(pairs, self.param_dict, cache)
            new = {t: [] for t in lists}
            for t, r in zip(owner, res):
                new[t].append(r)
            for t, dflist in lists.items():
                if len(dflist) > 1:
                    last = [dflist[-1]] if len(dflist) % 2 else []
                    lists[t] = new[t] + last
        return {t: v[0] for t, v in lists.items()}

    def merge_df_list(self, dflist, pool=None):
        """segment.py:157-165 for one tag."""
        return self.merge_groups({'x': dflist})['x']

    def dump_result(self, start_cpg, end_cpg):
        """segment.py:167-190"""
        if start_cpg.size == 0:
            eprint('Empty blocks array')
            return
        nr_blocks = start_cpg.size
        s, e = start_cpg, end_cpg
        if not (s[1:] >= s[:-1]).all():                          # (chromosome by chromosome they already come sorted)
            order = np.argsort(s, kind='stable')
            s, e = s[order], e[order]
        keep = (e - s) > self.args.min_cpg - 1
        if not keep.all():
            s, e = s[keep], e[keep]
        nr_blocks_filt = s.size
        nr_dropped = nr_blocks - nr_blocks_filt
        eprint(f'[wt segment] found {nr_blocks_filt:,} blocks\n'
               f'             (dropped {nr_dropped:,} short blocks)')
        if hasattr(self, 'report'):
            self.report.update(blocks_found=int(nr_blocks_filt), blocks_dropped=int(nr_dropped))
        write_bed(self.genome, s, e, self.args.out_path)


# ------------------------------------------------------------------------------------------------------------
# -L blocks file (beta_to_blocks.py:50-91 load_blocks_file, segment.py:25-38 is_block_file_nice)
# ------------------------------------------------------------------------------------------------------------
def load_blocks_file(blocks_path):
    """(startCpG, endCpG) int rows of a 5-column blocks BED (header line and '#' comments tolerated, rows with
    missing CpG columns dropped: segment.py:96 `.dropna()`)."""
    import gzip
    if not os.path.isfile(blocks_path):
        raise IllegalArgumentError(f'No such file: {blocks_path}')
    # the library's one-pass parser when the table is plain and complete (no missing CpG fields: their spellings are this
    # function's business); anything else line by line below
    from .beta_to_blocks import _load_blocks_native
    t = _load_blocks_native(blocks_path, None)
    if t is not None and not t.na.any():
        if (t.endCpG < t.startCpG).any():
            raise IllegalArgumentError(f'Invalid CpG columns in blocks file {blocks_path}')
        return np.stack([t.startCpG, t.endCpG], axis=1)
    opener = gzip.open if blocks_path.endswith('.gz') else open
    rows = []
    first = True
    with opener(blocks_path, 'rt') as f:
        for line in f:
            if not line.strip() or line.startswith('#'):
                continue
            tok = line.rstrip('\n').split('\t')
            if first:
                first = False
                if len(tok) < 5:
                    msg = f'Invalid blocks file: {blocks_path}. less than 5 columns.\n'
                    msg += f'Run wgbstools convert -L {blocks_path} -o OUTPUT_REGION_FILE to add the CpG columns'
                    raise IllegalArgumentError(msg)
                if not tok[1].isdigit():
                    continue                                     # header row
            if len(tok) < 5 or tok[3] in ('', 'NA') or tok[4] in ('', 'NA'):
                continue
            s, e = int(tok[3]), int(tok[4])
            if e - s < 0:
                raise IllegalArgumentError(f'Invalid CpG columns in blocks file {blocks_path}')
            rows.append((s, e))
    return np.array(rows, dtype=np.int64).reshape(-1, 2)


def is_block_file_nice(df):
    """segment.py:25-38"""
    if df.shape[0] != np.unique(df, axis=0).shape[0]:
        return False, 'Some blocks are duplicated'
    sdf = df[np.argsort(df[:, 0], kind='stable')]
    if not (sdf[1:, 0] - sdf[:-1, 1] >= 0).all():
        return False, 'Some blocks overlap'
    return True, ''


# ------------------------------------------------------------------------------------------------------------
# Main (segment.py:261-313)
# ------------------------------------------------------------------------------------------------------------
def add_GR_args(parser, required=False, bed_file=False):
    """utils_wgbs.py:233-247"""
    region_or_sites = parser.add_mutually_exclusive_group(required=required)
    region_or_sites.add_argument('-s', '--sites', help='a CpG index range, of the form: "450000-450050"')
    region_or_sites.add_argument('-r', '--region', help='genomic region of the form "chr1:10,000-10,500"')
    region_or_sites.add_argument('--array_id', help='Illumina array id, e.g. cg00001755')
    if bed_file:
        region_or_sites.add_argument('-L', '--bed_file', help='Bed file. Columns <chr, start, end>. '
                                     'For some features columns 4-5 should be <startCpG, endCpG> (run wgbstools convert -L BED_PATH)')
    parser.add_argument('--genome', help='Genome reference name. Default is "default".', default='default')
    return region_or_sites


def add_multi_thread_args(parser):
    """utils_wgbs.py:250-260 (kept for command-line compatibility; the GPU path does not fork workers)."""
    try:
        cpu_env = 'SLURM_JOB_CPUS_PER_NODE'
        if cpu_env in os.environ.keys():
            def_cpus = int(os.environ[cpu_env])
        else:
            def_cpus = multiprocessing.cpu_count()
    except Exception:
        def_cpus = 8
    parser.add_argument('-@', '--threads', type=int, default=def_cpus,
                        help='Number of threads to use (default: all available CPUs)')


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description=main.__doc__)
    add_GR_args(parser, bed_file=True)
    betas_or_file = parser.add_mutually_exclusive_group(required=True)
    betas_or_file.add_argument('--betas', nargs='+')
    betas_or_file.add_argument('--beta_file', '-F')
    parser.add_argument('-c', '--chunk_size', type=int, default=DEF_CHUNK,
                        help=f'Chunk size. Default {DEF_CHUNK} sites')
    parser.add_argument('-p', '--pcount', type=float, default=15,
                        help='Pseudo counts of C\'s and T\'s in each block. Default 15')
    parser.add_argument('--min_cpg', type=int, default=1,
                        help='Minimal block size (in #sites) to output. Shorter blocks will simply be '
                             'ommited from output (equivalent to set min_cpg to 1 and then filter output by '
                             'length). Default is 1')
    parser.add_argument('--max_cpg', type=int, default=1000,
                        help=f'Maximal allowed block size (in #sites). Default is 1000 (at most {MAX_CPG_CAP} here)')
    parser.add_argument('--max_bp', type=int, default=2000,
                        help='Maximal allowed block size (in bp). Default is 2000')
    parser.add_argument('-o', '--out_path', default=sys.stdout,
                        help='output path [stdout]')
    add_multi_thread_args(parser)
    parser.add_argument('--device', type=int, default=0, help='HIP device index of the (first) GPU [0]')
    parser.add_argument('--gpus', type=int, default=0,
                        help='Number of GPUs to spread the chunks over from this one process (the role of -@ in the '
                             'CPU implementation). Default: all visible GPUs')
    parser.add_argument('--stats', metavar='JSON_PATH',
                        help='Write a JSON report of the run: parameters, regions, chunks, junction patches, GPU batches, '
                             'blocks found / dropped, wall-clock phases and device timings')
    return parser.parse_args(argv)


def validate_single_file(fpath, suff=None):
    """utils_wgbs.py:383-406"""
    if fpath is None:
        raise IllegalArgumentError("Input file is None")
    if not os.path.isfile(fpath):
        raise IllegalArgumentError(f'No such file: {fpath}')
    if suff is not None and not fpath.endswith(suff):
        raise IllegalArgumentError(f'file {fpath} must end with {suff}')
    return fpath


def validate_file_list(files, min_len=1):
    """utils_wgbs.py:355-380"""
    if len(files) < min_len:
        raise IllegalArgumentError(f'Input error: at least {min_len} input files must be given')
    first = files[0]
    if len(first) == 1:
        raise IllegalArgumentError(f'Input is not a list of files: {files}')
    suff = os.path.splitext(first)[1]
    for fpath in files:
        validate_single_file(fpath, suff)
    if suff != '.beta':
        # the reference's segmentor silently ignores argv tokens that do not end in ".beta" (main.cpp:101-107)
        raise IllegalArgumentError(f'segment reads uint8 .beta files; got {first}')


def parse_betas_input(args):
    """segment.py:285-301"""
    if args.betas:
        betas = args.betas
    elif args.beta_file:
        validate_single_file(args.beta_file)
        with open(args.beta_file, 'r') as f:
            betas = [b.strip() for b in f.readlines() if b.strip() and not b.startswith('#')]
        if not betas:
            raise IllegalArgumentError(f'no beta files found in file {args.beta_file}')
    validate_file_list(betas)
    return betas


def main(argv=None):
    """
    Segment the genome, or a subset region, to homogenously methylated blocks.
    Input: one or more beta files to segment
    Output: blocks file (BED format + startCpG, endCpG columns)
    """
    args = parse_args(argv)
    betas = parse_betas_input(args)
    SegmentByChunks(args, betas).run()


if __name__ == '__main__':
    main()
