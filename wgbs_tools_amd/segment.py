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
import os
import sys
import threading
import time

import numpy as np

from .cliutil import add_threads_option, add_where_options, lines_of, require_files
from .genome import (GenomeRefPaths, GenomicRegion, IllegalArgumentError, beta_sanity_check, eprint, write_bed)

DEF_CHUNK = 60000
MAX_BLOCK_SITES = 65535       # WGBSSEG_MAX_CPG (include/wgbsseg.h): longest block; min(max_cpg, chunk_size) above it is refused by the library


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


SMALL_CHUNK_WARNING = ('[wt segment] WARNING: chunk_size is small compared to max_cpg and/or max_bp.\n'
                       '                      It may cause wt segment to fail. It\'s best setting\n'
                       '                      chunk_size > min{max_cpg, max_bp/2}')
STITCH_FAILED = 'Patch stitching Failed'      # segment.py:202-205,229-232: the reference raises IllegalArgumentError with this text


def _as_reference_error(e):
    """A failed stitch surfaces from the library as SegmentorError; the reference's driver raises IllegalArgumentError with the same text."""
    msg = getattr(e, 'msg', str(e))
    return IllegalArgumentError(msg) if STITCH_FAILED in msg else e


class SegmentByChunks:
    def __init__(self, args, betas, engine=None):
        self.betas = betas
        max_cpg = min(args.max_cpg, args.max_bp // 2)            # segment.py:65
        assert max_cpg > 1
        if min(max_cpg, args.chunk_size) > MAX_BLOCK_SITES:
            # include/wgbsseg.h, WGBSSEG_MAX_CPG: above 65,793 sites per block the reference's own float sums stop being exact; windows
            # are stored in 16 bits here.  Said before any upload.
            raise IllegalArgumentError(f'[wt segment] ERROR: blocks of up to min(max_cpg, max_bp/2, chunk_size) = {min(max_cpg, args.chunk_size)} sites '
                                       f'requested; at most {MAX_BLOCK_SITES} are supported (default max_cpg 1000).')
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
        """(tags, starts, ends) of the chunk grid — every region cut every chunk_size sites from its own start, regions never share a
        chunk (segment.py:84-135) — with the reference's warning when a chunk is shorter than the longest block allowed.  The grid the
        library walks is the same one (wgbsseg_first_batch_items / csrc/stitch.h); this list feeds the report and the tests."""
        size = self.args.chunk_size
        if size < self.args.max_cpg:
            eprint(SMALL_CHUNK_WARNING)
        from . import parallel
        regs = self.regions()
        grid = parallel.chunk_grid(regs, size)
        return ['%d-%d' % regs[ri] for ri, _, _ in grid], [a for _, a, _ in grid], [b for _, _, b in grid]

    def run(self):
        tags, starts, ends = self.break_to_chunks()
        from . import parallel
        rank, world, local = parallel.env_rank_world()
        own_engine = self.param_dict['engine'] is None
        csr = None
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
                # an engine that can leave its result as one CSR of absolute borders (the share groups: every whole-genome / -r / -s run and
                # sorted -L files) hands it to the BED writer as it is (wgbsseg_add_loci_borders): no per-region copies, no (start, end) arrays
                as_csr = lambda e: getattr(e, 'segment_regions_csr', None) if all(regs[i][0] >= regs[i - 1][1] for i in range(1, len(regs))) else None
                try:
                    if own_engine and as_csr(eng) and self._slices_pay(eng, regs):
                        # round 6: the regions in slices, the BED rows of one slice written while the next is segmented (and while the beta
                        # bytes of the later ones are still uploading): the 20 ms of BED text leave the run's critical path
                        sliced = self._run_sliced(eng, regs, prof, want_stats)
                        if sliced:
                            return
                    res = (as_csr(eng) or eng.segment_regions)(regs, self.args.chunk_size, self.param_dict)
                except Exception as e:
                    if not (own_engine and getattr(e, 'code', 0) == -7 and 'not resident on any single share' in str(e)):
                        raise _as_reference_error(e)
                    # a junction patch outgrew the halo between two shares (patch doubling past a chunk): one GPU, whole range
                    eprint('[wt segment] a junction patch outgrew the share halo; rerunning on one GPU')
                    eng.close()
                    eng = self.param_dict['engine'] = self.make_engine(starts, ends, gpus=1)
                    res = (as_csr(eng) or eng.segment_regions)(regs, self.args.chunk_size, self.param_dict)
                if as_csr(eng):
                    csr, merged = res, None
                else:
                    merged = dict(zip([f'{a}-{b}' for a, b in regs], res))
                if want_stats:                                   # (before the engine goes away)
                    self.report['engine'] = type(eng).__name__
                    self.report['stitching'] = getattr(eng, 'last_stats', None)
                    tm = eng.timings() if hasattr(eng, 'timings') else None
                    self.report['device'] = tm if isinstance(tm, list) else ([tm] if tm else None)
            else:
                # a chunk engine without the library's driver loop (the CPU engines of the test-suite): the library's chunk grid and
                # stitching tree (wgbsseg_stitch_regions, csrc/stitch.h) around its segment_many, asking for exactly the patches the
                # reference would (no speculation)
                from . import _lib
                regs = self.regions()
                try:
                    res, self.last_stats = _lib.stitch_regions(regs, self.args.chunk_size, lambda sites: eng.segment_many(sites, self.param_dict),
                                                               speculate=False)
                except _lib.SegmentorError as e:
                    raise _as_reference_error(e)
                merged = dict(zip([f'{a}-{b}' for a, b in regs], res))
        finally:
            closer = None
            if own_engine and self.param_dict['engine'] is not None:      # (None: the sliced form has closed it already)
                # releasing ~10 GB of device buffers takes as long as writing the BED: do both at once
                closer = threading.Thread(target=self.param_dict['engine'].close)
                closer.start()
                self.param_dict['engine'] = None
        try:
            if prof: prof.append(('segmentation (device + stitching)', time.perf_counter()))
            if csr is not None:
                self.dump_result_csr(*csr)
            else:
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
        # default: as many GPUs as the job has work for — a share of fewer than four chunks is all launch latency, and a one-chunk
        # -r / -s run should not create contexts, streams and loader threads on every device of a shared node.
        # An explicit --gpus N is taken as given (more shares than GPUs wrap around the devices: only useful for tests).
        n = want or min(have, max(1, -(-len(starts) // 4)))
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

    # Regions in slices (round 6): measured and NOT the default.  hg19 x 32 end to end 0.100-0.115 s with 2 / 4 / 8 slices against 0.105-0.112 in one
    # piece, x 200 0.325-0.373 against 0.314-0.403 with three runs of ~1.0 s among the sliced ones (profiles/r06_e2e_slices_ab.txt): what is exposed
    # behind the last kernel is not the 20 ms of BED text but the engine's close (7-17 GB of device buffers: 17-40 ms), which the one-piece form
    # already runs beside the BED writer.  WGBSSEG_BED_SLICES=n switches the sliced form on.
    SLICES = 1
    SLICE_MIN_SITES = 4000000   # below that a run is over before a second slice could help

    def _slices_pay(self, eng, regs):
        n = int(os.environ.get('WGBSSEG_BED_SLICES', self.SLICES))
        return (n > 1 and hasattr(eng, 'segment_region_slices') and len(regs) >= 2 and sum(b - a for a, b in regs) >= self.SLICE_MIN_SITES)

    def _run_sliced(self, eng, regs, prof, want_stats):
        """run()'s segmentation + dump_result for an engine that can take the regions in slices (multi.GroupEngine): slice k's BED rows
        (wgbsseg_add_loci_borders, append mode) are written by a second thread while slice k + 1 is segmented.  Same rows, same summary line,
        same errors as the one-piece form; False (nothing written) when a junction patch outgrew a share's halo — the caller then reruns on
        one GPU as before."""
        import queue
        import re
        from . import _lib
        names, sizes = self.genome.get_chrom_cpg_sizes()
        cum = np.cumsum(sizes)
        loci = self.genome.loci()
        out_path = self.args.out_path
        to_stdout = out_path is None or out_path is sys.stdout
        if to_stdout:
            sys.stdout.flush()
        todo = queue.Queue()
        state = dict(written=0, dropped=0, error=None, slices=0)

        def writer():
            while True:
                item = todo.get()
                if item is None:
                    return
                if state['error'] is not None:
                    continue                                     # (drain: nothing is written behind a failed slice)
                flat, off = item
                try:
                    w, d = _lib.add_loci_borders(loci, names, cum, flat, off, self.args.min_cpg, None if to_stdout else out_path, append=state['slices'] > 0)
                    state['written'] += w; state['dropped'] += d; state['slices'] += 1
                except _lib.SegmentorError as e:
                    msg = e.msg or ''
                    m = re.match(r'(\[wt add_loci\] line )(\d+)(: .*)', msg, re.S)
                    if m:                                        # the row's number counts the rows of the earlier slices too
                        msg = m.group(1) + str(int(m.group(2)) + state['written']) + m.group(3)
                    state['error'] = msg
        th = threading.Thread(target=writer)
        th.start()
        n_blocks = 0
        try:
            n = int(os.environ.get('WGBSSEG_BED_SLICES', self.SLICES))
            for first, end, flat, off in eng.segment_region_slices(regs, self.args.chunk_size, self.param_dict, n):
                n_blocks += int(np.maximum(np.diff(off) - 1, 0).sum())
                todo.put((flat, off))
                if state['error'] is not None:
                    break
        except Exception as e:
            todo.put(None); th.join()
            if getattr(e, 'code', 0) == -7 and 'not resident on any single share' in str(e):
                if not to_stdout and state['slices'] > 0:
                    os.remove(out_path)                          # (rows of the earlier slices: the rerun writes the file afresh)
                if to_stdout and state['slices'] > 0:
                    raise _as_reference_error(e)                 # (rows already on stdout cannot be taken back)
                return False
            raise _as_reference_error(e)
        if want_stats:                                           # (before the engine goes away)
            self.report['engine'] = type(eng).__name__
            self.report['stitching'] = getattr(eng, 'last_stats', None)
            tm = eng.timings() if hasattr(eng, 'timings') else None
            self.report['device'] = tm if isinstance(tm, list) else ([tm] if tm else None)
        # releasing ~10 GB of device buffers takes as long as writing the last slice: do both at once
        closer = threading.Thread(target=eng.close)
        closer.start()
        self.param_dict['engine'] = None
        if prof: prof.append(('segmentation (device + stitching; BED rows of the earlier slices beside it)', time.perf_counter()))
        todo.put(None)
        th.join()
        closer.join()
        if n_blocks == 0:
            eprint('Empty blocks array')
        elif state['error'] is not None:
            if '[wt add_loci] line' in state['error']:
                # the reference says the counts BEFORE it writes (segment.py:180): a run whose writer fails on a row has still said them
                eprint(f"[wt segment] found {n_blocks - state['dropped']:,} blocks\n             (dropped {state['dropped']:,} short blocks)")
            raise RuntimeError(state['error'])
        else:
            eprint(f"[wt segment] found {state['written']:,} blocks\n             (dropped {state['dropped']:,} short blocks)")
            self.report.update(blocks_found=int(state['written']), blocks_dropped=int(state['dropped']))
        if prof:
            prof.append(('blocks to BED (last slice)', time.perf_counter()))
            if os.environ.get('WGBSSEG_PROFILE'):
                eprint('[wt segment] phases: ' + ', '.join('%s %.3f s' % (n, t - prof[i][1]) for i, (n, t) in enumerate(prof[1:])))
        self.write_stats(prof)
        return True

    def dump_result_csr(self, flat, off):
        """dump_result (segment.py:167-190) straight from the merged border lists (CSR of absolute 1-based borders, regions ascending): the
        blocks are the pairs of consecutive borders (segment.py:154), the `min_cpg` filter and the BED rows happen in the library
        (wgbsseg_add_loci_borders) — same rows, same stderr summary (also when the writer fails on a row).  `flat` / `off` may be the engine's
        `last_csr` views: they are numpy-owned buffers of the binding (not device or library memory) and stay valid after the engine is closed,
        until its next segment_regions_csr call."""
        from . import _lib
        nr_blocks = int(np.maximum(np.diff(off) - 1, 0).sum())
        if nr_blocks == 0:
            eprint('Empty blocks array')
            return
        names, sizes = self.genome.get_chrom_cpg_sizes()
        out_path = self.args.out_path
        to_stdout = out_path is None or out_path is sys.stdout
        if to_stdout:
            sys.stdout.flush()
        def summary(found, short):
            eprint(f'[wt segment] found {found:,} blocks\n'
                   f'             (dropped {short:,} short blocks)')
            if hasattr(self, 'report'):
                self.report.update(blocks_found=int(found), blocks_dropped=int(short))
        try:
            written, dropped = _lib.add_loci_borders(self.genome.loci(), names, np.cumsum(sizes), flat, off, self.args.min_cpg,
                                                     None if to_stdout else out_path)
        except _lib.SegmentorError as e:
            # the reference reports the counts BEFORE it writes (segment.py:180), so a run whose writer fails has still said them: on this
            # path the counts come from one numpy pass over the lists (the library's counts exist only for a completed file)
            # — for a failing ROW only ('[wt add_loci] line N: ...', add_loci.cpp:42-49): an argument the library refused (descending borders, regions
            # out of order, a NULL path) is no list of blocks at all, and counts made of it would mean nothing
            if '[wt add_loci] line' not in (e.msg or ''):
                raise RuntimeError(e.msg)
            flat_, off_ = np.asarray(flat, dtype=np.int64), np.asarray(off, dtype=np.int64)
            d = np.diff(flat_)
            inner = np.ones(d.size, dtype=bool)
            cut = off_[1:-1] - 1                                  # the pair that straddles two regions is not a block
            inner[cut[(cut >= 0) & (cut < d.size)]] = False
            short = int((d[inner] < self.args.min_cpg).sum())
            summary(nr_blocks - short, short)
            raise RuntimeError(e.msg)
        summary(written, dropped)

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
# the tool's own options: (flags, type, default, help).  Flags, defaults and help texts are the reference's (segment.py:261-282) up to
# --device / --gpus / --stats, which exist only here.
_OPTIONS = (
    (('-c', '--chunk_size'), int, DEF_CHUNK, f'Chunk size. Default {DEF_CHUNK} sites'),
    (('-p', '--pcount'), float, 15, 'Pseudo counts of C\'s and T\'s in each block. Default 15'),
    (('--min_cpg',), int, 1, 'Minimal block size (in #sites) to output. Shorter blocks will simply be ommited from output (equivalent to set '
                             'min_cpg to 1 and then filter output by length). Default is 1'),
    (('--max_cpg',), int, 1000, 'Maximal allowed block size (in #sites). Default is 1000'),
    (('--max_bp',), int, 2000, 'Maximal allowed block size (in bp). Default is 2000'),
    (('-o', '--out_path'), None, sys.stdout, 'output path [stdout]'),
    (('--device',), int, 0, 'HIP device index of the (first) GPU [0]'),
    (('--gpus',), int, 0, 'Number of GPUs to spread the chunks over from this one process (the role of -@ in the CPU implementation). '
                          'Default: one GPU for jobs of a few chunks, else all visible GPUs'),
    (('--stats',), None, None, 'Write a JSON report of the run to this path: parameters, regions, chunks, junction patches, GPU batches, blocks '
                               'found / dropped, wall-clock phases and device timings'),
)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description=main.__doc__)
    add_where_options(parser, bed_file=True)
    source = parser.add_mutually_exclusive_group(required=True)
    source.add_argument('--betas', nargs='+')
    source.add_argument('--beta_file', '-F')
    for flags, kind, default, text in _OPTIONS:
        parser.add_argument(*flags, default=default, help=text, **({'type': kind} if kind else {}))
    add_threads_option(parser)
    return parser.parse_args(argv)


def beta_paths_of(args):
    """The input list of `--betas A B ...` or `-F list.txt` (one path per line; blank lines and lines starting with '#' skipped),
    checked before any work starts with the reference's messages (segment.py:285-301, utils_wgbs.py:355-406).  On top of that every
    file must be a uint8 `.beta`: the reference's segmentor silently ignores argv tokens that do not end in ".beta"
    (main.cpp:101-107), which would segment fewer samples than were asked for."""
    paths = list(args.betas) if args.betas else lines_of(args.beta_file)
    if not paths:
        raise IllegalArgumentError(f'no beta files found in file {args.beta_file}')
    if require_files(paths) != '.beta':
        raise IllegalArgumentError(f'segment reads uint8 .beta files; got {paths[0]}')
    return paths


parse_betas_input = beta_paths_of        # the reference's name for it (segment.py:285)


def main(argv=None):
    """
    Segment the genome, or a subset region, to homogenously methylated blocks.
    Input: one or more beta files to segment
    Output: blocks file (BED format + startCpG, endCpG columns)
    """
    args = parse_args(argv)
    betas = beta_paths_of(args)
    SegmentByChunks(args, betas).run()


if __name__ == '__main__':
    main()
