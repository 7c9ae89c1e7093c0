"""Parity at the north star's sizes (BASELINE.json configs[2..4]) on the GPU, through the C ABI.

  x32   hg19-shaped 28,217,448 CpGs x 32 betas, whole genome: a spread sample of 256 full-size chunks against the
        reference binary (oracle/_ref/segmentor), one whole chromosome chunk by chunk AND patch by patch against it,
        and the STITCHED border lists of four chromosomes against the reference's pairwise tree (segment.py:157-165,
        199-252; restated in tests/reftree.py and pinned there by vectors captured from the reference's
        driver) walked over the same chunk / patch DPs.
  x200  the same genome x 200 betas (7 LDS sample groups in the scoring kernel), north_star's named target, IN FULL: all 483
        chunks and every junction patch against the reference binary, all 25 stitched chromosomes against the tree
        (tests/fullref.py; hosts with fewer than 64 CPUs: 64 chunks, one chromosome's patches, three trees).
  x512  one piece of configs[4] (max_cpg 5000, max_bp 1e6, chunk 50000, 512 betas): eight 50,000-site chunks on the
        GPU; one FULL chunk against the many-thread oracle restatement (1.3e11 evaluations), four reduced chunks
        against the reference binary, stitched result against the tree.

Bit-exact everywhere.  The genome-wide properties: borders strictly ascending, first/last = the chromosome's ends,
every block <= max_cpg sites and <= max_bp base pairs (segmentor.cpp:111-117).
"""
import json
import os
import os.path as op

import ctypes as C
import numpy as np
import pytest

import fullref
from oracle import oracle
from wgbs_tools_amd import _lib, synth, segment as S

pytestmark = pytest.mark.gpu
SEED = 20260926
SITES = synth.HG19_NR_SITES


def _device_genome(n_sites, n_samples):
    import torch
    dev = torch.device('cuda', 0)
    pitch = ((2 * n_sites + 255) // 256) * 256 + 256
    buf = torch.empty((n_samples, pitch), dtype=torch.uint8, device=dev)
    rc = _lib.load_synth().wgbssynth_fill_betas(C.c_void_p(buf.data_ptr()), pitch, n_sites, 0, n_samples, SEED, None)
    assert rc == 0
    torch.cuda.synchronize()
    return buf, pitch


# the reference runner, the recording chunk engine, the reference's grid and tree: tests/fullref.py
_ref_on_ranges = fullref.ref_on_ranges
_Recorder = fullref.Recorder
_tree = fullref.tree


def _check_properties(res, regions, loci, max_cpg, max_bp):
    lo = loci.astype(np.int64)
    for b, (a, e) in zip(res, regions):
        b = np.asarray(b, dtype=np.int64)
        assert b[0] == a and b[-1] == e
        d = np.diff(b)
        assert (d > 0).all(), 'borders not strictly ascending'
        assert d.max() <= max_cpg, 'a block has more than max_cpg sites'
        bp = lo[b[1:] - 2] - lo[b[:-1] - 1]                 # last site of the block minus its first site
        assert bp.max() <= max_bp, 'a block spans more than max_bp'


_grid = fullref.grid


def _run_whole(seg, regions, chunk, pcount, max_cpg, max_bp):
    st = np.array([r[0] for r in regions], dtype=np.int64)
    en = np.array([r[1] for r in regions], dtype=np.int64)
    res, stats = seg.segment_regions(st, en, chunk, pcount, max_cpg, max_bp)
    return res, stats


def _stitched_vs_tree(seg, res, regions, which, chunk, pcount, max_cpg, max_bp):
    """stitched result of the regions `which` == the reference's tree over the per-chunk DPs; returns the recorders."""
    recs = []
    for ri in which:
        a, e = regions[ri]
        eng = _Recorder(seg, pcount, max_cpg, max_bp)
        chunks = eng.segment_many(_grid(a, e, chunk), {})
        n_chunks = len(chunks)
        want = _tree(chunks, eng)
        assert np.array_equal(np.asarray(res[ri], dtype=np.int64), want), 'stitched borders of region %d differ from the reference tree' % ri
        recs.append((eng, n_chunks))
    return recs


@pytest.fixture(scope='module')
def hg19():
    names, sizes = synth.genome_shape(SITES, 25)
    sizes = [int(s) for s in sizes]
    loci = synth.synth_loci(SEED, sizes)
    regions, pos = [], 1
    for s in sizes:
        regions.append((pos, pos + s))
        pos += s
    return dict(sizes=sizes, loci=loci, regions=regions)


def _spread_chunks(sizes, chunk, count):
    grid, pos = [], 0
    for sz in sizes:
        grid += list(range(pos, pos + sz - chunk + 1, chunk))
        pos += sz
    pick = sorted(set(grid[i] for i in np.linspace(0, len(grid) - 1, min(len(grid), count)).astype(int)))
    return pick


def test_hg19_x32_whole_genome(hg19):
    """BASELINE.json configs[2]."""
    N, chunk, pc, mc, mb = 32, 60000, 15.0, 1000, 2000
    buf, pitch = _device_genome(SITES, N)
    loci, regions, sizes = hg19['loci'], hg19['regions'], hg19['sizes']
    with _lib.Segmenter(0) as seg:
        seg.set_betas_device(buf.data_ptr(), N, pitch, SITES, keepalive=buf)
        seg.set_loci(loci)
        res, stats = _run_whole(seg, regions, chunk, pc, mc, mb)
        assert stats['chunks'] == 483
        _check_properties(res, regions, loci, mc, mb)
        # (a) 256 full-size chunks spread over the genome against the reference binary
        pick = _spread_chunks(sizes, chunk, 256)
        ref = _ref_on_ranges(buf, loci, [(st, chunk) for st in pick], pc, mc, mb)
        got = seg.segment_chunks(pick, [chunk] * len(pick), pc, mc, mb)
        for st, g in zip(pick, got):
            assert np.array_equal(g.astype(np.int64), ref[(st, chunk)]), 'chunk at site %d differs from the reference binary' % st
        # (b) stitched chromosomes against the reference's pairwise tree: chr1 (41 chunks), chr21, chrX, chrM (one short chunk)
        which = [0, 20, 22, 24]
        recs = _stitched_vs_tree(seg, res, regions, which, chunk, pc, mc, mb)
        # (c) chr21 end to end against the reference binary: every chunk and every patch the tree asked for
        eng, _ = recs[1]
        ranges = [(a - 1, b - a) for (a, b) in eng.asked]
        ref21 = _ref_on_ranges(buf, loci, ranges, pc, mc, mb)
        for (a, b), r in eng.asked.items():
            assert np.array_equal(r - a, ref21[(a - 1, b - a)]), 'chr21 range %s differs from the reference binary' % ((a, b),)
        # and the patches of chr1's 40 junctions
        eng1, n1 = recs[0]
        patches = [(a, b) for (a, b) in eng1.asked if b - a < chunk and (a, b) not in _grid(*regions[0], chunk)]
        assert len(patches) >= n1 - 1
        refp = _ref_on_ranges(buf, loci, [(a - 1, b - a) for a, b in patches], pc, mc, mb)
        for a, b in patches:
            assert np.array_equal(eng1.asked[(a, b)] - a, refp[(a - 1, b - a)])
    del buf


def _x200_mode():
    """'full': every chunk, every junction patch, every chromosome against the reference binary (north_star's named target; two
    minutes on the GPU box's 256 CPUs); 'sample': 64 chunks, the patches of one chromosome, three trees — what a host with few
    cores can afford.  WGBSSEG_X200_CHECK = full | sample overrides the choice by core count."""
    mode = os.environ.get('WGBSSEG_X200_CHECK', '')
    if mode not in ('full', 'sample'):
        mode = 'full' if (os.cpu_count() or 1) >= 64 else 'sample'
    return mode


def test_hg19_x200_atlas_scale(hg19):
    """BASELINE.json configs[3] (one GPU's view: the whole genome fits a single MI355X) = north_star's "bit-exact block boundaries
    at 28M CpGs x 200 betas": ALL 483 chunks of the reference's grid and EVERY junction patch its 25 pairwise trees ask for through
    the reference binary (segmentor.cpp:60-159), ALL 25 stitched chromosomes against the reference's tree (segment.py:157-165,
    199-252) over those DPs.  The counts go to gpurun_out/x200_full_vs_reference.json."""
    import torch
    N, chunk, pc, mc, mb = 200, 60000, 15.0, 1000, 2000
    torch.cuda.empty_cache()
    buf, pitch = _device_genome(SITES, N)
    loci, regions, sizes = hg19['loci'], hg19['regions'], hg19['sizes']
    with _lib.Segmenter(0) as seg:
        seg.set_betas_device(buf.data_ptr(), N, pitch, SITES, keepalive=buf)
        seg.set_loci(loci)
        res, stats = _run_whole(seg, regions, chunk, pc, mc, mb)
        assert stats['chunks'] == 483
        _check_properties(res, regions, loci, mc, mb)
        mode = _x200_mode()
        if mode == 'full':
            out = fullref.whole_genome_vs_reference(seg, buf, loci, regions, chunk, pc, mc, mb, res, log=print)
            out.update(mode=mode, sites=SITES, samples=N, blocks=int(sum(len(r) - 1 for r in res)))
            os.makedirs('gpurun_out', exist_ok=True)
            with open(op.join('gpurun_out', 'x200_full_vs_reference.json'), 'w') as f:
                json.dump(out, f, indent=1)
            print('x200 whole genome vs the reference binary: %s' % json.dumps(out))
            assert out['differences'] == 0, out['different']
            assert (out['chunks_identical'], out['chunks']) == (483, 483)
            assert (out['chromosomes_identical'], out['chromosomes']) == (25, 25)
            assert out['patches_identical'] == out['patches'] >= 483 - 25          # one junction between neighbouring chunks, at least
        else:
            pick = _spread_chunks(sizes, chunk, 64)
            ref = _ref_on_ranges(buf, loci, [(st, chunk) for st in pick], pc, mc, mb)
            got = seg.segment_chunks(pick, [chunk] * len(pick), pc, mc, mb)
            for st, g in zip(pick, got):
                assert np.array_equal(g.astype(np.int64), ref[(st, chunk)]), 'chunk at site %d differs from the reference binary' % st
            recs = _stitched_vs_tree(seg, res, regions, [1, 21, 24], chunk, pc, mc, mb)
            # the patches of chr22 against the reference binary
            eng, _ = recs[1]
            grid = set(_grid(*regions[21], chunk))
            patches = [(a, b) for (a, b) in eng.asked if (a, b) not in grid]
            refp = _ref_on_ranges(buf, loci, [(a - 1, b - a) for a, b in patches], pc, mc, mb)
            for a, b in patches:
                assert np.array_equal(eng.asked[(a, b)] - a, refp[(a - 1, b - a)])
        # sharded == unsharded: the 8-piece split of the chunk grid (one piece per GPU of a node), every piece on its own
        # context holding only its share of the beta bytes, stitched on the host
        from wgbs_tools_amd import multi
        sharded = multi.segment_regions_on_shares(buf.data_ptr(), N, pitch, SITES, loci, regions, chunk, pc, mc, mb,
                                                  devices=[0] * 8)
        for r1, r8 in zip(res, sharded):
            assert np.array_equal(np.asarray(r1), np.asarray(r8)), 'the 8-share run differs from the single-context run'
    del buf
    torch.cuda.empty_cache()


def test_x512_deep_share():
    """A piece of BASELINE.json configs[4]: 512 betas, max_cpg 5000, max_bp 1e6, chunk_size 50000 (deep-DP stress)."""
    import torch
    N, chunk, pc, max_bp = 512, 50000, 15.0, 1000000
    mc = min(5000, max_bp // 2)                              # segment.py:65
    n_sites = 8 * chunk
    torch.cuda.empty_cache()
    buf, pitch = _device_genome(n_sites, N)
    loci = synth.synth_loci(SEED, [n_sites])
    regions = [(1, n_sites + 1)]
    with _lib.Segmenter(0) as seg:
        seg.set_betas_device(buf.data_ptr(), N, pitch, n_sites, keepalive=buf)
        seg.set_loci(loci)
        res, stats = _run_whole(seg, regions, chunk, pc, mc, max_bp)
        assert stats['chunks'] == 8
        _check_properties(res, regions, loci, mc, max_bp)
        recs = _stitched_vs_tree(seg, res, regions, [0], chunk, pc, mc, max_bp)
        eng, _ = recs[0]
        # one FULL 50,000-site chunk (1.3e11 evaluations) against the many-thread restatement of the reference
        st = 3 * chunk
        host = buf[:, 2 * st:2 * (st + chunk)].cpu().numpy()
        want = oracle.segment_chunk_mt([host[s].reshape(-1, 2) for s in range(N)], loci[st:st + chunk], pc, mc, max_bp)
        assert np.array_equal(eng.asked[(st + 1, st + chunk + 1)] - (st + 1), want.astype(np.int64)), \
            'full deep chunk differs from the oracle restatement'
        # four reduced chunks and the junction patches against the reference binary itself
        small = [(0, 2000), (123457, 2000), (250001, 1800), (n_sites - 1500, 1500)]
        grid = set(_grid(1, n_sites + 1, chunk))
        patches = [(a - 1, b - a) for (a, b) in eng.asked if (a, b) not in grid]
        ref = _ref_on_ranges(buf, loci, small + patches, pc, mc, max_bp)
        got = seg.segment_chunks([s for s, _ in small], [n for _, n in small], pc, mc, max_bp)
        for (s0, n), g in zip(small, got):
            assert np.array_equal(g.astype(np.int64), ref[(s0, n)]), 'reduced deep chunk at %d differs from the reference binary' % s0
        for (s0, n) in patches:
            assert np.array_equal(eng.asked[(s0 + 1, s0 + 1 + n)] - (s0 + 1), ref[(s0, n)])
    del buf
    torch.cuda.empty_cache()


def test_x512_deep_full_genome(hg19):
    """BASELINE.json configs[4] at FULL size on one GPU: 28,217,448 CpGs x 512 betas (28.9 GB resident), max_cpg 5000, max_bp 1e6,
    chunk_size 50000 — 576 chunks, every window 5000 sites wide: 1.4e11 scored blocks x 512 samples = 7.2e13 evaluations through
    the staged scored-block buffer (WGBSSEG_COST_BUDGET_MB), the wide scoring tiles, the <15,32> recurrence and its L2 ring.
    Checked: the genome-wide properties; three 6,000-site ranges spread over the genome (or WGBSSEG_DEEP_ORACLE_CHUNKS full 50,000-site
    chunks: 1.3e11 evaluations and two minutes on all host threads each; 8 of them in profiles/r03_deep_full_genome.log) against the oracle's many-thread restatement; the stitched trees of chr21 and chr22
    against the reference's pairwise tree walked over the same DPs.  The run's timing goes to gpurun_out/deep_full_timing.json."""
    import json
    import time
    import torch
    N, chunk, pc, max_bp = 512, 50000, 15.0, 1000000
    mc = min(5000, max_bp // 2)                              # segment.py:65
    torch.cuda.empty_cache()
    buf, pitch = _device_genome(SITES, N)
    loci, regions, sizes = hg19['loci'], hg19['regions'], hg19['sizes']
    with _lib.Segmenter(0) as seg:
        seg.set_betas_device(buf.data_ptr(), N, pitch, SITES, keepalive=buf)
        seg.set_loci(loci)
        t0 = time.perf_counter()
        res, stats = _run_whole(seg, regions, chunk, pc, mc, max_bp)
        wall = time.perf_counter() - t0
        tm = seg.timings()
        n_chunks = sum(len(_grid(a, e, chunk)) for a, e in regions)
        assert stats['chunks'] == n_chunks == 576
        _check_properties(res, regions, loci, mc, max_bp)
        rec = {'workload': 'hg19-shaped %d CpGs x %d betas, max_cpg %d, max_bp %d, chunk_size %d, pcount %g (BASELINE.json configs[4], whole genome, ONE MI355X)' % (SITES, N, mc, max_bp, chunk, pc),
               'wall_s': wall, 'value_CpG_sites_per_s': SITES / wall, 'chunks': int(stats['chunks']), 'stitch_stats': {k: int(v) for k, v in stats.items()},
               'device_ms': {k: tm[k] for k in ('scan_ms', 'window_ms', 'cost_ms', 'dp_ms', 'trace_ms', 'total_ms')},
               'evals': int(tm['evals']), 'pairs': int(tm['pairs']), 'stages': int(tm['n_stages']), 'max_window': int(tm['max_window']),
               'evals_per_s_in_k_cost': tm['evals'] / (tm['cost_ms'] * 1e-3),
               # the recurrence's row traffic: every scored block (8 B) is read once by the workers of k_dp<15,32> and pushed
               'dp_row_bytes': int(tm['pairs']) * 8, 'dp_row_GB_per_s': tm['pairs'] * 8 / (tm['dp_ms'] * 1e-3) / 1e9,
               'dp_row_frac_of_hbm_peak': tm['pairs'] * 8 / (tm['dp_ms'] * 1e-3) / 1e9 / 8000.0,
               'blocks': int(sum(len(r) - 1 for r in res))}
        os.makedirs('gpurun_out', exist_ok=True)
        with open(op.join('gpurun_out', 'deep_full_timing.json'), 'w') as f:
            json.dump(rec, f, indent=1)
        print('deep full genome: %.1f s, %s' % (wall, json.dumps(rec['device_ms'])))
        # ranges of the resident genome against the oracle restatement.  In the suite: WGBSSEG_DEEP_ORACLE_SITES (default 6,000) sites at
        # three places of the genome (3 x 9e9 evaluations on the host: the suite has to fit the driver's window, and
        # test_x512_deep_share compares a FULL 50,000-site chunk of the same configuration); WGBSSEG_DEEP_ORACLE_CHUNKS = k: k full
        # chunks spread over the genome instead (two minutes of all host threads each: profiles/r03_deep_full_genome.log has 8)
        k = int(os.environ.get('WGBSSEG_DEEP_ORACLE_CHUNKS', '0'))
        part = int(os.environ.get('WGBSSEG_DEEP_ORACLE_SITES', '6000'))
        pick = [(st, chunk) for st in _spread_chunks(sizes, chunk, k)] if k > 0 else [(st + 777, part) for st in _spread_chunks(sizes, chunk, 3)]
        got = seg.segment_chunks([st for st, _ in pick], [ln for _, ln in pick], pc, mc, max_bp)
        for (st, ln), g in zip(pick, got):
            host = buf[:, 2 * st:2 * (st + ln)].cpu().numpy()
            want = oracle.segment_chunk_mt([host[s].reshape(-1, 2) for s in range(N)], loci[st:st + ln], pc, mc, max_bp)
            assert np.array_equal(g.astype(np.int64), want.astype(np.int64)), 'deep range [%d, +%d) differs from the oracle restatement' % (st, ln)
            del host
        # the stitched trees of chr21 and chr22 (9 and 10 chunks) against the reference's pairwise tree over the same DPs
        _stitched_vs_tree(seg, res, regions, [20, 21], chunk, pc, mc, max_bp)
    del buf
    torch.cuda.empty_cache()
