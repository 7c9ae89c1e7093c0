"""End-to-end on the GPU: `wgbstools segment` (CLI -> driver -> C ABI -> HIP kernels -> native stitching -> BED)
against the golden vectors captured from the reference's own driver (tests/golden/driver_cases.json)."""
import contextlib
import hashlib
import io
import json
import os.path as op

import numpy as np
import pytest

from wgbs_tools_amd import synth, wgbs_tools, segment as S, genome as G

pytestmark = pytest.mark.gpu
ROOT = op.dirname(op.dirname(op.abspath(__file__)))
CASES = ['wg_c20000', 'wg_c60000_min3', 'sites_3chunks', 'sites_single', 'wg_pcount0', 'small_chunks', 'tiny_chunks',
         'wide_bp', 'bed_regions']


@pytest.fixture(scope='module')
def driver_golden():
    with open(op.join(ROOT, 'tests', 'golden', 'driver_cases.json')) as f:
        return json.load(f)


@pytest.fixture(scope='module')
def world(driver_golden, tmp_path_factory):
    meta = driver_golden['meta']
    names = [c for c, _ in meta['chrom_sizes']]
    sizes = [s for _, s in meta['chrom_sizes']]
    loci = synth.synth_loci(meta['seed'], sizes)
    total = int(sum(sizes))
    d = tmp_path_factory.mktemp('world')
    refdir = synth.write_genome(str(d / 'references' / 'synth'), names, sizes, loci)
    paths = []
    for i in range(meta['n_betas']):
        p = str(d / ('s%d.beta' % i))
        synth.write_beta(p, synth.synth_betas(meta['seed'], i, 0, total))
        paths.append(p)
    return dict(refdir=refdir, paths=paths, loci=loci, names=names, sizes=sizes)


@pytest.mark.parametrize('name', CASES)
def test_cli_matches_reference_driver(name, driver_golden, world, tmp_path):
    g = driver_golden['cases'][name]
    out = str(tmp_path / 'blocks.bed')
    argv = ['wgbstools', 'segment', '--betas'] + world['paths'] + ['--genome', world['refdir'], '-o', out]
    for k, v in g['args'].items():
        flag = {'chunk_size': '-c', 'min_cpg': '--min_cpg', 'sites': '-s', 'pcount': '-p', 'max_cpg': '--max_cpg', 'max_bp': '--max_bp'}[k]
        argv += [flag, str(v)]
    if g['bed_rows'] is not None:
        bed = str(tmp_path / 'regions.bed')
        with open(bed, 'w') as f:
            for s, e in g['bed_rows']:
                f.write('chrN\t0\t1\t%d\t%d\n' % (s, e))
        argv += ['-L', bed]
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        rc = wgbs_tools.main(argv)
    assert rc == 0, err.getvalue()
    want = g['stderr']
    k = want.find('[wt segment] found')
    warn = want[:k]                                   # the golden capture ran break_to_chunks twice: warning x2
    assert err.getvalue() == warn[:len(warn) // 2] + want[k:]
    rows = [l.rstrip('\n').split('\t') for l in open(out)]
    table = np.array([[int(r[3]), int(r[4])] for r in rows], dtype=np.int64).reshape(-1, 2)
    assert table.shape[0] == g['n_blocks']
    assert hashlib.sha1(table.tobytes()).hexdigest() == g['table_sha1']
    loci = world['loci'].astype(np.int64)
    for r, (s, e) in list(zip(rows, table))[:: max(1, len(rows) // 300)]:
        assert int(r[1]) == loci[s - 1] and int(r[2]) == loci[e - 2] + 1


@pytest.mark.parametrize('name,gpus', [('wg_c20000', 3), ('wg_pcount0', 2), ('sites_3chunks', 4), ('bed_regions', 3), ('tiny_chunks', 5),
                                       ('small_chunks', 8)])
def test_cli_over_a_share_group_matches_reference_driver(name, gpus, driver_golden, world, tmp_path):
    """`wgbstools segment --gpus N`: one process, N shares (on this 1-GPU box they wrap around device 0), every share
    holding only its own window of the beta files, one host-side tree: the reference driver's blocks, whatever N."""
    g = driver_golden['cases'][name]
    out = str(tmp_path / 'blocks.bed')
    argv = ['wgbstools', 'segment', '--betas'] + world['paths'] + ['--genome', world['refdir'], '-o', out, '--gpus', str(gpus)]
    for k, v in g['args'].items():
        flag = {'chunk_size': '-c', 'min_cpg': '--min_cpg', 'sites': '-s', 'pcount': '-p', 'max_cpg': '--max_cpg', 'max_bp': '--max_bp'}[k]
        argv += [flag, str(v)]
    if g['bed_rows'] is not None:
        bed = str(tmp_path / 'regions.bed')
        with open(bed, 'w') as f:
            for s, e in g['bed_rows']:
                f.write('chrN\t0\t1\t%d\t%d\n' % (s, e))
        argv += ['-L', bed]
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        rc = wgbs_tools.main(argv)
    assert rc == 0, err.getvalue()
    rows = [l.rstrip('\n').split('\t') for l in open(out)]
    table = np.array([[int(r[3]), int(r[4])] for r in rows], dtype=np.int64).reshape(-1, 2)
    assert table.shape[0] == g['n_blocks']
    assert hashlib.sha1(table.tobytes()).hexdigest() == g['table_sha1']


@pytest.mark.parametrize('name,gpus,slices', [('wg_c20000', 1, 2), ('wg_c60000_min3', 1, 3), ('wg_pcount0', 2, 4), ('bed_regions', 3, 2), ('tiny_chunks', 1, 8)])
def test_cli_in_region_slices_writes_the_same_file(name, gpus, slices, driver_golden, world, tmp_path, monkeypatch):
    """Round 6: the regions in slices, slice k's BED rows written (append mode, a second thread) while slice k + 1 is segmented — the file, the
    summary line and the --stats counts are those of the one-piece run and of the reference driver."""
    g = driver_golden['cases'][name]
    monkeypatch.setattr(S.SegmentByChunks, 'SLICE_MIN_SITES', 1)
    argv0 = ['wgbstools', 'segment', '--betas'] + world['paths'] + ['--genome', world['refdir'], '--gpus', str(gpus)]
    for k, v in g['args'].items():
        flag = {'chunk_size': '-c', 'min_cpg': '--min_cpg', 'sites': '-s', 'pcount': '-p', 'max_cpg': '--max_cpg', 'max_bp': '--max_bp'}[k]
        argv0 += [flag, str(v)]
    if g['bed_rows'] is not None:
        bed = str(tmp_path / 'regions.bed')
        with open(bed, 'w') as f:
            for s, e in g['bed_rows']:
                f.write('chrN\t0\t1\t%d\t%d\n' % (s, e))
        argv0 += ['-L', bed]
    outs, errs = [], []
    for n in (1, slices):
        monkeypatch.setenv('WGBSSEG_BED_SLICES', str(n))
        out = str(tmp_path / ('blocks_%d.bed' % n))
        stats = str(tmp_path / ('stats_%d.json' % n))
        err = io.StringIO()
        with contextlib.redirect_stderr(err):
            rc = wgbs_tools.main(argv0 + ['-o', out, '--stats', stats])
        assert rc == 0, err.getvalue()
        outs.append(open(out, 'rb').read()); errs.append(err.getvalue())
        rep = json.load(open(stats))
        assert rep['blocks_found'] == g['n_blocks']
        if n > 1 and len(world['sizes']) >= 2 and g['bed_rows'] is None and 'sites' not in g['args']:
            assert rep['stitching']['slices'] >= 2                  # (a whole-genome run of several chromosomes really went through the sliced form)
    assert outs[0] == outs[1] and errs[0] == errs[1]
    rows = [l.split('\t') for l in outs[1].decode().splitlines()]
    table = np.array([[int(r[3]), int(r[4])] for r in rows], dtype=np.int64).reshape(-1, 2)
    assert table.shape[0] == g['n_blocks'] and hashlib.sha1(table.tobytes()).hexdigest() == g['table_sha1']


def test_share_group_plan_windows_and_errors(world):
    """The group API directly: windows cover the shares, a share reports #meth > #cov with the ABSOLUTE site, a share
    without data is refused, the group result equals the one-context result."""
    from wgbs_tools_amd import _lib
    sizes, loci = world['sizes'], world['loci']
    total = int(sum(sizes))
    regions, pos = [], 1
    for sz in sizes:
        regions.append((pos, pos + sz))
        pos += sz
    maps = [np.fromfile(p, dtype=np.uint8) for p in world['paths']]
    with _lib.Segmenter(0) as seg:
        seg.set_betas([m.reshape(-1, 2) for m in maps])
        seg.set_loci(loci)
        st = np.array([r[0] for r in regions]); en = np.array([r[1] for r in regions])
        want, _ = seg.segment_regions(st, en, 9000, 15.0, 1000, 2000)
    with _lib.SegmenterGroup([0, 0, 0, 0]) as grp:
        w = grp.plan(loci, regions, 9000, 15.0, 1000, 2000)
        assert w['chunks'].sum() == sum(-(-sz // 9000) for sz in sizes) and (w['chunks'] > 0).all()
        with pytest.raises(_lib.SegmentorError, match='no beta data'):
            grp.segment_regions()
        grp.load_host(maps)
        got, stats = grp.segment_regions()
        assert stats['chunks'] == w['chunks'].sum()
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        # a bad site inside the third share
        bad_site = int(w['win_lo'][2]) + 9000 + 77
        maps2 = [m.copy() for m in maps]
        maps2[1][2 * bad_site] = maps2[1][2 * bad_site + 1] + 1
        grp.load_host(maps2)
        with pytest.raises(_lib.SegmentorError, match=r'sample 1 .*site %d \(0-based\)' % bad_site):
            grp.segment_regions()
        # a halo too small for the junction patches between shares is refused, not silently wrong
        w = grp.plan(loci, regions, 9000, 15.0, 1000, 2000, halo=10)
        sh = _lib.plan_shares(loci, regions, 9000, 15.0, 1000, 2000, 4, halo=10)
        assert np.array_equal(sh['win_lo'], w['win_lo']) and np.array_equal(sh['win_hi'], w['win_hi'])
        ends = set(r[1] - 1 for r in regions)
        uncovered = False
        for d in range(3):                                   # first-attempt patch of the junction between shares d and d+1
            j0 = int(sh['own_hi'][d])
            if j0 in ends:
                continue                                     # a chromosome boundary: no junction
            uncovered = uncovered or not any(sh['win_lo'][q] <= j0 - 50 and j0 + 50 <= sh['win_hi'][q] for q in (d, d + 1))
        grp.load_host(maps)
        if uncovered:
            with pytest.raises(_lib.SegmentorError, match='not resident on any single share'):
                grp.segment_regions()
        else:
            got2, _ = grp.segment_regions()
            assert all(np.array_equal(a, b) for a, b in zip(got2, want))


def test_stitcher_around_segment_many_equals_the_native_driver_loop(driver_golden, world, tmp_path):
    """An engine that only offers segment_many goes through wgbsseg_stitch_regions (the library's tree around a caller's chunk engine,
    no speculation: the reference's own patch requests); one that offers segment_regions through wgbsseg_segment_regions (batched,
    speculative).  Same BED over the HIP chunk engine."""
    import argparse
    g = driver_golden['cases']['wg_c20000']
    args = argparse.Namespace(sites=None, region=None, array_id=None, bed_file=None, genome=world['refdir'], betas=world['paths'],
                              beta_file=None, chunk_size=20000, pcount=15, min_cpg=1, max_cpg=1000, max_bp=2000,
                              out_path=str(tmp_path / 'a.bed'), threads=1, device=0, gpus=1)
    gen = G.GenomeRefPaths(world['refdir'])

    class PyOnly:                                   # hides segment_regions -> wgbsseg_stitch_regions around segment_many
        def __init__(self, eng):
            self.eng = eng

        def segment_many(self, sites, params):
            return self.eng.segment_many(sites, params)
    eng = S.HipEngine(world['paths'], gen)
    try:
        with contextlib.redirect_stderr(io.StringIO()):
            S.SegmentByChunks(args, world['paths'], engine=PyOnly(eng)).run()
            args.out_path = str(tmp_path / 'b.bed')
            S.SegmentByChunks(args, world['paths'], engine=eng).run()
    finally:
        eng.close()
    a, b = open(str(tmp_path / 'a.bed')).read(), open(str(tmp_path / 'b.bed')).read()
    assert a == b and a.count('\n') == g['n_blocks']


def test_two_ranks_one_process_per_gpu_cli(driver_golden, world, tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 wgbstools segment ...`: the sharded product path (two ranks;
    on a 1-GPU box both map to device 0) must write the same BED as the reference driver."""
    import subprocess
    import sys
    g = driver_golden['cases']['wg_c20000']
    out = str(tmp_path / 'sharded.bed')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', op.join(ROOT, 'wgbstools'), 'segment', '--betas'] + world['paths'] + \
          ['--genome', world['refdir'], '-c', '20000', '-o', out]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert res.returncode == 0, res.stdout.decode()[-3000:]
    rows = [l.rstrip('\n').split('\t') for l in open(out)]
    table = np.array([[int(r[3]), int(r[4])] for r in rows], dtype=np.int64).reshape(-1, 2)
    assert table.shape[0] == g['n_blocks']
    assert hashlib.sha1(table.tobytes()).hexdigest() == g['table_sha1']


def test_two_contexts_on_two_host_threads_share_the_stitching_pool():
    """The junction stitching uses one process-wide pool of host threads; a second caller does its own work itself.  Two contexts
    segmenting different genomes at the same time from two Python threads (ctypes releases the GIL) must each get the answer they
    get alone."""
    import threading
    from wgbs_tools_amd import _lib
    worlds = []
    for seed, n_chr in ((5, 6), (6, 9)):
        sizes = [int(x) for x in np.random.default_rng(seed).integers(30000, 90000, n_chr)]
        loci = synth.synth_loci(seed, sizes)
        total = int(sum(sizes))
        betas = [synth.synth_betas(seed, i, 0, total) for i in range(4)]
        cum = np.concatenate([[0], np.cumsum(sizes)])
        worlds.append((loci, betas, cum[:-1] + 1, cum[1:] + 1))
    segs = []
    for loci, betas, st, en in worlds:
        sg = _lib.Segmenter(0)
        sg.set_betas(betas)
        sg.set_loci(loci)
        segs.append(sg)
    try:
        alone = [sg.segment_regions(st, en, 7000, 15.0, 1000, 2000)[0] for sg, (_, _, st, en) in zip(segs, worlds)]
        for _ in range(5):
            got = [None, None]

            def work(i):
                got[i] = segs[i].segment_regions(worlds[i][2], worlds[i][3], 7000, 15.0, 1000, 2000)[0]
            ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            for i in range(2):
                assert len(got[i]) == len(alone[i])
                for a, b in zip(got[i], alone[i]):
                    assert np.array_equal(a, b)
    finally:
        for sg in segs:
            sg.close()


@pytest.mark.parametrize('gpus', [1, 3])
def test_cli_stats_report(gpus, driver_golden, world, tmp_path):
    """--stats: the JSON report of a run on the device (engine, chunks, junction patches, batches, blocks, phases, device timings of
    every share)."""
    g = driver_golden['cases']['wg_c20000']
    out, rep_path = str(tmp_path / 'blocks.bed'), str(tmp_path / 'run.json')
    argv = ['wgbstools', 'segment', '--betas'] + world['paths'] + ['--genome', world['refdir'], '-o', out, '-c', str(g['args']['chunk_size']),
                                                                     '--gpus', str(gpus), '--stats', rep_path]
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        assert wgbs_tools.main(argv) == 0, err.getvalue()
    rep = json.load(open(rep_path))
    assert rep['engine'] in ('HipEngine', 'GroupEngine') and rep['blocks_found'] == g['n_blocks'] == sum(1 for _ in open(out))
    assert rep['chunks'] == len(g['chunks']['starts']) == rep['stitching']['chunks'] and rep['stitching']['batches'] >= 1
    assert len(rep['device']) in (1, gpus) and all(d['cost_ms'] >= 0 and d['sites'] >= 0 for d in rep['device'])
    assert sum(d['evals'] for d in rep['device']) > 0 and rep['wall_s'] > 0
