"""Pin the oracle: our C restatement (oracle/segment_oracle.c) must reproduce, bit for bit, the border lists the
reference's own `segmentor` printed on the same seeded inputs (tests/golden/chunk_cases.json, captured by
tests/golden/make_golden.py from oracle/_ref/segmentor).  When the reference binary is present (build container
and, as a prebuilt file, the GPU box) it is also re-run live."""
import numpy as np
import pytest

import cases
from oracle import oracle

SMALL = [k for k in cases.CHUNK_CASES]


@pytest.mark.parametrize('name', SMALL)
def test_restatement_matches_reference_golden(name, golden_chunks):
    g = golden_chunks[name]
    spec = g['spec']
    slices, loci = cases.build_case(spec)
    assert cases.case_checksum(slices, loci) == g['input_crc32'], 'synthetic input generator drifted'
    b = oracle.segment_chunk(slices, loci, spec['pcount'], spec['max_cpg'], spec['max_bp'])
    assert b.tolist() == g['borders']
    # format contract of print_borders (segmentor.cpp:30-34): ascending, starts at 0, ends at n
    assert b[0] == 0 and b[-1] == spec['n'] and (np.diff(b) > 0).all()


def test_chr21_multichunk_matches_reference_golden(golden_chunks):
    g = golden_chunks['chr21']
    spec = g['spec']
    slices, loci = cases.build_case(spec)
    assert cases.case_checksum(slices, loci) == g['input_crc32']
    starts = np.array(g['starts'], dtype=np.int64)
    lens = np.minimum(spec['chunk'], spec['n'] - starts).astype(np.int32)
    res = oracle.segment_chunks(slices, loci, starts, lens, spec['pcount'], spec['max_cpg'], spec['max_bp'], threads=8)
    for got, want in zip(res, g['borders']):
        assert got.tolist() == want


@pytest.mark.parametrize('name', list(cases.OFFSET_CASES))
def test_chunks_inside_a_world_match_reference_golden(name, golden_offsets):
    """`segmentor -s start0 -n len` on whole-world files (tests/golden/offset_cases.json): chunks that begin and end on the
    device kernels' boundaries.  The restatement has no such boundaries: this pins the expected values the GPU test uses."""
    g = golden_offsets[name]
    spec = g['spec']
    slices, loci = cases.build_case(spec)
    assert cases.case_checksum(slices, loci) == g['input_crc32'], 'synthetic input generator drifted'
    assert [list(c) for c in cases.offset_chunks(spec)] == g['chunks']
    starts = [c[0] for c in g['chunks']]
    lens = [c[1] for c in g['chunks']]
    res = oracle.segment_chunks(slices, loci, starts, lens, spec['pcount'], spec['max_cpg'], spec['max_bp'], threads=8)
    for (st, ln), got, want in zip(g['chunks'], res, g['borders']):
        assert got.tolist() == want, 'chunk [%d,+%d)' % (st, ln)


@pytest.mark.skipif(not oracle.have_ref(), reason='oracle/_ref/segmentor not built here')
@pytest.mark.parametrize('name', ['tiny', 'pcount0', 'zero_stretch', 'dense_w_gt_64'])
def test_live_reference_binary_agrees(name, golden_chunks):
    g = golden_chunks[name]
    spec = g['spec']
    slices, loci = cases.build_case(spec)
    b = oracle.ref_segment_arrays(slices, loci, spec['pcount'], spec['max_cpg'], spec['max_bp'])
    assert b.tolist() == g['borders']


@pytest.mark.skipif(not oracle.have_ref(), reason='oracle/_ref/segmentor not built here')
@pytest.mark.parametrize('seed', range(24))
def test_restatement_against_the_live_reference_on_random_worlds(seed):
    """Beyond the committed vectors: random worlds (dense runs, equal positions, long gaps; zero, saturated and meth == cov counts),
    random sample counts, pseudo counts nobody chose (any float in 2^-12 .. 2^12, besides 0 and the usual ones), random window
    limits — the C restatement against the reference binary run on the spot, chunk by chunk with `-s start0 -n len`."""
    import os
    import tempfile
    rng = np.random.default_rng(31000 + seed)
    n = int(rng.integers(300, 2500))
    n_samples = int(rng.choice([1, 2, 3, 5, 9]))
    kind = rng.integers(0, 4, n)
    gap = np.where(kind == 0, 0, np.where(kind == 1, rng.integers(1, 12, n), np.where(kind == 2, rng.integers(2, 300, n), rng.integers(300, 9000, n))))
    loci = (np.cumsum(gap) + 1000).astype(np.uint32)
    slices = []
    for _ in range(n_samples):
        cov = rng.integers(0, 256, n)
        mode = rng.integers(0, 6, n)
        cov = np.where(mode == 0, 0, np.where(mode == 1, 255, cov))
        meth = np.minimum(cov, np.where(mode == 2, cov, np.where(mode == 3, 0, rng.integers(0, 256, n))))
        slices.append(np.stack([meth, cov], axis=1).astype(np.uint8))
    pcount = float(np.float32(rng.choice([0.0, 0.5, 1.0, 15.0, float(np.exp2(rng.uniform(-12, 12))), float(np.exp2(rng.uniform(-12, 12)))])))
    max_cpg = int(rng.choice([2, 7, 60, 61, 129, 500, 1000]))
    max_bp = int(rng.choice([1, 40, 700, 2000, 100000]))
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for i, s in enumerate(slices):
            p = os.path.join(td, 's%02d.beta' % i)
            s.tofile(p)
            paths.append(p)
        for _ in range(3):
            ln = int(rng.integers(1, n + 1))
            st = int(rng.integers(0, n - ln + 1))
            want = oracle.ref_segment_chunk(paths, st, ln, loci[st:st + ln], pcount, max_cpg, max_bp)
            got = oracle.segment_chunks(slices, loci, [st], [ln], pcount, max_cpg, max_bp)[0]
            assert got.tolist() == want.tolist(), (seed, pcount, max_cpg, max_bp, st, ln)


@pytest.mark.parametrize('name', ['tiny', 'max_cpg2', 'pcount0', 'zero_stretch', 'dense_w_gt_64', 'deep', 'n512_deep', 'n200_islands'])
def test_threaded_restatement_matches_reference_golden(name, golden_chunks):
    """The many-thread variant the full-size GPU tests use as their checker (rows in parallel slabs, recurrence
    sequential) is the same function as the single-threaded restatement and the reference binary."""
    g = golden_chunks[name]
    spec = g['spec']
    slices, loci = cases.build_case(spec)
    for th in (1, 3, 8):
        b = oracle.segment_chunk_mt(slices, loci, spec['pcount'], spec['max_cpg'], spec['max_bp'], threads=th)
        assert b.tolist() == g['borders']


def test_threaded_restatement_on_a_default_chunk(golden_chunks):
    g = golden_chunks['default_chunk']                       # 60,000 sites: 30 slabs, ring wrap-around
    spec = g['spec']
    slices, loci = cases.build_case(spec)
    b = oracle.segment_chunk_mt(slices, loci, spec['pcount'], spec['max_cpg'], spec['max_bp'], threads=8)
    assert b.tolist() == g['borders']


def test_meth_gt_cov_is_an_error():
    """segmentor.cpp:181-188: a site with #meth > #cov aborts the run."""
    spec = cases.CHUNK_CASES['tiny']
    slices, loci = cases.build_case(spec)
    slices[1][37, 0] = slices[1][37, 1] + 1
    with pytest.raises(oracle.OracleError) as e:
        oracle.segment_chunk(slices, loci, spec['pcount'], spec['max_cpg'], spec['max_bp'])
    assert e.value.code == oracle.ORACLE_E_METH_GT_COV and e.value.sample == 1 and e.value.site == 37


def test_debug_outputs_consistent():
    spec = cases.CHUNK_CASES['tiny']
    slices, loci = cases.build_case(spec)
    b, M, T, band = oracle.segment_chunk(slices, loci, spec['pcount'], spec['max_cpg'], spec['max_bp'], debug=True)
    n = spec['n']
    assert M[0] == 0 and (M[1:] <= 0).all() and (T[1:] >= 0).all() and (T[1:] < np.arange(1, n + 1)).all()
    # recompute M from the band: M[i+1] = max_k M[k] + band[k, i-k]
    for i in (0, 1, 17, n - 1):
        ks = np.arange(max(0, i + 1 - spec['max_cpg']), i + 1)
        v = M[ks] + band[ks, i - ks]
        assert M[i + 1] == v.max() and T[i + 1] == ks[np.argmax(v)]


def test_restatement_on_non_ascending_loci_matches_reference_golden():
    """Loci that go backwards inside the chunk: the reference bars the extension and leaves the site out of the start's running sums
    (segmentor.cpp:114-117); goldens printed by the reference binary (tests/golden/make_golden_disorder.py)."""
    import json
    import os.path as op
    with open(op.join(op.dirname(__file__), 'golden', 'disorder_cases.json')) as f:
        golden = json.load(f)
    assert sorted(golden) == sorted(cases.DISORDER_CASES)
    for name, g in golden.items():
        spec = cases.DISORDER_CASES[name]
        slices, loci = cases.build_disorder_case(spec)
        assert cases.case_checksum(slices, loci) == g['input_crc32'], name
        assert np.any(np.diff(loci.astype(np.int64)) < 0), name
        b = oracle.segment_chunk(slices, loci, spec['pcount'], spec['max_cpg'], spec['max_bp'])
        assert b.tolist() == g['borders'], name
        if oracle.have_ref():
            assert oracle.ref_segment_arrays(slices, loci, spec['pcount'], spec['max_cpg'], spec['max_bp']).tolist() == g['borders'], name
