"""The arithmetic of the scoring kernel (wgbs_tools_amd/csrc/exact_log2.h), HOST build, against the LIVE libm over the
path's whole input domain (third-party arithmetic pinned by ourselves):

  * wg_log2f (fused form, used by the kernels) and wg_log2f_nofma (operation-by-operation form): bit-identical to
    libm's log2f for every float in (0, 1] and every subnormal;
  * wg_log2: bit-identical to libm's log2(1.0-(double)p) for every float p in (0, 1);
  * wg_fast_log2: within 1 ulp of libm's log2 on that same domain — the measured bound the Ziv-style rounding test of
    wg_sample_term relies on;
  * wg_sample_term (fast path + exact fallback) == wg_sample_term_plain == the oracle's term, bit for bit.
"""
import ctypes as C
import os
import os.path as op
import subprocess

import numpy as np
import pytest

from oracle import oracle

ROOT = op.dirname(op.dirname(op.abspath(__file__)))
SRC = op.join(ROOT, 'tests', 'native', 'exact_host.cpp')
LIB = op.join(ROOT, 'tests', 'native', 'libexact_host.so')
HDR = op.join(ROOT, 'wgbs_tools_amd', 'csrc', 'exact_log2.h')
FAST_FIRST = 0x25000000          # bits of 2^-53f: below it 1.0 - (double)p == 1.0


def load_exact():
    if not op.isfile(LIB) or op.getmtime(LIB) < max(op.getmtime(SRC), op.getmtime(HDR)):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC', '-pthread', SRC, '-o', LIB])
    L = C.CDLL(LIB)
    for f in ('exact_log2f_fill', 'exact_log2f_nofma_fill'):
        getattr(L, f).argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_int]
    for f in ('exact_log2_1mp_fill', 'fast_log2_1mp_fill'):
        getattr(L, f).argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_int]
    for f in ('exact_sample_terms', 'exact_sample_terms_plain'):
        getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]
    L.exact_ks_mismatches.restype = C.c_uint64
    L.exact_ks_mismatches.argtypes = [C.c_uint32, C.c_uint64, C.c_int, C.c_void_p]
    L.lookup_rows_violations.restype = C.c_uint64
    L.lookup_rows_violations.argtypes = [C.c_float, C.c_uint32, C.c_uint32]
    L.ks_log2_max_ulp.restype = C.c_uint64
    L.ks_log2_max_ulp.argtypes = [C.c_uint32, C.c_uint64, C.c_int]
    L.ks_sum_ulp_gap.restype = C.c_uint64
    L.ks_sum_ulp_gap.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float]
    L.sum_ulp_gap.restype = C.c_uint64
    L.sum_ulp_gap.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float]
    return L


@pytest.fixture(scope='module')
def exact():
    return load_exact()


def test_log2f_and_log2_exhaustive_vs_libm(exact):
    O = oracle.lib()
    th = os.cpu_count() or 1
    first, last = 0x00000001, 0x3f800000            # subnormals, normals below 1, and 1.0 itself
    B = 1 << 26
    bad_f = bad_n = bad_d = 0
    max_ulp = 0
    q = first
    while q <= last:
        cnt = min(B, last - q + 1)
        a = np.empty(cnt, np.uint32)
        fb = C.c_uint32(0)
        exact.exact_log2f_fill(q, cnt, a.ctypes.data, th)
        bad_f += O.probe_log2f_compare(q, cnt, a.ctypes.data, th, C.byref(fb))
        exact.exact_log2f_nofma_fill(q, cnt, a.ctypes.data, th)
        bad_n += O.probe_log2f_compare(q, cnt, a.ctypes.data, th, C.byref(fb))
        if q + cnt - 1 >= 0x00800000:               # log2(1-p): normal p < 1 (tiny p gives log2(1.0) = 0 either way)
            lo = max(q, 0x00800000)
            c2 = q + cnt - lo - (1 if q + cnt - 1 == last else 0)
            d = np.empty(c2, np.uint64)
            exact.exact_log2_1mp_fill(lo, c2, d.ctypes.data, th)
            bad_d += O.probe_log2_1mp_compare(lo, c2, d.ctypes.data, th, C.byref(fb))
            lo_f = max(lo, FAST_FIRST)                  # wg_fast_log2's domain: x = 1 - p strictly below 1
            c3 = lo + c2 - lo_f
            if c3 > 0:
                exact.fast_log2_1mp_fill(lo_f, c3, d.ctypes.data, th)
                max_ulp = max(max_ulp, int(O.probe_log2_1mp_maxulp(lo_f, c3, d.ctypes.data, th)))
        q += cnt
    assert bad_f == 0 and bad_n == 0 and bad_d == 0
    assert max_ulp <= 1, 'fast log2 strays %d ulp from libm: the 6-ulp bound of wg_sample_term no longer holds' % max_ulp


def _term_inputs(seed, n_random):
    rng = np.random.default_rng(seed)
    big = np.repeat(np.array([255 * 8000, 255 * 8000 - 1, 255 * 1000, 2 ** 20, 2 ** 20 + 1, 65535, 65536]), 4)   # the ABI's largest sums
    t = np.concatenate([np.arange(0, 5000), big, rng.integers(0, 255 * 1000, n_random),
                        rng.integers(0, 4000, n_random)]).astype(np.float32)
    m = np.minimum(np.floor(rng.random(t.size) * (t + 1)), t).astype(np.float32)
    m[:5000:3] = 0
    m[1:5000:3] = t[1:5000:3]
    m[5000:5000 + big.size:4] = 0                       # nmeth = 0, = ntotal, = ntotal - 1, = 1 at the extremes
    m[5001:5000 + big.size:4] = t[5001:5000 + big.size:4]
    m[5002:5000 + big.size:4] = t[5002:5000 + big.size:4] - 1
    m[5003:5000 + big.size:4] = 1
    return m, t


@pytest.mark.parametrize('pcount', [15.0, 0.0, 0.5, 0.99999994, 1.0, 3.25, 4.0, 3.9999998, 1000.0, 16777216.0, 16777218.0, 1e30, 3e38, 1e-30, 2e-6, 1e-7])
def test_sample_term_forms_match_oracle(exact, pcount):
    m, t = _term_inputs(11, 1000000)
    want = oracle.sample_terms(m, t, pcount)
    for fn in (exact.exact_sample_terms, exact.exact_sample_terms_plain):
        got = np.empty_like(t)
        fn(m.ctypes.data, t.ctypes.data, t.size, C.c_float(pcount), got.ctypes.data)
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), fn


def test_sample_term_forms_on_random_pseudo_counts(exact):
    """The same comparison for pseudo counts nobody chose: 160 floats, log-uniform over 2^-24 .. 2^24, plus the neighbours of
    every power of two in between (where the exponent of nmeth + pc changes: the k-scaled tables' row index) — all three term
    modes (plain, fast with guards, guard-free on the k-scaled tables) against the oracle's libm arithmetic."""
    rng = np.random.default_rng(2024)
    pcs = list(np.exp2(rng.uniform(-24, 24, 160)).astype(np.float32))
    for k in range(-6, 22, 3):
        x = np.float32(2.0 ** k)
        pcs += [np.nextafter(x, np.float32(0)), x, np.nextafter(x, np.float32(np.inf))]
    m, t = _term_inputs(12, 60000)
    for pc in pcs:
        want = oracle.sample_terms(m, t, float(pc))
        for fn in (exact.exact_sample_terms, exact.exact_sample_terms_plain):
            got = np.empty_like(t)
            fn(m.ctypes.data, t.ctypes.data, t.size, C.c_float(float(pc)), got.ctypes.data)
            bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
            assert bad.size == 0, (float(pc), fn, m[bad[0]], t[bad[0]], got[bad[0]], want[bad[0]])


def test_k_scaled_table_forms_are_the_same_functions(exact):
    """wg_fast_log2_ks / wg_log2f_ks (normalisation and exponent term folded into per-(k, i) tables; the narrow scoring
    tiles) against wg_fast_log2(1 - p) and the libm restatement wg_log2f(p) for EVERY float p in [2^-53, 1) whose
    argument the tables cover (exponent >= -23): bit-identical."""
    skipped = C.c_uint64(0)
    count = 0x3f800000 - FAST_FIRST
    bad = exact.exact_ks_mismatches(FAST_FIRST, count, os.cpu_count() or 1, C.byref(skipped))
    assert bad == 0
    n_small_p = 0x34000000 - FAST_FIRST                 # p below ~0.7 * 2^-23: outside the log2f table
    assert skipped.value < n_small_p + 0x800000 + 2000, skipped.value


def _hdr_const(name):
    import re
    m = re.search(r'#define\s+%s\s+(\d+)' % name, open(HDR).read())
    return int(m.group(1))


@pytest.mark.parametrize('pcount', [15.0, 1.0, 2.5, 0.5, 0.0])
def test_sample_term_on_the_whole_small_count_lattice(exact, pcount):
    """Every (nmeth, ntotal) with ntotal <= 2200 (2.4 M blocks — the counts short blocks really have), all forms against
    the oracle's term."""
    t = np.repeat(np.arange(0, 2201), np.arange(1, 2202)).astype(np.float32)
    m = np.concatenate([np.arange(0, k + 1) for k in range(0, 2201)]).astype(np.float32)
    want = oracle.sample_terms(m, t, pcount)
    for fn in (exact.exact_sample_terms, exact.exact_sample_terms_plain):
        got = np.empty_like(t)
        fn(m.ctypes.data, t.ctypes.data, t.size, C.c_float(pcount), got.ctypes.data)
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), fn


@pytest.mark.parametrize('pcount', [1.0, 1.0000001, 1.5, 3.3333333, 4.0, 15.0, 15.000001, 100.0, 16777216.0])
def test_lookup_table_rows_cover_every_computed_argument(exact, pcount):
    """wg_lookup_rows (how many exponent rows the k-scaled tables get) against the p and 1 - p the kernels compute, for
    EVERY block total the ABI admits (1 .. 255 * 8000) at the extreme counts: no index outside the rows, 0 < p < 1."""
    assert exact.lookup_rows_violations(C.c_float(pcount), 1, 255 * 8000) == 0


def test_shortened_log2_polynomial_error_and_guard_band(exact):
    """The narrow scoring tiles' log2(1 - p) drops the highest polynomial term: its distance from libm, measured over EVERY
    float p the tables cover, must stay within WG_KS_LOG2_MAX_ULP, and the guard band must cover the derived 2E + 2 bound;
    the sums actually rounded stay inside it on 8 M blocks per pseudo count."""
    e_max, guard = _hdr_const('WG_KS_LOG2_MAX_ULP'), _hdr_const('WG_GUARD_ULPS_KS')
    got = int(exact.ks_log2_max_ulp(FAST_FIRST, 0x3f800000 - FAST_FIRST, os.cpu_count() or 1))
    assert 1 < got <= e_max, got
    assert guard >= 2 * e_max + 4
    m, t = _term_inputs(5, 4000000)
    for pcount in (15.0, 4.0, 100.0):
        gap = int(exact.ks_sum_ulp_gap(m.ctypes.data, t.ctypes.data, t.size, C.c_float(pcount)))
        assert gap <= 2 * e_max + 2, (pcount, gap)


@pytest.mark.parametrize('pcount', [15.0, 0.5, 0.0])
def test_fast_sum_stays_inside_the_guard_band(exact, pcount):
    """|s' - s| of the second term, measured on 8 M blocks: must respect the derived 6-ulp bound (guard band is 16)."""
    m, t = _term_inputs(5, 4000000)
    gap = int(exact.sum_ulp_gap(m.ctypes.data, t.ctypes.data, t.size, C.c_float(pcount)))
    assert gap <= 6, gap
