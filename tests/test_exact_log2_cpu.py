"""The bit-exact restatements of glibc's log2f / log2 (wgbs_tools_amd/csrc/exact_log2.h), HOST build, against the
live libm over the path's whole input domain: every float p in (0,1] for log2f(p) — plus every subnormal — and
every float p in (0,1) for log2(1.0-(double)p).  Third-party arithmetic pinned by ourselves."""
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


@pytest.fixture(scope='module')
def exact():
    if not op.isfile(LIB) or op.getmtime(LIB) < max(op.getmtime(SRC), op.getmtime(HDR)):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC', '-pthread', SRC, '-o', LIB])
    L = C.CDLL(LIB)
    L.exact_log2f_fill.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_int]
    L.exact_log2_1mp_fill.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_int]
    L.exact_sample_terms.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]
    return L


def test_log2f_and_log2_exhaustive_vs_libm(exact):
    O = oracle.lib()
    th = os.cpu_count() or 1
    first, last = 0x00000001, 0x3f800000            # subnormals, normals below 1, and 1.0 itself
    B = 1 << 26
    bad_f = bad_d = 0
    q = first
    while q <= last:
        cnt = min(B, last - q + 1)
        a = np.empty(cnt, np.uint32)
        exact.exact_log2f_fill(q, cnt, a.ctypes.data, th)
        fb = C.c_uint32(0)
        bad_f += O.probe_log2f_compare(q, cnt, a.ctypes.data, th, C.byref(fb))
        if q + cnt - 1 >= 0x00800000:               # log2(1-p): normal p < 1 (tiny p gives log2(1.0) = 0 either way)
            lo = max(q, 0x00800000)
            c2 = q + cnt - lo - (1 if q + cnt - 1 == last else 0)
            d = np.empty(c2, np.uint64)
            exact.exact_log2_1mp_fill(lo, c2, d.ctypes.data, th)
            bad_d += O.probe_log2_1mp_compare(lo, c2, d.ctypes.data, th, C.byref(fb))
        q += cnt
    assert bad_f == 0 and bad_d == 0


@pytest.mark.parametrize('pcount', [15.0, 0.0, 0.5, 1.0, 3.25])
def test_sample_term_host_build_matches_oracle(exact, pcount):
    rng = np.random.default_rng(11)
    t = np.concatenate([np.arange(0, 5000), rng.integers(0, 255 * 1000, 1000000)]).astype(np.float32)
    m = np.minimum(np.floor(rng.random(t.size) * (t + 1)), t).astype(np.float32)
    m[:5000:3] = 0
    m[1:5000:3] = t[1:5000:3]
    got = np.empty_like(t)
    exact.exact_sample_terms(m.ctypes.data, t.ctypes.data, t.size, C.c_float(pcount), got.ctypes.data)
    want = oracle.sample_terms(m, t, pcount)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
