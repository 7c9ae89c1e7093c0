"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes access to (a) ``liboracle_segment.so``, our plain-C restatement of the reference chunk DP
(segment_oracle.c, cites segmentor.cpp line by line) and (b) ``_ref/segmentor``, the reference's own
sources compiled where they lie (Makefile target ``ref``; exists only where /root/reference was available
at build time, travels to the GPU box as a prebuilt binary).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity status: pinned (tests/test_oracle_golden.py checks the restatement against border lists captured from
``_ref/segmentor``: tests/golden/chunk_cases.json).
"""
import ctypes as C
import os
import os.path as op
import subprocess
import tempfile

import numpy as np

HERE = op.dirname(op.abspath(__file__))
LIB_PATH = op.join(HERE, 'liboracle_segment.so')
REF_BIN = op.join(HERE, '_ref', 'segmentor')

ORACLE_OK, ORACLE_E_ARG, ORACLE_E_METH_GT_COV, ORACLE_E_NOMEM = 0, -1, -2, -3

_lib = None


def build(ref=True):
    """(Re)build the restatement and, when the reference tree is present, the reference binary."""
    subprocess.check_call(['make', '-s', '-C', HERE, 'all'] + (['ref'] if ref else []))


def lib():
    global _lib
    if _lib is None:
        if not op.isfile(LIB_PATH):
            build(ref=False)
        L = C.CDLL(LIB_PATH)
        L.oracle_segment_chunk.restype = C.c_int
        L.oracle_segment_chunk.argtypes = [
            C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_uint32,
            C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
            C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_segment_chunks.restype = C.c_int
        L.oracle_segment_chunks.argtypes = [
            C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
            C.c_float, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
        L.oracle_segment_chunk_mt.restype = C.c_int
        L.oracle_segment_chunk_mt.argtypes = [
            C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_uint32, C.c_int,
            C.c_void_p, C.POINTER(C.c_int)]
        L.oracle_sample_terms.restype = None
        L.oracle_sample_terms.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]
        L.probe_log2f_fill.restype = None
        L.probe_log2f_fill.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_int]
        L.probe_log2_1mp_fill.restype = None
        L.probe_log2_1mp_fill.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_int]
        L.probe_log2f_compare.restype = C.c_uint64
        L.probe_log2f_compare.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
        L.probe_log2_1mp_compare.restype = C.c_uint64
        L.probe_log2_1mp_compare.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
        L.probe_log2_1mp_maxulp.restype = C.c_uint64
        L.probe_log2_1mp_maxulp.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_int]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    def __init__(self, code, sample=-1, site=-1):
        super().__init__('oracle rc=%d (sample %d, site %d)' % (code, sample, site))
        self.code, self.sample, self.site = code, sample, site


def _ptr_array(arrs):
    P = (C.c_void_p * len(arrs))()
    for i, a in enumerate(arrs):
        P[i] = a.ctypes.data
    return P


def segment_chunk(slices, loci, pcount, max_cpg, max_bp, debug=False):
    """slices: list of uint8 arrays [n,2] (one per sample, CLI order); loci: uint32[n].
    Returns borders (int32, ascending, incl. 0 and n); with debug=True also (M, T, band[n,max_cpg])."""
    slices = [np.ascontiguousarray(s, dtype=np.uint8) for s in slices]
    n = slices[0].shape[0]
    loci = np.ascontiguousarray(loci, dtype=np.uint32)
    assert loci.shape[0] == n
    borders = np.empty(n + 1, dtype=np.int32)
    nb, bs, bi = C.c_int(0), C.c_int(-1), C.c_int(-1)
    M = T = band = None
    if debug:
        M = np.empty(n + 1, dtype=np.float64)
        T = np.empty(n + 1, dtype=np.int32)
        band = np.empty((n, max_cpg), dtype=np.float64)
    rc = lib().oracle_segment_chunk(
        _ptr_array(slices), len(slices), n, loci.ctypes.data, C.c_float(pcount), int(max_cpg), int(max_bp),
        borders.ctypes.data, C.byref(nb), C.byref(bs), C.byref(bi),
        M.ctypes.data if debug else None, T.ctypes.data if debug else None, band.ctypes.data if debug else None)
    if rc != ORACLE_OK:
        raise OracleError(rc, bs.value, bi.value)
    b = borders[:nb.value].copy()
    return (b, M, T, band) if debug else b


def segment_chunk_mt(slices, loci, pcount, max_cpg, max_bp, threads=None):
    """One big chunk on many threads (cost rows in parallel, recurrence sequential): same arithmetic as
    segment_chunk, for full-size cases (50,000 sites x 512 samples at max_cpg 5000)."""
    slices = [np.ascontiguousarray(s, dtype=np.uint8) for s in slices]
    n = slices[0].shape[0]
    loci = np.ascontiguousarray(loci, dtype=np.uint32)
    assert loci.shape[0] == n
    borders = np.empty(n + 1, dtype=np.int32)
    nb = C.c_int(0)
    rc = lib().oracle_segment_chunk_mt(_ptr_array(slices), len(slices), n, loci.ctypes.data, C.c_float(pcount), int(max_cpg),
                                       int(max_bp), int(threads or os.cpu_count() or 1), borders.ctypes.data, C.byref(nb))
    if rc != ORACLE_OK:
        raise OracleError(rc)
    return borders[:nb.value].copy()


def segment_chunks(samples, loci, start0, lens, pcount, max_cpg, max_bp, threads=1):
    """samples: list of whole uint8 arrays [n_total,2]; chunks [start0[c], start0[c]+lens[c]).
    Returns list of int32 border arrays (relative to each chunk start)."""
    samples = [np.ascontiguousarray(s, dtype=np.uint8) for s in samples]
    loci = np.ascontiguousarray(loci, dtype=np.uint32)
    start0 = np.ascontiguousarray(start0, dtype=np.int64)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    nch = len(start0)
    cap = int(lens.sum()) + nch
    out = np.empty(cap, dtype=np.int32)
    off = np.empty(nch + 1, dtype=np.int64)
    rc = lib().oracle_segment_chunks(
        _ptr_array(samples), len(samples), loci.ctypes.data, start0.ctypes.data, lens.ctypes.data, nch,
        C.c_float(pcount), int(max_cpg), int(max_bp), int(threads), out.ctypes.data, cap, off.ctypes.data)
    if rc != ORACLE_OK:
        raise OracleError(rc)
    return [out[off[c]:off[c + 1]].copy() for c in range(nch)]


def sample_terms(nmeth, ntotal, pcount):
    nmeth = np.ascontiguousarray(nmeth, dtype=np.float32)
    ntotal = np.ascontiguousarray(ntotal, dtype=np.float32)
    out = np.empty_like(nmeth)
    lib().oracle_sample_terms(nmeth.ctypes.data, ntotal.ctypes.data, nmeth.size, C.c_float(pcount), out.ctypes.data)
    return out


# ------------------------------------------------------------------------------------------------
# the reference binary (oracle/_ref/segmentor)
# ------------------------------------------------------------------------------------------------
def have_ref():
    return op.isfile(REF_BIN) and os.access(REF_BIN, os.X_OK)


def ref_segment_chunk(beta_paths, start0, n, loci_slice, pcount, max_cpg, max_bp, timeout=None):
    """Run the reference binary exactly the way segment.py:48-55 does, minus tabix: loci on stdin (one per
    line), `-s start0 -n n -max_cpg M -ps P -max_bp B`; returns the ints it prints (relative borders)."""
    assert have_ref(), 'oracle/_ref/segmentor is not built (make -C oracle ref needs /root/reference)'
    for p in beta_paths:
        assert p.endswith('.beta'), 'segmentor only accepts argv tokens ending in .beta (main.cpp:101-107)'
    cmd = [REF_BIN] + list(beta_paths) + ['-s', str(int(start0)), '-n', str(int(n)),
                                            '-max_cpg', str(int(max_cpg)), '-ps', repr(float(pcount)),
                                            '-max_bp', str(int(max_bp))]
    stdin = ('\n'.join(str(int(x)) for x in loci_slice) + '\n').encode()
    res = subprocess.run(cmd, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    if res.returncode != 0:
        raise subprocess.CalledProcessError(res.returncode, cmd, res.stdout, res.stderr)
    return np.array(list(map(int, res.stdout.decode().split())), dtype=np.int32)


def ref_segment_arrays(slices, loci, pcount, max_cpg, max_bp):
    """Convenience: write the slices to temporary .beta files and run the reference binary on them."""
    n = slices[0].shape[0]
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for i, s in enumerate(slices):
            p = op.join(td, 's%04d.beta' % i)
            np.ascontiguousarray(s, dtype=np.uint8).tofile(p)
            paths.append(p)
        return ref_segment_chunk(paths, 0, n, loci, pcount, max_cpg, max_bp)
