"""ctypes binding of libwgbsseg.so (include/wgbsseg.h).  Thin by design: numpy arrays in, numpy arrays out.

There is no Python/CPU implementation behind this module: if the HIP library is missing or no gfx950 device is
visible, every entry point raises (``NativeLibraryError`` / ``SegmentorError``) -- loudly, never a fallback.
"""
import ctypes as C
import os
import os.path as op

import numpy as np

HERE = op.dirname(op.abspath(__file__))
LIB_PATH = op.join(HERE, 'csrc', 'libwgbsseg.so')
# Another build of the same ABI (tests and A/B measurements only: tools/build_carrybug_lib.sh — the library with a fixed defect put back, to
# show that the suite's fuzz catches it — and tools/build_variants.sh): honoured only together with WGBSSEG_ALLOW_LIB_OVERRIDE=1, and said on
# stderr, so that a stray variable cannot make a CLI run load some other shared object silently.
if os.environ.get('WGBSSEG_LIB') and os.environ.get('WGBSSEG_ALLOW_LIB_OVERRIDE') == '1':
    LIB_PATH = os.environ['WGBSSEG_LIB']
    import sys as _sys
    print('[wgbsseg] WGBSSEG_LIB: loading %s instead of the in-tree library' % LIB_PATH, file=_sys.stderr)
SYNTH_LIB_PATH = op.join(HERE, 'csrc', 'libwgbssynth.so')

OK, E_ARG, E_METH_GT_COV, E_NOMEM, E_HIP, E_LOCI_ORDER, E_CAPACITY, E_STATE = 0, -1, -2, -3, -4, -5, -6, -7

# every symbol include/wgbsseg.h declares (tests check the built library exports exactly these)
ABI_VERSION = 220          # include/wgbsseg.h WGBSSEG_VERSION this binding's prototypes describe
EXPORTS = ['wgbsseg_version', 'wgbsseg_device_count', 'wgbsseg_create', 'wgbsseg_destroy',
           'wgbsseg_set_betas_host', 'wgbsseg_set_betas_device', 'wgbsseg_set_loci_host', 'wgbsseg_set_loci_device',
           'wgbsseg_segment_chunks', 'wgbsseg_segment_regions', 'wgbsseg_segment_chunks_host', 'wgbsseg_prefix_sums', 'wgbsseg_scan_only',
           'wgbsseg_get_timings', 'wgbsseg_debug_fetch', 'wgbsseg_debug_sample_terms', 'wgbsseg_debug_log2',
           'wgbsseg_debug_div', 'wgbsseg_debug_check_div', 'wgbsseg_debug_div_short', 'wgbsseg_add_loci', 'wgbsseg_add_loci_borders', 'wgbsseg_block_sums', 'wgbsseg_last_block_sums_ms',
           'wgbsseg_set_site_base', 'wgbsseg_stitch_regions', 'wgbsseg_group_create', 'wgbsseg_group_destroy', 'wgbsseg_group_size',
           'wgbsseg_group_plan', 'wgbsseg_group_load_host', 'wgbsseg_group_share_set_device', 'wgbsseg_group_segment_regions', 'wgbsseg_group_segment_region_range',
           'wgbsseg_group_get_timings', 'wgbsseg_plan_shares', 'wgbsseg_set_lbetas_host',
           'wgbsseg_convert_regions', 'wgbsseg_patbeta_create', 'wgbsseg_patbeta_feed', 'wgbsseg_patbeta_finish',
           'wgbsseg_patbeta_destroy', 'wgbsseg_patbeta_kernel_ms', 'wgbsseg_group_load_host_async', 'wgbsseg_group_load_wait',
           'wgbsseg_marker_stats', 'wgbsseg_blocks_parse', 'wgbsseg_blocks_write_table', 'wgbsseg_blocks_write_bedgraph',
           'wgbsseg_format_fixed', 'wgbsseg_bed_parse', 'wgbsseg_bed_write_annotated', 'wgbsseg_debug_canonical_float',
           'wgbsseg_first_batch_items', 'wgbsseg_plan_shares_weighted']


class NativeLibraryError(RuntimeError):
    pass


class SegmentorError(RuntimeError):
    """A failed native call; `.code` is the WGBSSEG_E_* value, the message is the library's."""

    def __init__(self, code, msg):
        super().__init__('[wgbsseg %d] %s' % (code, msg))
        self.code = code
        self.msg = msg


class Params(C.Structure):
    _fields_ = [('pseudo_count', C.c_float), ('max_cpg', C.c_uint32), ('max_bp', C.c_uint32)]


class Timings(C.Structure):
    _fields_ = [('scan_ms', C.c_double), ('window_ms', C.c_double), ('cost_ms', C.c_double), ('dp_ms', C.c_double),
                ('trace_ms', C.c_double), ('total_ms', C.c_double), ('sites', C.c_int64), ('pairs', C.c_int64),
                ('evals', C.c_int64), ('scan_bytes', C.c_int64), ('max_window', C.c_int32), ('n_stages', C.c_int32),
                ('scan_launches', C.c_int32), ('div_short', C.c_int32), ('scan_main_ms', C.c_double),
                ('scan_main_bytes', C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# wgbsseg_batch_fn: (user, starts, ends, n, out_ptr, out_cnt) -> 0 on success
BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_void_p),
                       C.POINTER(C.c_int64))

_lib = None
_synth = None
ERRLEN = 512
_hip_preloaded = False


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP/HSA runtime.  The PyTorch wheel bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7, the same as /opt/rocm's): whichever copy is loaded first serves both torch and this library,
    but if ours came first torch would load a second runtime by file name and find no GPU.  So when torch is
    installed and not yet imported, map its copy first (no `import torch`: that costs seconds)."""
    global _hip_preloaded
    if _hip_preloaded:
        return
    _hip_preloaded = True
    import importlib.util
    import sys
    if 'torch' in sys.modules:
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = op.join(op.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if op.isfile(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen libwgbsseg.so and declare the prototypes.  Raises NativeLibraryError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not op.isfile(LIB_PATH):
        raise NativeLibraryError('%s is missing: build it with `python -m wgbs_tools_amd.build` (needs hipcc). '
                                 'There is no CPU fallback.' % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.wgbsseg_version.restype = i32
    got = L.wgbsseg_version()
    if got != ABI_VERSION:
        # argument lists differ between versions (e.g. wgbsseg_scan_only, 200 -> 210): never call through prototypes of another one
        raise NativeLibraryError('%s reports ABI version %d, this binding was written against %d (include/wgbsseg.h WGBSSEG_VERSION): '
                                 'rebuild it with `python -m wgbs_tools_amd.build -f`' % (LIB_PATH, got, ABI_VERSION))
    L.wgbsseg_device_count.restype = i32
    L.wgbsseg_create.restype = i32
    L.wgbsseg_create.argtypes = [i32, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.wgbsseg_destroy.restype = None
    L.wgbsseg_destroy.argtypes = [vp]
    L.wgbsseg_set_betas_host.restype = i32
    L.wgbsseg_set_betas_host.argtypes = [vp, C.POINTER(vp), i64, i64, C.c_char_p, C.c_size_t]
    L.wgbsseg_set_lbetas_host.restype = i32
    L.wgbsseg_set_lbetas_host.argtypes = [vp, C.POINTER(vp), i64, i64, C.c_char_p, C.c_size_t]
    L.wgbsseg_set_betas_device.restype = i32
    L.wgbsseg_set_betas_device.argtypes = [vp, vp, i64, i64, i64, C.c_char_p, C.c_size_t]
    L.wgbsseg_set_loci_host.restype = i32
    L.wgbsseg_set_loci_host.argtypes = [vp, vp, i64, C.c_char_p, C.c_size_t]
    L.wgbsseg_set_loci_device.restype = i32
    L.wgbsseg_set_loci_device.argtypes = [vp, vp, i64, C.c_char_p, C.c_size_t]
    L.wgbsseg_segment_chunks.restype = i32
    L.wgbsseg_segment_chunks.argtypes = [vp, vp, vp, i64, C.POINTER(Params), vp, i64, vp, C.c_char_p, C.c_size_t]
    L.wgbsseg_segment_regions.restype = i32
    L.wgbsseg_segment_regions.argtypes = [vp, vp, vp, i64, i64, C.POINTER(Params), vp, i64, vp, vp, C.c_char_p, C.c_size_t]
    L.wgbsseg_segment_chunks_host.restype = i32
    L.wgbsseg_segment_chunks_host.argtypes = [vp, i64, i64, i64, vp, vp, vp, i64, C.POINTER(Params), i32, vp, i64, vp,
                                              C.c_char_p, C.c_size_t]
    L.wgbsseg_prefix_sums.restype = i32
    L.wgbsseg_prefix_sums.argtypes = [vp, i64, i64, vp, C.c_char_p, C.c_size_t]
    L.wgbsseg_scan_only.restype = i32
    L.wgbsseg_scan_only.argtypes = [vp, vp, vp, i64, i32, i32, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(i64), C.c_char_p, C.c_size_t]
    L.wgbsseg_get_timings.restype = i32
    L.wgbsseg_get_timings.argtypes = [vp, C.POINTER(Timings)]
    L.wgbsseg_debug_fetch.restype = i64
    L.wgbsseg_debug_fetch.argtypes = [vp, C.c_char_p, vp, i64]
    L.wgbsseg_debug_sample_terms.restype = i32
    L.wgbsseg_debug_sample_terms.argtypes = [vp, vp, vp, i64, C.c_float, vp]
    L.wgbsseg_debug_log2.restype = i32
    L.wgbsseg_debug_log2.argtypes = [vp, C.c_uint32, i64, vp, vp, vp]
    L.wgbsseg_debug_div_short.restype = i32
    L.wgbsseg_debug_div_short.argtypes = [vp, vp, vp, i64, vp]
    L.wgbsseg_debug_check_div.restype = i32
    L.wgbsseg_debug_check_div.argtypes = [vp, C.c_float, i32, vp]
    L.wgbsseg_debug_div.restype = i32
    L.wgbsseg_debug_div.argtypes = [vp, vp, vp, i64, vp, vp]
    L.wgbsseg_block_sums.restype = i32
    L.wgbsseg_block_sums.argtypes = [vp, vp, vp, i64, i32, C.c_uint32, vp, C.c_char_p, C.c_size_t]
    L.wgbsseg_last_block_sums_ms.restype = C.c_double
    L.wgbsseg_last_block_sums_ms.argtypes = [vp]
    L.wgbsseg_set_site_base.restype = i32
    L.wgbsseg_set_site_base.argtypes = [vp, i64]
    L.wgbsseg_stitch_regions.restype = i32
    L.wgbsseg_stitch_regions.argtypes = [vp, vp, i64, i64, BATCH_FN, vp, i32, vp, i64, vp, vp, C.c_char_p, C.c_size_t]
    L.wgbsseg_first_batch_items.restype = i32
    L.wgbsseg_first_batch_items.argtypes = [vp, vp, i64, i64, i32, vp, vp, i64, C.POINTER(i64), C.POINTER(i64), C.c_char_p, C.c_size_t]
    L.wgbsseg_plan_shares.restype = i32
    L.wgbsseg_plan_shares.argtypes = [vp, i64, vp, vp, i64, i64, C.POINTER(Params), i32, i64, vp, vp, vp, vp, vp, vp, C.c_char_p, C.c_size_t]
    L.wgbsseg_plan_shares_weighted.restype = i32
    L.wgbsseg_plan_shares_weighted.argtypes = [vp, i64, vp, vp, i64, i64, C.POINTER(Params), i32, vp, i64, vp, vp, vp, vp, vp, vp, C.c_char_p, C.c_size_t]
    L.wgbsseg_group_create.restype = i32
    L.wgbsseg_group_create.argtypes = [vp, i32, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.wgbsseg_group_destroy.restype = None
    L.wgbsseg_group_destroy.argtypes = [vp]
    L.wgbsseg_group_size.restype = i32
    L.wgbsseg_group_size.argtypes = [vp]
    L.wgbsseg_group_plan.restype = i32
    L.wgbsseg_group_plan.argtypes = [vp, vp, i64, vp, vp, i64, i64, C.POINTER(Params), i64, vp, vp, vp, vp, C.c_char_p, C.c_size_t]
    L.wgbsseg_group_load_host.restype = i32
    L.wgbsseg_group_load_host.argtypes = [vp, C.POINTER(vp), i64, i64, C.c_char_p, C.c_size_t]
    L.wgbsseg_group_load_host_async.restype = i32
    L.wgbsseg_group_load_host_async.argtypes = [vp, C.POINTER(vp), i64, i64, C.c_char_p, C.c_size_t]
    L.wgbsseg_group_load_wait.restype = i32
    L.wgbsseg_group_load_wait.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.wgbsseg_group_share_set_device.restype = i32
    L.wgbsseg_group_share_set_device.argtypes = [vp, i32, vp, i64, i64, C.c_char_p, C.c_size_t]
    L.wgbsseg_group_segment_regions.restype = i32
    L.wgbsseg_group_segment_regions.argtypes = [vp, vp, i64, vp, vp, C.c_char_p, C.c_size_t]
    L.wgbsseg_group_segment_region_range.restype = i32
    L.wgbsseg_group_segment_region_range.argtypes = [vp, i64, i64, vp, i64, vp, vp, C.c_char_p, C.c_size_t]
    L.wgbsseg_group_get_timings.restype = i32
    L.wgbsseg_group_get_timings.argtypes = [vp, i32, C.POINTER(Timings)]
    L.wgbsseg_convert_regions.restype = i32
    L.wgbsseg_convert_regions.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, C.c_char_p, C.c_size_t]
    L.wgbsseg_patbeta_create.restype = i32
    L.wgbsseg_patbeta_create.argtypes = [i32, i64, i64, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.wgbsseg_patbeta_feed.restype = i32
    L.wgbsseg_patbeta_feed.argtypes = [vp, C.c_char_p, i64, C.c_char_p, C.c_size_t]
    L.wgbsseg_patbeta_finish.restype = i32
    L.wgbsseg_patbeta_finish.argtypes = [vp, i32, vp, C.c_char_p, C.c_size_t]
    L.wgbsseg_patbeta_destroy.restype = None
    L.wgbsseg_patbeta_destroy.argtypes = [vp]
    L.wgbsseg_patbeta_kernel_ms.restype = C.c_double
    L.wgbsseg_patbeta_kernel_ms.argtypes = [vp]
    L.wgbsseg_marker_stats.restype = i32
    L.wgbsseg_marker_stats.argtypes = [vp, vp, i32, vp, i32, i64, vp, C.c_char_p, C.c_size_t]
    L.wgbsseg_add_loci.restype = i32
    L.wgbsseg_add_loci.argtypes = [vp, i64, vp, C.POINTER(C.c_char_p), i32, vp, vp, i64, C.c_char_p, i32, i32, C.c_char_p, C.c_size_t]
    L.wgbsseg_add_loci_borders.restype = i32
    L.wgbsseg_add_loci_borders.argtypes = [vp, i64, vp, C.POINTER(C.c_char_p), i32, vp, vp, i64, i64, C.c_char_p, i32, i32,
                                           C.POINTER(i64), C.POINTER(i64), C.c_char_p, C.c_size_t]
    L.wgbsseg_blocks_parse.restype = i32
    L.wgbsseg_blocks_parse.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, vp, C.POINTER(i64), vp, vp, C.POINTER(i32), C.POINTER(i32)]
    L.wgbsseg_blocks_write_table.restype = i32
    L.wgbsseg_blocks_write_table.argtypes = [C.c_char_p, i32, vp, vp, vp, vp, vp, vp, i64, vp, i64, i64, i32, i32, C.c_char_p, C.c_size_t]
    L.wgbsseg_blocks_write_bedgraph.restype = i32
    L.wgbsseg_blocks_write_bedgraph.argtypes = [C.c_char_p, vp, vp, vp, i64, vp, i32, i32, C.c_char_p, C.c_size_t]
    L.wgbsseg_format_fixed.restype = i64
    L.wgbsseg_format_fixed.argtypes = [vp, i64, i32, vp, i64]
    L.wgbsseg_bed_parse.restype = i32
    L.wgbsseg_bed_parse.argtypes = [vp, i64, i64, C.POINTER(C.c_char_p), i32, vp, vp, vp, vp, vp, vp, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)]
    L.wgbsseg_debug_canonical_float.restype = i64
    L.wgbsseg_debug_canonical_float.argtypes = [C.c_char_p, i64, vp, i64]
    L.wgbsseg_bed_write_annotated.restype = i32
    L.wgbsseg_bed_write_annotated.argtypes = [C.c_char_p, vp, vp, vp, vp, vp, vp, i64, i32, C.c_char_p, C.c_size_t]
    _lib = L
    return L


def load_synth():
    global _synth
    if _synth is None:
        if not op.isfile(SYNTH_LIB_PATH):
            raise NativeLibraryError('%s is missing: build it with `python -m wgbs_tools_amd.build`' % SYNTH_LIB_PATH)
        _share_hip_runtime_with_torch()
        S = C.CDLL(SYNTH_LIB_PATH)
        S.wgbssynth_fill_betas.restype = C.c_int
        S.wgbssynth_fill_betas.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_uint64, C.c_void_p]
        S.wgbssynth_fill_betas_range.restype = C.c_int
        S.wgbssynth_fill_betas_range.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_uint64, C.c_int]
        _synth = S
    return _synth


def device_count():
    return int(load().wgbsseg_device_count())


def _check(rc, buf):
    if rc != OK:
        raise SegmentorError(rc, buf.value.decode(errors='replace'))


class Segmenter:
    """One GPU context: resident betas + loci, batched chunk segmentation (wgbsseg_segment_chunks)."""

    def __init__(self, device=0):
        self._L = load()
        self._h = C.c_void_p()
        self._err = C.create_string_buffer(ERRLEN)
        self._keep = []
        _check(self._L.wgbsseg_create(int(device), C.byref(self._h), self._err, ERRLEN), self._err)
        self.n_sites = 0
        self.n_samples = 0
        self._rbuf = None

    def close(self):
        if self._h:
            self._L.wgbsseg_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs -------------------------------------------------------------------------------------------
    def set_betas(self, samples):
        """samples: list of uint8 arrays shaped [n_sites, 2] (np.memmap of a .beta file works), CLI order."""
        arrs = [np.ascontiguousarray(s, dtype=np.uint8).reshape(-1) for s in samples]
        n2 = arrs[0].size
        assert all(a.size == n2 for a in arrs) and n2 % 2 == 0
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        _check(self._L.wgbsseg_set_betas_host(self._h, ptrs, len(arrs), n2 // 2, self._err, ERRLEN), self._err)
        self.n_sites, self.n_samples = n2 // 2, len(arrs)

    def set_lbetas(self, samples):
        """samples: list of uint16 arrays [n_sites, 2] (.lbeta files); block sums only."""
        arrs = [np.ascontiguousarray(s, dtype=np.uint16).reshape(-1) for s in samples]
        n2 = arrs[0].size
        assert all(a.size == n2 for a in arrs) and n2 % 2 == 0
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        _check(self._L.wgbsseg_set_lbetas_host(self._h, ptrs, len(arrs), n2 // 2, self._err, ERRLEN), self._err)
        self.n_sites, self.n_samples = n2 // 2, len(arrs)

    def set_betas_device(self, data_ptr, n_samples, pitch_bytes, n_sites, keepalive=None):
        """Borrow a device buffer [n_samples][pitch_bytes] (e.g. a torch uint8 tensor's data_ptr())."""
        _check(self._L.wgbsseg_set_betas_device(self._h, C.c_void_p(int(data_ptr)), int(n_samples), int(pitch_bytes),
                                                int(n_sites), self._err, ERRLEN), self._err)
        self._keep = [keepalive]
        self.n_sites, self.n_samples = int(n_sites), int(n_samples)

    def set_loci(self, loci):
        loci = np.ascontiguousarray(loci, dtype=np.uint32)
        _check(self._L.wgbsseg_set_loci_host(self._h, loci.ctypes.data, loci.size, self._err, ERRLEN), self._err)

    def set_site_base(self, base):
        """absolute 0-based index of resident site 0: error messages then name absolute sites"""
        self._L.wgbsseg_set_site_base(self._h, int(base))

    # ---- hot path -----------------------------------------------------------------------------------------
    def segment_chunks(self, start0, lens, pcount, max_cpg, max_bp):
        """-> list of int32 arrays: borders of each chunk relative to its start (first 0, last len)."""
        flat, off = self.segment_chunks_csr(start0, lens, pcount, max_cpg, max_bp)
        return [flat[off[c]:off[c + 1]] for c in range(len(off) - 1)]

    def segment_chunks_csr(self, start0, lens, pcount, max_cpg, max_bp, out=None, off=None):
        """CSR form: (flat int32 borders relative to each chunk's start, int64 offsets [n + 1]).  `out` / `off`: caller's buffers
        (e.g. views into a shared-memory slot of a multi-process run) of at least sum(lens) + n int32 / n + 1 int64."""
        start0 = np.ascontiguousarray(start0, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        n = start0.size
        assert lens.size == n and n >= 1
        cap = int(lens.astype(np.int64).sum()) + n
        if out is None:
            out = np.empty(cap, dtype=np.int32)
        if off is None:
            off = np.empty(n + 1, dtype=np.int64)
        assert out.dtype == np.int32 and out.size >= cap and out.flags.c_contiguous and off.dtype == np.int64 and off.size >= n + 1
        p = Params(float(pcount), int(max_cpg), int(max_bp))
        _check(self._L.wgbsseg_segment_chunks(self._h, start0.ctypes.data, lens.ctypes.data, n, C.byref(p),
                                              out.ctypes.data, cap, off.ctypes.data, self._err, ERRLEN), self._err)
        return out[:off[n]], off

    def segment_regions(self, starts, ends, chunk_size, pcount, max_cpg, max_bp, copy=True):
        """starts/ends: 1-based half-open CpG ranges RELATIVE TO THE RESIDENT DATA (site 1 = first resident site).
        -> (list of arrays: merged absolute border list of each region, stats dict).  copy=True: int64 copies;
        copy=False: int32 views into a buffer that the next call overwrites."""
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        ends = np.ascontiguousarray(ends, dtype=np.int64)
        n = starts.size
        cap = int((ends - starts).sum()) + n
        if self._rbuf is None or self._rbuf.size < cap:       # reused across calls: no 100+ MB allocation per call
            self._rbuf = np.empty(cap, dtype=np.int32)
        out = self._rbuf
        off = np.empty(n + 1, dtype=np.int64)
        stats = np.zeros(8, dtype=np.int64)
        p = Params(float(pcount), int(max_cpg), int(max_bp))
        _check(self._L.wgbsseg_segment_regions(self._h, starts.ctypes.data, ends.ctypes.data, n, int(chunk_size), C.byref(p),
                                               out.ctypes.data, cap, off.ctypes.data, stats.ctypes.data, self._err, ERRLEN),
               self._err)
        res = [out[off[r]:off[r + 1]].astype(np.int64) if copy else out[off[r]:off[r + 1]] for r in range(n)]
        return res, _stats_dict(stats)

    def prefix_sums(self, start0, length):
        out = np.empty((self.n_samples, length + 1, 2), dtype=np.uint32)
        _check(self._L.wgbsseg_prefix_sums(self._h, int(start0), int(length), out.ctypes.data, self._err, ERRLEN), self._err)
        return out

    def scan_only(self, start0, lens, repeat=10, want_carry=False):
        """The scan pass alone: (ms per launch, algorithmic bytes per launch[, carry bytes written per launch when want_carry])."""
        start0 = np.ascontiguousarray(start0, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        ms, nbytes, cbytes = C.c_double(0), C.c_int64(0), C.c_int64(0)
        _check(self._L.wgbsseg_scan_only(self._h, start0.ctypes.data, lens.ctypes.data, start0.size, int(repeat), int(bool(want_carry)),
                                         C.byref(ms), C.byref(nbytes), C.byref(cbytes), self._err, ERRLEN), self._err)
        return (ms.value, nbytes.value, cbytes.value) if want_carry else (ms.value, nbytes.value)

    def block_sums(self, start0, end0, mode=0, min_cov=1):
        """wgbsseg_block_sums over the resident samples: 0-based half-open site ranges -> array [n_samples][n_blocks]
        of uint32 pairs (mode 0), uint8 pairs (1, .bin rows), uint16 pairs (2, .lbeta rows) or float64 means (3)."""
        s = np.ascontiguousarray(start0, dtype=np.int64)
        e = np.ascontiguousarray(end0, dtype=np.int64)
        assert s.shape == e.shape and s.ndim == 1
        n = s.size
        shape, dt = {0: ((self.n_samples, n, 2), np.uint32), 1: ((self.n_samples, n, 2), np.uint8),
                     2: ((self.n_samples, n, 2), np.uint16), 3: ((self.n_samples, n), np.float64)}[int(mode)]
        out = np.empty(shape, dtype=dt)
        _check(self._L.wgbsseg_block_sums(self._h, s.ctypes.data, e.ctypes.data, n, int(mode), int(min_cov), out.ctypes.data,
                                          self._err, ERRLEN), self._err)
        return out

    def convert_regions(self, chrom_lo, chrom_hi, chrom_bp, start, end, slow):
        """wgbsseg_convert_regions against the resident loci -> (startCpG, endCpG) int64 arrays, 0 = NA."""
        a = [np.ascontiguousarray(x, dtype=np.int64) for x in (chrom_lo, chrom_hi, chrom_bp, start, end)]
        sl = np.ascontiguousarray(slow, dtype=np.uint8)
        n = a[0].size
        s = np.zeros(n, dtype=np.int64)
        e = np.zeros(n, dtype=np.int64)
        _check(self._L.wgbsseg_convert_regions(self._h, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data,
                                               a[4].ctypes.data, sl.ctypes.data, n, s.ctypes.data, e.ctypes.data, self._err, ERRLEN),
               self._err)
        return s, e

    def marker_stats(self, tg, bg, n_blocks):
        """wgbsseg_marker_stats over the table of the last mode-3 block_sums call -> float64 [n_blocks, 8]"""
        tg = np.ascontiguousarray(tg, dtype=np.int32)
        bg = np.ascontiguousarray(bg, dtype=np.int32)
        out = np.empty((int(n_blocks), 8), dtype=np.float64)
        _check(self._L.wgbsseg_marker_stats(self._h, tg.ctypes.data, tg.size, bg.ctypes.data, bg.size, int(n_blocks), out.ctypes.data,
                                            self._err, ERRLEN), self._err)
        return out

    def last_block_sums_ms(self):
        return float(self._L.wgbsseg_last_block_sums_ms(self._h))

    def timings(self):
        t = Timings()
        self._L.wgbsseg_get_timings(self._h, C.byref(t))
        return t.as_dict()

    # ---- test hooks ---------------------------------------------------------------------------------------
    def debug_fetch(self, what, dtype, count):
        out = np.empty(count, dtype=dtype)
        got = self._L.wgbsseg_debug_fetch(self._h, what.encode(), out.ctypes.data, out.nbytes)
        if got < 0:
            raise SegmentorError(int(got), 'debug_fetch(%s) failed' % what)
        return out[:got // out.itemsize]

    def debug_sample_terms(self, nmeth, ntotal, pcount):
        nmeth = np.ascontiguousarray(nmeth, dtype=np.float32)
        ntotal = np.ascontiguousarray(ntotal, dtype=np.float32)
        out = np.empty_like(nmeth)
        rc = self._L.wgbsseg_debug_sample_terms(self._h, nmeth.ctypes.data, ntotal.ctypes.data, nmeth.size,
                                                C.c_float(pcount), out.ctypes.data)
        if rc != OK:
            raise SegmentorError(rc, 'debug_sample_terms failed')
        return out

    def debug_check_div(self, pseudo_count, max_total=255 * 60):
        """Operand pairs of a narrow scoring tile for which the short division core is NOT the IEEE quotient (0: usable)."""
        n = C.c_int64(0)
        rc = self._L.wgbsseg_debug_check_div(self._h, C.c_float(pseudo_count), int(max_total), C.byref(n))
        if rc != 0:
            raise SegmentorError(rc, 'debug_check_div failed')
        return int(n.value)

    def debug_div_short(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        out = np.empty(a.size, dtype=np.uint32)
        rc = self._L.wgbsseg_debug_div_short(self._h, a.ctypes.data, b.ctypes.data, a.size, out.ctypes.data)
        if rc != 0:
            raise SegmentorError(rc, 'debug_div_short failed')
        return out

    def debug_div(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        f = np.empty(a.size, dtype=np.uint32)
        g = np.empty(a.size, dtype=np.uint32)
        rc = self._L.wgbsseg_debug_div(self._h, a.ctypes.data, b.ctypes.data, a.size, f.ctypes.data, g.ctypes.data)
        if rc != OK:
            raise SegmentorError(rc, 'debug_div failed')
        return f, g

    def debug_log2(self, first_bits, count, want_f=True, want_d=True, want_fast=False):
        f = np.empty(count, dtype=np.uint32) if want_f else None
        d = np.empty(count, dtype=np.uint64) if want_d else None
        g = np.empty(count, dtype=np.uint64) if want_fast else None
        rc = self._L.wgbsseg_debug_log2(self._h, int(first_bits), int(count), f.ctypes.data if want_f else None,
                                        d.ctypes.data if want_d else None, g.ctypes.data if want_fast else None)
        if rc != OK:
            raise SegmentorError(rc, 'debug_log2 failed')
        return (f, d, g) if want_fast else (f, d)


class PatBeta:
    """wgbsseg_patbeta: pat text -> (#meth, #cov) rows of the CpGs [start_cpg, end_cpg) on one GPU."""

    def __init__(self, start_cpg, end_cpg, device=0):
        self._L = load()
        self._h = C.c_void_p()
        self._err = C.create_string_buffer(ERRLEN)
        self.n = int(end_cpg) - int(start_cpg)
        _check(self._L.wgbsseg_patbeta_create(int(device), int(start_cpg), int(end_cpg), C.byref(self._h), self._err, ERRLEN), self._err)

    def feed(self, text):
        """text: bytes made of whole lines (must end with a newline)"""
        _check(self._L.wgbsseg_patbeta_feed(self._h, text, len(text), self._err, ERRLEN), self._err)

    def kernel_ms(self):
        """device time of the counting kernel over every chunk fed so far (waits for them)"""
        return float(self._L.wgbsseg_patbeta_kernel_ms(self._h))

    def finish(self, lbeta=False):
        out = np.empty((self.n, 2), dtype=np.uint16 if lbeta else np.uint8)
        _check(self._L.wgbsseg_patbeta_finish(self._h, 1 if lbeta else 0, out.ctypes.data, self._err, ERRLEN), self._err)
        return out

    def close(self):
        if self._h:
            self._L.wgbsseg_patbeta_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _stats_dict(stats):
    return dict(chunks=int(stats[0]), patch_dps=int(stats[1]), batches=int(stats[2]), patches_planned=int(stats[3]),
                wall_us=int(stats[4]), first_batch_us=int(stats[5]), later_batches_us=int(stats[6]))


def stitch_regions(regions, chunk_size, engine_many, speculate=True):
    """wgbsseg_stitch_regions: the native chunk grid + pairwise-tree stitching (segment.py:124-135,157-165,199-252) around a
    Python chunk engine.  regions: [(startCpG, endCpG)] 1-based half-open; engine_many([(start, end), ...]) -> list of ABSOLUTE
    border arrays (first start, last end), one per range.  -> (list of int64 border arrays per region, stats dict)."""
    L = load()
    rs = np.ascontiguousarray([r[0] for r in regions], dtype=np.int64)
    re_ = np.ascontiguousarray([r[1] for r in regions], dtype=np.int64)
    n = rs.size
    cap = int((re_ - rs).sum()) + n
    out = np.empty(cap, dtype=np.int32)
    off = np.empty(n + 1, dtype=np.int64)
    stats = np.zeros(8, dtype=np.int64)
    keep, failure = [], []

    def cb(user, starts, ends, cnt, out_ptr, out_cnt):
        try:
            sites = [(int(starts[i]), int(ends[i])) for i in range(cnt)]
            res = engine_many(sites)
            for i, (r, (a, _)) in enumerate(zip(res, sites)):
                rel = np.ascontiguousarray(np.asarray(r, dtype=np.int64) - a, dtype=np.int32)
                keep.append(rel)
                out_ptr[i] = rel.ctypes.data
                out_cnt[i] = rel.size
            return 0
        except BaseException as e:           # never let an exception cross the C frame
            failure.append(e)
            return -1
    err = C.create_string_buffer(ERRLEN)
    rc = L.wgbsseg_stitch_regions(rs.ctypes.data, re_.ctypes.data, n, int(chunk_size), BATCH_FN(cb), None, 1 if speculate else 0,
                                  out.ctypes.data, cap, off.ctypes.data, stats.ctypes.data, err, ERRLEN)
    if failure:
        raise failure[0]
    _check(rc, err)
    return [out[off[r]:off[r + 1]].astype(np.int64) for r in range(n)], _stats_dict(stats)


def first_batch_items(regions, chunk_size, speculate=True):
    """wgbsseg_first_batch_items: the site ranges (1-based half-open) wgbsseg_stitch_regions asks its chunk engine for FIRST over
    these regions — the chunks of the grid, then the junction patches planned up front.  -> (starts, ends int64 arrays, n_chunks)."""
    L = load()
    rs = np.ascontiguousarray([r[0] for r in regions], dtype=np.int64)
    re_ = np.ascontiguousarray([r[1] for r in regions], dtype=np.int64)
    err = C.create_string_buffer(ERRLEN)
    n, nch = C.c_int64(0), C.c_int64(0)
    _check(L.wgbsseg_first_batch_items(rs.ctypes.data, re_.ctypes.data, rs.size, int(chunk_size), 1 if speculate else 0, None, None, 0,
                                       C.byref(n), C.byref(nch), err, ERRLEN), err)
    st, en = np.empty(n.value, dtype=np.int64), np.empty(n.value, dtype=np.int64)
    _check(L.wgbsseg_first_batch_items(rs.ctypes.data, re_.ctypes.data, rs.size, int(chunk_size), 1 if speculate else 0, st.ctypes.data,
                                       en.ctypes.data, st.size, C.byref(n), C.byref(nch), err, ERRLEN), err)
    return st, en, int(nch.value)


def stitch_regions_csr(regions, chunk_size, batch_csr, speculate=True, copy=True):
    """wgbsseg_stitch_regions around an array-level chunk engine (no per-item Python work: the multi-process driver's form).
    batch_csr(starts, ends) — int64 arrays of 1-based half-open ranges — returns (ptr, cnt): uint64 addresses and int64 lengths of
    each range's int32 border list RELATIVE to its start; whatever owns that memory must live until this call returns.
    -> (list of border arrays per region, stats dict)."""
    L = load()
    rs = np.ascontiguousarray([r[0] for r in regions], dtype=np.int64)
    re_ = np.ascontiguousarray([r[1] for r in regions], dtype=np.int64)
    n = rs.size
    cap = int((re_ - rs).sum()) + n
    out = np.empty(cap, dtype=np.int32)
    off = np.empty(n + 1, dtype=np.int64)
    stats = np.zeros(8, dtype=np.int64)
    failure = []

    def cb(user, starts, ends, cnt, out_ptr, out_cnt):
        try:
            st = np.ctypeslib.as_array(starts, shape=(cnt,))
            en = np.ctypeslib.as_array(ends, shape=(cnt,))
            ptr, num = batch_csr(st, en)
            C.memmove(out_ptr, np.ascontiguousarray(ptr, dtype=np.uint64).ctypes.data, 8 * cnt)
            C.memmove(out_cnt, np.ascontiguousarray(num, dtype=np.int64).ctypes.data, 8 * cnt)
            return 0
        except BaseException as e:           # never let an exception cross the C frame
            failure.append(e)
            return -1
    err = C.create_string_buffer(ERRLEN)
    rc = L.wgbsseg_stitch_regions(rs.ctypes.data, re_.ctypes.data, n, int(chunk_size), BATCH_FN(cb), None, 1 if speculate else 0,
                                  out.ctypes.data, cap, off.ctypes.data, stats.ctypes.data, err, ERRLEN)
    if failure:
        raise failure[0]
    _check(rc, err)
    return [out[off[r]:off[r + 1]].astype(np.int64) if copy else out[off[r]:off[r + 1]] for r in range(n)], _stats_dict(stats)


def plan_shares(loci, regions, chunk_size, pcount, max_cpg, max_bp, n_shares, halo=-1, weights=None):
    """wgbsseg_plan_shares[_weighted] (host only): contiguous work-balanced runs of chunks (weights: share d takes weights[d] /
    sum(weights) of the work).  -> dict of int64 arrays [n_shares]: own_lo/own_hi (0-based sites [lo, hi) of the chunks of each
    share), win_lo/win_hi (+- halo), chunks, work."""
    L = load()
    loci = np.ascontiguousarray(loci, dtype=np.uint32)
    rs = np.ascontiguousarray([r[0] for r in regions], dtype=np.int64)
    re_ = np.ascontiguousarray([r[1] for r in regions], dtype=np.int64)
    out = {k: np.zeros(n_shares, dtype=np.int64) for k in ('own_lo', 'own_hi', 'win_lo', 'win_hi', 'chunks', 'work')}
    err = C.create_string_buffer(ERRLEN)
    p = Params(float(pcount), int(max_cpg), int(max_bp))
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    assert w is None or w.size == n_shares
    _check(L.wgbsseg_plan_shares_weighted(loci.ctypes.data, loci.size, rs.ctypes.data, re_.ctypes.data, rs.size, int(chunk_size), C.byref(p),
                                          int(n_shares), None if w is None else w.ctypes.data, int(halo), out['own_lo'].ctypes.data,
                                          out['own_hi'].ctypes.data, out['win_lo'].ctypes.data, out['win_hi'].ctypes.data,
                                          out['chunks'].ctypes.data, out['work'].ctypes.data, err, ERRLEN), err)
    return out


class SegmenterGroup:
    """wgbsseg_group: one context per share (GPU), contiguous work-balanced shares of the chunk grid, one host-side tree."""

    def __init__(self, devices):
        self._L = load()
        self._h = C.c_void_p()
        self._err = C.create_string_buffer(ERRLEN)
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        _check(self._L.wgbsseg_group_create(dv.ctypes.data, dv.size, C.byref(self._h), self._err, ERRLEN), self._err)
        self.n_shares = int(dv.size)
        self.devices = [int(d) for d in dv]
        self._keep = []
        self._rbuf = None
        self.last_csr = None          # (flat int32 borders, int64 offsets) of the last segment_regions call
        self._cap = 0
        self.n_regions = 0

    def close(self):
        if self._h:
            self._L.wgbsseg_group_destroy(self._h)
            self._h = C.c_void_p()
            self._keep = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def plan(self, loci, regions, chunk_size, pcount, max_cpg, max_bp, halo=-1):
        """regions: [(startCpG, endCpG)] ABSOLUTE 1-based half-open, ascending; loci: whole-genome uint32.
        -> dict(win_lo, win_hi, chunks, work): per-share 0-based resident windows and load."""
        loci = np.ascontiguousarray(loci, dtype=np.uint32)
        rs = np.ascontiguousarray([r[0] for r in regions], dtype=np.int64)
        re_ = np.ascontiguousarray([r[1] for r in regions], dtype=np.int64)
        G = self.n_shares
        lo, hi, ch, wk = (np.zeros(G, dtype=np.int64) for _ in range(4))
        p = Params(float(pcount), int(max_cpg), int(max_bp))
        _check(self._L.wgbsseg_group_plan(self._h, loci.ctypes.data, loci.size, rs.ctypes.data, re_.ctypes.data, rs.size, int(chunk_size),
                                          C.byref(p), int(halo), lo.ctypes.data, hi.ctypes.data, ch.ctypes.data, wk.ctypes.data,
                                          self._err, ERRLEN), self._err)
        self.n_regions = int(rs.size)
        self._cap = int((re_ - rs).sum()) + self.n_regions
        self.n_sites = int(loci.size)
        self.windows = dict(win_lo=lo, win_hi=hi, chunks=ch, work=wk)
        return self.windows

    def load_host(self, samples, wait=True):
        """samples: whole-genome uint8 arrays (np.memmap of .beta files work); every share uploads only its window.
        wait=False: the upload runs in the background and segment_regions() starts on what has arrived (the arrays are kept
        alive here until then)."""
        arrs = [s if isinstance(s, np.ndarray) and s.dtype == np.uint8 and s.ndim == 1 else np.ascontiguousarray(s, dtype=np.uint8).reshape(-1)
                for s in samples]
        assert all(a.size == 2 * self.n_sites for a in arrs), 'every sample must hold 2 bytes per site of the planned genome'
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        if wait:
            _check(self._L.wgbsseg_group_load_host(self._h, ptrs, len(arrs), self.n_sites, self._err, ERRLEN), self._err)
        else:
            self._streaming = arrs
            _check(self._L.wgbsseg_group_load_host_async(self._h, ptrs, len(arrs), self.n_sites, self._err, ERRLEN), self._err)

    def load_wait(self):
        _check(self._L.wgbsseg_group_load_wait(self._h, self._err, ERRLEN), self._err)
        self._streaming = None

    def share_set_device(self, share, data_ptr, n_samples, pitch_bytes, keepalive=None):
        _check(self._L.wgbsseg_group_share_set_device(self._h, int(share), C.c_void_p(int(data_ptr)), int(n_samples), int(pitch_bytes),
                                                      self._err, ERRLEN), self._err)
        self._keep.append(keepalive)

    def segment_regions(self, copy=True):
        if self._rbuf is None or self._rbuf.size < self._cap:
            self._rbuf = np.empty(self._cap, dtype=np.int32)
        out = self._rbuf
        n = self.n_regions
        off = np.empty(n + 1, dtype=np.int64)
        stats = np.zeros(8, dtype=np.int64)
        try:
            _check(self._L.wgbsseg_group_segment_regions(self._h, out.ctypes.data, out.size, off.ctypes.data, stats.ctypes.data,
                                                         self._err, ERRLEN), self._err)
        finally:
            if getattr(self, '_streaming', None) is not None:
                self._L.wgbsseg_group_load_wait(self._h, None, 0)      # (already collected on success; a failed call may have left uploaders running)
                self._streaming = None
        self.last_csr = (out, off)                # the same lists as one CSR (views into the buffer the next call overwrites)
        res = [out[off[r]:off[r + 1]].astype(np.int64) if copy else out[off[r]:off[r + 1]] for r in range(n)]
        return res, _stats_dict(stats)

    def segment_region_range(self, first, end, cap):
        """wgbsseg_group_segment_region_range: the planned regions [first, end) only -> (flat int32 absolute borders, off int64 [end - first + 1], stats).
        The CSR lives in a buffer of its OWN (cap ints: the slice's sites + regions suffice): a caller writes one slice's BED rows while the next
        slice is being segmented.  Slices in ascending order; the uploaders of a streaming load are collected with the last one."""
        n = int(end) - int(first)
        out = np.empty(int(cap), dtype=np.int32)
        off = np.empty(n + 1, dtype=np.int64)
        stats = np.zeros(8, dtype=np.int64)
        try:
            _check(self._L.wgbsseg_group_segment_region_range(self._h, int(first), int(end), out.ctypes.data, out.size, off.ctypes.data, stats.ctypes.data,
                                                              self._err, ERRLEN), self._err)
        except BaseException:
            if getattr(self, '_streaming', None) is not None:
                self._L.wgbsseg_group_load_wait(self._h, None, 0)
                self._streaming = None
            raise
        if int(end) == self.n_regions and getattr(self, '_streaming', None) is not None:
            self._L.wgbsseg_group_load_wait(self._h, None, 0)
            self._streaming = None
        return out, off, _stats_dict(stats)

    def timings(self, share):
        t = Timings()
        self._L.wgbsseg_group_get_timings(self._h, int(share), C.byref(t))
        return t.as_dict()


def segment_chunks_host(samples, loci, start0, lens, pcount, max_cpg, max_bp, device=0):
    """One-shot form over wgbsseg_segment_chunks_host (host buffers in, borders out)."""
    L = load()
    arr = np.ascontiguousarray(np.stack([np.asarray(s, dtype=np.uint8).reshape(-1) for s in samples]))
    n_samples, pitch = arr.shape
    loci = np.ascontiguousarray(loci, dtype=np.uint32)
    start0 = np.ascontiguousarray(start0, dtype=np.int64)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    n = start0.size
    cap = int(lens.astype(np.int64).sum()) + n
    out = np.empty(cap, dtype=np.int32)
    off = np.empty(n + 1, dtype=np.int64)
    err = C.create_string_buffer(ERRLEN)
    p = Params(float(pcount), int(max_cpg), int(max_bp))
    _check(L.wgbsseg_segment_chunks_host(arr.ctypes.data, n_samples, pitch, pitch // 2, loci.ctypes.data,
                                         start0.ctypes.data, lens.ctypes.data, n, C.byref(p), int(device),
                                         out.ctypes.data, cap, off.ctypes.data, err, ERRLEN), err)
    return [out[off[c]:off[c + 1]].copy() for c in range(n)]


def add_loci(loci, chrom_names, chrom_cum, start_cpg, end_cpg, path=None, append=False, threads=0):
    """wgbsseg_add_loci: BED rows of the blocks appended to `path` (None: the process's stdout)."""
    L = load()
    loci = np.ascontiguousarray(loci, dtype=np.uint32)
    cum = np.ascontiguousarray(chrom_cum, dtype=np.int64)
    s = np.ascontiguousarray(start_cpg, dtype=np.int64)
    e = np.ascontiguousarray(end_cpg, dtype=np.int64)
    names = (C.c_char_p * len(chrom_names))(*[str(n).encode() for n in chrom_names])
    err = C.create_string_buffer(ERRLEN)
    rc = L.wgbsseg_add_loci(loci.ctypes.data, loci.size, cum.ctypes.data, names, len(chrom_names), s.ctypes.data, e.ctypes.data,
                            s.size, None if path is None else os.fsencode(path), 1 if append else 0, int(threads), err, ERRLEN)
    _check(rc, err)


def add_loci_borders(loci, chrom_names, chrom_cum, flat, off, min_cpg=1, path=None, append=False, threads=0):
    """wgbsseg_add_loci_borders: the BED rows of a segmentation straight from its merged border lists (CSR: region r's ascending 1-based
    borders are flat[off[r]:off[r+1]], int32) -> (rows written, blocks dropped as shorter than min_cpg)."""
    L = load()
    loci = np.ascontiguousarray(loci, dtype=np.uint32)
    cum = np.ascontiguousarray(chrom_cum, dtype=np.int64)
    flat = np.ascontiguousarray(flat, dtype=np.int32)
    off = np.ascontiguousarray(off, dtype=np.int64)
    names = (C.c_char_p * len(chrom_names))(*[str(n).encode() for n in chrom_names])
    err = C.create_string_buffer(ERRLEN)
    nw, nd = C.c_int64(0), C.c_int64(0)
    rc = L.wgbsseg_add_loci_borders(loci.ctypes.data, loci.size, cum.ctypes.data, names, len(chrom_names), flat.ctypes.data, off.ctypes.data,
                                    off.size - 1, int(min_cpg), None if path is None else os.fsencode(path), 1 if append else 0, int(threads),
                                    C.byref(nw), C.byref(nd), err, ERRLEN)
    _check(rc, err)
    return nw.value, nd.value


# ------------------------------------------------------------------------------------------------------------
# text of the block tools (include/wgbsseg.h: wgbsseg_blocks_*; host side, no device)
# ------------------------------------------------------------------------------------------------------------
class ParsedBlocks:
    """What wgbsseg_blocks_parse leaves: the table's bytes and, per row, where it begins, how long its `chr \\t start \\t end`
    text is, the two CpG columns (int64, 0 where missing) and the missing flags; bp_start / bp_end (the second and third fields as
    integers) when every row's are plain digits, else None; first_fields = fields of the first non-comment line (at most 7)."""

    def __init__(self, text, line_off, len3, start_cpg, end_cpg, na, bp_start=None, bp_end=None, first_fields=0, data=None):
        self.text, self.line_off, self.len3, self.start_cpg, self.end_cpg, self.na = text, line_off, len3, start_cpg, end_cpg, na
        self.bp_start, self.bp_end, self.first_fields, self.data = bp_start, bp_end, first_fields, data

    def __len__(self):
        return self.line_off.size

    def take(self, idx):
        """rows idx (a slice or an index array) as a new ParsedBlocks over the same bytes"""
        cut = (lambda a: None if a is None else a[idx])
        return ParsedBlocks(self.text, self.line_off[idx], self.len3[idx], self.start_cpg[idx], self.end_cpg[idx], self.na[idx],
                            cut(self.bp_start), cut(self.bp_end), self.first_fields, self.data)

    def rows(self, a, b):
        return self.take(slice(a, b))

    def coords(self, idx=None):
        """-> (chr, start, end) lists of str: the first three fields of every row (of the rows idx), as the file has them"""
        t = self.text
        lo, l3 = (self.line_off, self.len3) if idx is None else (self.line_off[idx], self.len3[idx])
        parts = [bytes(t[o:o + l]).decode('ascii').split('\t') for o, l in zip(lo.tolist(), l3.tolist())]
        return [p[0] for p in parts], [p[1] for p in parts], [p[2] for p in parts]

    def fields(self, idx, first, count):
        """-> `count` lists of str: fields first .. first+count-1 of the rows idx ('' where a row has fewer)"""
        data = self.data if self.data is not None else self.text.tobytes()
        cols = [[] for _ in range(count)]
        for o in (self.line_off if idx is None else self.line_off[idx]).tolist():
            e = data.find(b'\n', o)
            tok = data[o:(len(data) if e < 0 else e)].decode('ascii').split('\t')
            for c in range(count):
                cols[c].append(tok[first + c] if len(tok) > first + c else '')
        return cols

    def _ptrs(self):
        self.line_off = np.ascontiguousarray(self.line_off, dtype=np.int64)
        self.len3 = np.ascontiguousarray(self.len3, dtype=np.int32)
        self.start_cpg = np.ascontiguousarray(self.start_cpg, dtype=np.int64)
        self.end_cpg = np.ascontiguousarray(self.end_cpg, dtype=np.int64)
        self.na = np.ascontiguousarray(self.na, dtype=np.uint8)
        return (self.text.ctypes.data, self.line_off.ctypes.data, self.len3.ctypes.data, self.start_cpg.ctypes.data,
                self.end_cpg.ctypes.data, self.na.ctypes.data)


def blocks_parse(data, max_rows=None):
    """wgbsseg_blocks_parse on the bytes of a blocks table -> ParsedBlocks, or None when the text is not a plain table (the
    caller then parses it line by line)."""
    L = load()
    text = np.frombuffer(data, dtype=np.uint8)
    if text.size == 0:
        return None
    cap = (data.count(b'\n') if isinstance(data, (bytes, bytearray)) else int(np.count_nonzero(text == 10))) + 1
    if max_rows is not None:
        cap = min(cap, max(int(max_rows), 0) + 1)
    line_off = np.empty(cap, dtype=np.int64)
    len3 = np.empty(cap, dtype=np.int32)
    s = np.empty(cap, dtype=np.int64)
    e = np.empty(cap, dtype=np.int64)
    na = np.empty(cap, dtype=np.uint8)
    b0 = np.empty(cap, dtype=np.int64)
    b1 = np.empty(cap, dtype=np.int64)
    n = C.c_int64(0)
    ok = C.c_int32(0)
    ff = C.c_int32(0)
    rc = L.wgbsseg_blocks_parse(text.ctypes.data, text.size, -1 if max_rows is None else int(max_rows), cap, line_off.ctypes.data,
                                len3.ctypes.data, s.ctypes.data, e.ctypes.data, na.ctypes.data, C.byref(n), b0.ctypes.data, b1.ctypes.data,
                                C.byref(ok), C.byref(ff))
    if rc != OK:
        return None
    k = int(n.value)
    return ParsedBlocks(text, line_off[:k], len3[:k], s[:k], e[:k], na[:k], b0[:k] if ok.value else None, b1[:k] if ok.value else None,
                        int(ff.value), data if isinstance(data, bytes) else None)


def blocks_write_table(path, parsed, values, digits, append=True, threads=0):
    """wgbsseg_blocks_write_table: the rows of `parsed` with values[r][c] as %.<digits>f (NaN: NA), appended to `path` (None: the
    process's standard output; flush sys.stdout first)."""
    L = load()
    v = np.ascontiguousarray(values, dtype=np.float64)
    if v.ndim != 2 or v.shape[0] != len(parsed):
        raise ValueError('values must be [rows][columns]')
    t, lo, l3, s, e, na = parsed._ptrs()
    err = C.create_string_buffer(ERRLEN)
    rc = L.wgbsseg_blocks_write_table(None if path is None else os.fsencode(path), 1 if append else 0, t, lo, l3, s, e, na, len(parsed), v.ctypes.data,
                                      v.shape[1], v.shape[1], int(digits), int(threads), err, ERRLEN)
    _check(rc, err)


def blocks_write_bedgraph(path, parsed, rows, threads=0):
    """wgbsseg_blocks_write_bedgraph: chr, start, end, meth / cov (%.2f, -1 for 0 / 0), cov from uint8 / uint16 pairs."""
    L = load()
    rows = np.ascontiguousarray(rows)
    if rows.dtype not in (np.uint8, np.uint16) or rows.shape != (len(parsed), 2):
        raise ValueError('rows must be uint8 / uint16 [n][2]')
    t, lo, l3, _, _, _ = parsed._ptrs()
    err = C.create_string_buffer(ERRLEN)
    rc = L.wgbsseg_blocks_write_bedgraph(os.fsencode(path), t, lo, l3, len(parsed), rows.ctypes.data, 1 if rows.dtype == np.uint16 else 0,
                                         int(threads), err, ERRLEN)
    _check(rc, err)


def format_fixed(values, digits):
    """wgbsseg_format_fixed -> list of str: printf('%.<digits>f') of every value (NaN: 'NA')"""
    L = load()
    v = np.ascontiguousarray(values, dtype=np.float64).ravel()
    out = np.empty(v.size * 24 + 1024 + 420 * int(np.count_nonzero(~((v >= 0) & (v <= 1)))), dtype=np.uint8)
    n = L.wgbsseg_format_fixed(v.ctypes.data, v.size, int(digits), out.ctypes.data, out.size)
    if n < 0:
        raise ValueError('format_fixed: buffer too small')
    return out[:n].tobytes().decode('ascii').split('\n')[:-1]


class ParsedBed:
    """What wgbsseg_bed_parse leaves: the table's bytes and, per row, its offset, the length of `chr \\t start \\t end` and of the
    whole row, the chromosome's index in the names given (-1: unknown), start and end."""

    def __init__(self, text, line_off, len3, row_len, chrom_idx, start, end, width, header=False):
        self.text, self.line_off, self.len3, self.row_len = text, line_off, len3, row_len
        self.chrom_idx, self.start, self.end, self.width, self.header = chrom_idx, start, end, width, header

    def __len__(self):
        return self.line_off.size


def bed_parse(data, chrom_names):
    """wgbsseg_bed_parse -> ParsedBed, or None when the table is not one whose rows can be written back verbatim."""
    L = load()
    text = np.frombuffer(data, dtype=np.uint8)
    if text.size == 0:
        return None
    cap = (data.count(b'\n') if isinstance(data, (bytes, bytearray)) else int(np.count_nonzero(text == 10))) + 1
    line_off = np.empty(cap, dtype=np.int64)
    len3 = np.empty(cap, dtype=np.int32)
    row_len = np.empty(cap, dtype=np.int32)
    chrom = np.empty(cap, dtype=np.int32)
    start = np.empty(cap, dtype=np.int64)
    end = np.empty(cap, dtype=np.int64)
    names = (C.c_char_p * len(chrom_names))(*[str(n).encode() for n in chrom_names])
    n = C.c_int64(0)
    w = C.c_int32(0)
    h = C.c_int32(0)
    rc = L.wgbsseg_bed_parse(text.ctypes.data, text.size, cap, names, len(chrom_names), line_off.ctypes.data, len3.ctypes.data,
                             row_len.ctypes.data, chrom.ctypes.data, start.ctypes.data, end.ctypes.data, C.byref(n), C.byref(w), C.byref(h))
    if rc != OK:
        return None
    k = int(n.value)
    return ParsedBed(text, line_off[:k], len3[:k], row_len[:k], chrom[:k], start[:k], end[:k], int(w.value), bool(h.value))


def bed_write_annotated(path, parsed, start_cpg, end_cpg, keep=None, threads=0):
    """wgbsseg_bed_write_annotated: the rows of `parsed` (those with keep[r], when given) with the two CpG columns, to `path` (None:
    the process's standard output; flush sys.stdout first)."""
    L = load()
    sel = (lambda a, dt: np.ascontiguousarray(a if keep is None else a[keep], dtype=dt))
    lo, l3, rl = sel(parsed.line_off, np.int64), sel(parsed.len3, np.int32), sel(parsed.row_len, np.int32)
    s, e = sel(start_cpg, np.int64), sel(end_cpg, np.int64)
    err = C.create_string_buffer(ERRLEN)
    rc = L.wgbsseg_bed_write_annotated(None if path is None else os.fsencode(path), parsed.text.ctypes.data, lo.ctypes.data, l3.ctypes.data,
                                       rl.ctypes.data, s.ctypes.data, e.ctypes.data, lo.size, int(threads), err, ERRLEN)
    _check(rc, err)


def debug_canonical_float(tokens):
    """wgbsseg_debug_canonical_float -> bool array: which tokens are decimals that print as they read through a float"""
    L = load()
    data = '\n'.join(tokens).encode('ascii')
    out = np.zeros(len(tokens) + 1, dtype=np.uint8)
    n = L.wgbsseg_debug_canonical_float(data, len(data), out.ctypes.data, out.size)
    if n < 0:
        raise ValueError('debug_canonical_float failed')
    return out[:len(tokens)].astype(bool)
