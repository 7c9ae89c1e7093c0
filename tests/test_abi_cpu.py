"""The drop-in boundary without a GPU: libwgbsseg.so loads, exports every symbol include/wgbsseg.h declares (and
nothing is declared that is not exported), and refuses — loudly, no CPU fallback — to work without a device."""
import ctypes as C
import os.path as op
import re

import pytest

from wgbs_tools_amd import _lib

ROOT = op.dirname(op.dirname(op.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    if not op.isfile(_lib.LIB_PATH):
        from wgbs_tools_amd import build
        build.build()
    return _lib.load()


def test_header_and_library_agree(lib):
    hdr = open(op.join(ROOT, 'include', 'wgbsseg.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = sorted(set(re.findall(r'\b(wgbsseg_[a-z0-9_]+)\s*\(', hdr)))
    assert declared == sorted(_lib.EXPORTS)
    raw = C.CDLL(_lib.LIB_PATH)
    for sym in declared:
        assert hasattr(raw, sym), 'libwgbsseg.so does not export %s' % sym
    assert lib.wgbsseg_version() == int(re.search(r'#define WGBSSEG_VERSION (\d+)', hdr).group(1)) == _lib.ABI_VERSION


def test_binding_refuses_a_library_of_another_abi_version(lib, monkeypatch):
    """Argument lists differ between ABI versions (wgbsseg_scan_only, 200 -> 210): load() compares wgbsseg_version() with the version its
    prototypes describe and refuses anything else."""
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'ABI_VERSION', _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.NativeLibraryError, match='ABI version'):
        _lib.load()


def test_error_codes_match_header():
    hdr = open(op.join(ROOT, 'include', 'wgbsseg.h')).read()
    codes = dict(re.findall(r'#define (WGBSSEG_(?:OK|E_[A-Z_]+))\s+(-?\d+)', hdr))
    assert int(codes['WGBSSEG_OK']) == _lib.OK and int(codes['WGBSSEG_E_METH_GT_COV']) == _lib.E_METH_GT_COV
    assert int(codes['WGBSSEG_E_HIP']) == _lib.E_HIP and int(codes['WGBSSEG_E_LOCI_ORDER']) == _lib.E_LOCI_ORDER
    assert int(codes['WGBSSEG_E_CAPACITY']) == _lib.E_CAPACITY and int(codes['WGBSSEG_E_STATE']) == _lib.E_STATE


def test_no_device_means_loud_failure(lib):
    if lib.wgbsseg_device_count() > 0:
        pytest.skip('a GPU is visible here')
    with pytest.raises(_lib.SegmentorError) as e:
        _lib.Segmenter(0)
    assert e.value.code == _lib.E_HIP and 'no CPU fallback' in e.value.msg


def test_product_code_never_touches_the_oracle():
    """wgbs_tools_amd/ must not import, link or execute anything under oracle/."""
    import glob
    for path in glob.glob(op.join(ROOT, 'wgbs_tools_amd', '**', '*'), recursive=True):
        if op.isfile(path) and path.endswith(('.py', '.hip', '.h', '.cpp')):
            src = open(path, errors='replace').read()
            assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), path
            assert 'liboracle' not in src and 'oracle/' not in src.replace('oracle/segment_oracle.c', '').replace('oracle/libm_probe.c', ''), path
