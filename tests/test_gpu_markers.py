"""find_markers on the GPU (block reduction mode 3 + k_marker_stats behind wgbsseg_marker_stats) against the files captured from
the reference's own find_markers.py, and the statistics kernel against numpy."""
import contextlib
import io
import os.path as op

import numpy as np
import pytest

from wgbs_tools_amd import _lib, synth, wgbs_tools
from test_markers_cpu import mworld, OracleMarkerEngine          # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['default', 'hypo_top', 'hyper_quants', 'two_targets_bg', 'mw_test', 'mvalue_test', 'single_sample_target'])
def test_cli_matches_reference(mworld, name, tmp_path):
    rec = mworld['g']['cases'][name]
    od = str(tmp_path / 'out')
    argv = ['wgbstools', 'find_markers', '-b', mworld['blocks'], '-g', mworld['groups'], '--betas'] + mworld['betas'] + ['-o', od] + rec['args']
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        assert wgbs_tools.main(argv) == 0
    for fname, want in rec['files'].items():
        assert open(op.join(od, fname)).read() == want, (name, fname)
    assert err.getvalue().replace(od, '<OUT>') == rec['stderr'].replace('<TMP>/out_' + name, '<OUT>')
    assert op.isfile(op.join(od, 'params.txt'))


def test_marker_stats_kernel_against_numpy():
    """300,000 blocks x 24 samples with missing values (also whole sets missing): counts, sequential sums, minima, maxima bit for bit"""
    n, N = 3000000, 24
    data = [synth.synth_betas(31, s, 0, n) for s in range(N)]
    for s in (3, 4, 5):
        data[s][100000:200000, 1] = 0
        data[s][100000:200000, 0] = 0
    b = np.arange(0, n + 10, 10, dtype=np.int64).clip(max=n)
    s0, e0 = b[:-1], b[1:]
    with _lib.Segmenter(0) as sg:
        sg.set_betas(data)
        table = sg.block_sums(s0, e0, mode=3, min_cov=40)
        tg, bg = [3, 4, 5], [0, 7, 1, 23, 11, 2, 9]
        got = sg.marker_stats(tg, bg, s0.size)
        ms = sg.last_block_sums_ms()
        with pytest.raises(_lib.SegmentorError, match='not a mode-3 reduction'):
            sg.marker_stats(tg, bg, s0.size - 1)
        sg.block_sums(s0[:5], e0[:5], mode=1)
        with pytest.raises(_lib.SegmentorError, match='not a mode-3 reduction'):
            sg.marker_stats(tg, bg, 5)
    eng = OracleMarkerEngine.__new__(OracleMarkerEngine)
    eng.table = table
    want = eng.marker_stats(tg, bg, s0.size)
    assert np.isnan(got[:, 2]).sum() > 5000                            # blocks where no target sample has coverage
    assert np.array_equal(got.view(np.uint64) == want.view(np.uint64), np.ones_like(got, dtype=bool)) or \
        (np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)]))
    print('k_marker_stats: %.3f ms for %d blocks x %d samples' % (ms, s0.size, len(tg) + len(bg)))
