"""Block reduction on the GPU (k_block_sums behind wgbsseg_block_sums; the beta_to_blocks / beta_to_table mirrors)
against the vectors captured from the reference's Python and against the oracle."""
import base64
import hashlib
import json
import os
import os.path as op

import numpy as np
import pytest

from oracle import block_sums as OB
import cases
from wgbs_tools_amd import _lib, synth, wgbs_tools
from test_blocks_cpu import world          # noqa: F401  (fixture: betas, blocks tables and goldens on disk)

pytestmark = pytest.mark.gpu


def _sha(text):
    return hashlib.sha1(text.encode()).hexdigest()


@pytest.mark.parametrize('name', ['nice', 'ragged'])
def test_cli_outputs_match_reference(world, name, tmp_path):
    """`wgbstools beta_to_blocks` (.bin, .lbeta, .bedGraph) and `wgbstools beta_to_table` byte for byte."""
    rec = world['g']['tables'][name]
    for lbeta in (False, True):
        od = tmp_path / ('o%d' % lbeta)
        od.mkdir()
        argv = ['wgbstools', 'beta_to_blocks'] + world['betas'] + ['-b', world['blocks'][name], '-o', str(od), '--bedGraph'] + (['-l'] if lbeta else [])
        assert wgbs_tools.main(argv) == 0
        for b in world['betas']:
            key = op.basename(b)
            stem = str(od / op.splitext(key)[0])
            got = open(stem + ('.lbeta' if lbeta else '.bin'), 'rb').read()
            assert got == base64.b64decode(rec['lbeta' if lbeta else 'bin'][key]), (name, key, lbeta)
            if not lbeta:
                bg = open(stem + '.bedGraph').read()
                assert bg[:600] == rec['bedgraph'][key]['head'] and _sha(bg) == rec['bedgraph'][key]['sha1']
    for tag, extra in (('table_plain', ['-c', '4', '--digits', '2']), ('table_groups', ['-g', world['groups'], '-c', '10', '--digits', '3', '--chunk_size', '257'])):
        out = str(tmp_path / (tag + '.tsv'))
        assert wgbs_tools.main(['wgbstools', 'beta_to_table', world['blocks'][name], '--betas'] + world['betas'] + ['-o', out] + extra) == 0
        text = open(out).read()
        assert text[:600] == rec[tag]['head'], (name, tag)
        assert len(text) == rec[tag]['len'] and _sha(text) == rec[tag]['sha1']
    # uint16 .lbeta INPUT files: .bin / .lbeta bytes and the table, against what the reference made of them
    lrec = rec['lbeta_inputs']
    for lbeta in (False, True):
        od = tmp_path / ('lo%d' % lbeta)
        od.mkdir()
        argv = ['wgbstools', 'beta_to_blocks'] + world['lbetas'] + ['-b', world['blocks'][name], '-o', str(od)] + (['-l'] if lbeta else [])
        assert wgbs_tools.main(argv) == 0
        for b in world['lbetas']:
            key = op.basename(b)
            got = open(str(od / op.splitext(key)[0]) + ('.lbeta' if lbeta else '.bin'), 'rb').read()
            assert hashlib.sha1(got).hexdigest() == lrec['lbeta_sha1' if lbeta else 'bin_sha1'][key], (name, key, lbeta)
    out = str(tmp_path / 'ltable.tsv')
    assert wgbs_tools.main(['wgbstools', 'beta_to_table', world['blocks'][name], '--betas'] + world['lbetas'] + ['-o', out, '-c', '4', '--digits', '3']) == 0
    text = open(out).read()
    assert text[:600] == lrec['table_plain']['head'] and _sha(text) == lrec['table_plain']['sha1']
    # second run without --force skips existing files (beta_to_blocks.py:168-178)
    od = tmp_path / 'o0'
    before = {f: os.stat(od / f).st_mtime_ns for f in os.listdir(od)}
    assert wgbs_tools.main(['wgbstools', 'beta_to_blocks'] + world['betas'] + ['-b', world['blocks'][name], '-o', str(od)]) == 0
    assert before == {f: os.stat(od / f).st_mtime_ns for f in os.listdir(od)}


def test_block_sums_all_modes_against_oracle():
    """Random tables on 1.2 M sites x 5 samples: unaligned edges, 1-site blocks, blocks of thousands of sites, empty
    blocks, overlaps, the whole range; every mode bit for bit (mode 3: the doubles' bits, NaNs included)."""
    n, N = 1200000, 5
    seed = 4242
    data = [synth.synth_betas(seed, s, 0, n) for s in range(N)]
    data[3][5000:9000, :] = 255                                   # saturated stretch: sums that need the uint16 / uint8 trim
    data[4][100000:100700, :] = 0
    rng = np.random.default_rng(1)
    s0 = rng.integers(0, n - 1, 150000)
    ln = np.where(rng.random(150000) < 0.95, rng.integers(0, 40, 150000), rng.integers(40, 6000, 150000))
    e0 = np.minimum(s0 + ln, n)
    s0 = np.concatenate([s0, [0, n - 1, n, 0, 7, 8, 9]]); e0 = np.concatenate([e0, [n, n, n, 1, 8, 8, 25]])
    sg = _lib.Segmenter(0)
    try:
        sg.set_betas(data)
        raw = sg.block_sums(s0, e0, mode=0)
        b8 = sg.block_sums(s0, e0, mode=1)
        b16 = sg.block_sums(s0, e0, mode=2)
        mean = sg.block_sums(s0, e0, mode=3, min_cov=7)
        assert sg.last_block_sums_ms() > 0
        for s in range(N):
            want = OB.block_sums(data[s], s0, e0)
            assert (raw[s].astype(np.int64) == want).all(), 'sample %d: first bad block %d' % (s, int(np.flatnonzero((raw[s] != want).any(1))[0]))
            assert (b8[s] == OB.trim(want, False)).all() and (b16[s] == OB.trim(want, True)).all()
            assert (mean[s].view(np.uint64) == OB.beta2vec(want, 7).view(np.uint64)).all() or \
                   (np.isnan(mean[s]) == np.isnan(OB.beta2vec(want, 7))).all() and np.array_equal(mean[s][~np.isnan(mean[s])], OB.beta2vec(want, 7)[~np.isnan(mean[s])])
        with pytest.raises(_lib.SegmentorError):
            sg.block_sums([5], [n + 1])
        with pytest.raises(_lib.SegmentorError):
            sg.block_sums([9], [3])
    finally:
        sg.close()


@pytest.mark.parametrize('general', [0, 1])
def test_block_sums_of_ordered_tables_against_oracle(general, monkeypatch):
    """Tables ordered by first and last site (what a segmentation writes) go through the streaming kernel: tiles of 1024
    sites, runs of 8 tiles, a two-tile ring of prefixes.  Blocks of one site, empty blocks, blocks across tile and run
    boundaries, blocks longer than a tile and longer than a run, gaps, overlaps that keep the order, a row length that is no
    multiple of anything, the rows shuffled (the host sorts; results go back to the caller's rows); all four modes, and the
    general kernel forced onto the same tables."""
    if general: monkeypatch.setenv('WGBSSEG_BLOCK_SUMS_GENERAL', '1')
    n, N = 1000003, 6
    data = [synth.synth_betas(777, s, 0, n) for s in range(N)]
    data[2][40000:47000, :] = 255                                  # saturated: the trims of modes 1 and 2
    data[5][8000:8400, :] = 0
    rng = np.random.default_rng(8)
    tables = {}
    cuts = np.unique(np.concatenate([rng.integers(0, n, 90000), [0, n], np.arange(1024, n, 1024)[::7], np.arange(8192, n, 8192)[::3] + 1]))
    tables['tiling'] = (cuts[:-1], cuts[1:])
    st = np.sort(rng.integers(0, n - 3000, 30000))
    ln = np.where(rng.random(30000) < 0.9, rng.integers(0, 30, 30000), rng.integers(30, 2500, 30000))
    en = np.maximum.accumulate(np.minimum(st + ln, n))             # ends forced into order: overlaps, nesting never
    en = np.maximum(en, st)
    tables['gaps_and_overlaps'] = (st, en)
    tables['long'] = (np.array([5, 1000, 3000, 20000, 20000, 50000, 700000, n - 1, n]), np.array([1000, 3000, 20000, 20000, 50000, 700000, n, n, n]))
    edge = np.array([0, 1023, 1024, 1025, 2047, 8191, 8192, 8193, 16383, 16384, n - 17, n - 1])
    tables['edges'] = (edge[:-1], edge[1:])
    tables['single_tile_late'] = (np.array([900000, 900010]), np.array([900010, 900500]))
    with _lib.Segmenter(0) as sg:
        sg.set_betas(data)
        for name, (s0, e0) in tables.items():
            s0 = np.asarray(s0, dtype=np.int64); e0 = np.asarray(e0, dtype=np.int64)
            for shuffled in (False, True):
                if shuffled:
                    o = rng.permutation(s0.size)
                    s0, e0 = s0[o], e0[o]
                raw = sg.block_sums(s0, e0, mode=0)
                b8 = sg.block_sums(s0, e0, mode=1)
                b16 = sg.block_sums(s0, e0, mode=2)
                mean = sg.block_sums(s0, e0, mode=3, min_cov=9)
                for s in range(N):
                    want = OB.block_sums(data[s], s0, e0)
                    bad = np.flatnonzero((raw[s].astype(np.int64) != want).any(1))
                    assert bad.size == 0, '%s (shuffled %s) sample %d: block %d = [%d, %d): got %s want %s' % (
                        name, shuffled, s, bad[0], s0[bad[0]], e0[bad[0]], raw[s][bad[0]], want[bad[0]])
                    assert (b8[s] == OB.trim(want, False)).all() and (b16[s] == OB.trim(want, True)).all(), (name, s)
                    w3 = OB.beta2vec(want, 9)
                    assert (np.isnan(mean[s]) == np.isnan(w3)).all() and np.array_equal(mean[s][~np.isnan(w3)].view(np.uint64), w3[~np.isnan(w3)].view(np.uint64)), (name, s)


def test_block_sums_of_uint16_rows_against_oracle():
    """.lbeta rows (uint16 pairs): random tables incl. unsorted, overlapping, empty and tile-crossing blocks; the segment
    calls refuse such rows."""
    n, N = 300000, 3
    data = [cases.lbeta_twin(synth.synth_betas(99, s, 0, n)) for s in range(N)]
    data[1][1000:3000, :] = 65535                                 # coverage sums far beyond uint16 / uint32-per-tile worries
    rng = np.random.default_rng(5)
    s0 = rng.integers(0, n - 1, 40000)
    ln = np.where(rng.random(40000) < 0.9, rng.integers(0, 30, 40000), rng.integers(30, 9000, 40000))
    e0 = np.minimum(s0 + ln, n)
    s0 = np.concatenate([s0, [0, n - 1, n, 1023, 1024, 1025]]); e0 = np.concatenate([e0, [n, n, n, 1025, 2048, 1025]])
    with _lib.Segmenter(0) as sg:
        sg.set_lbetas(data)
        b8 = sg.block_sums(s0, e0, mode=1)
        b16 = sg.block_sums(s0, e0, mode=2)
        mean = sg.block_sums(s0, e0, mode=3, min_cov=1000)
        short = (e0 - s0) <= 65536
        raw = sg.block_sums(s0[short], e0[short], mode=0)
        for s in range(N):
            want = OB.block_sums(data[s], s0, e0)
            assert (b8[s] == OB.trim(want, False)).all() and (b16[s] == OB.trim(want, True)).all()
            w3 = OB.beta2vec(want, 1000)
            assert (np.isnan(mean[s]) == np.isnan(w3)).all() and np.array_equal(mean[s][~np.isnan(w3)], w3[~np.isnan(w3)])
            assert (raw[s].astype(np.int64) == (want[short] & 0xffffffff)).all()
        with pytest.raises(_lib.SegmentorError, match='exact up to 65536 sites'):
            sg.block_sums([0], [70000], mode=0)
        sg.set_loci(np.arange(n, dtype=np.uint32) * 50)
        with pytest.raises(_lib.SegmentorError, match='uint16'):
            sg.segment_chunks([0], [1000], 15.0, 1000, 2000)


def test_block_sums_full_size_tiling_property():
    """hg19-sized rows: any tiling of the sites sums to the column totals of the raw bytes (two granularities)."""
    n, N = 28217448, 3
    import ctypes as C
    import torch
    pitch = ((2 * n + 255) // 256) * 256 + 256
    buf = torch.empty((N, pitch), dtype=torch.uint8, device='cuda:0')
    assert _lib.load_synth().wgbssynth_fill_betas(C.c_void_p(buf.data_ptr()), pitch, n, 0, N, 77, None) == 0
    sg = _lib.Segmenter(0)
    try:
        sg.set_betas_device(buf.data_ptr(), N, pitch, n, keepalive=buf)
        host = buf[:, :2 * n].cpu().numpy().reshape(N, n, 2).astype(np.int64).sum(axis=1)
        for step in (16, 997):
            b = np.arange(0, n + step, step, dtype=np.int64).clip(max=n)
            got = sg.block_sums(b[:-1], b[1:], mode=0).astype(np.int64).sum(axis=1)
            assert (got == host).all(), (step, got, host)
        print('k_block_sums: %.3f ms for %d blocks x %d samples' % (sg.last_block_sums_ms(), b.size - 1, N))
    finally:
        sg.close()


@pytest.mark.parametrize('general', [0, 1])
def test_bin_rows_of_exact_quotients(general, monkeypatch):
    """.bin rows whose rescale 255 m / c is an EXACT integer — every site of a sample (K, 255), so a block of j sites has m = K j, c = 255 j — plus m = 0, m = c
    and counts one off such quotients: round 5's integer rescale in the streaming kernel (and the float64 one of the general kernel) against numpy's float64
    trim_to_uint8 restatement, blocks of 2 .. 1000 sites."""
    monkeypatch.setenv('WGBSSEG_BLOCK_SUMS_GENERAL', str(general))
    n = 400000
    ks = [0, 1, 51, 85, 127, 128, 254, 255]
    data = []
    for K in ks:
        d = np.empty((n, 2), dtype=np.uint8)
        d[:, 0] = K; d[:, 1] = 255
        data.append(d)
    rng = np.random.default_rng(23)
    off = data[3].copy(); off[rng.integers(0, n, 40000), 0] = 86          # one count off the exact quotient here and there
    few = data[1].copy(); few[rng.integers(0, n, 40000), 0] = 0
    data += [off, few, synth.synth_betas(5, 0, 0, n)]
    ln = np.where(rng.random(60000) < 0.7, rng.integers(2, 30, 60000), rng.integers(30, 1000, 60000))
    edges = np.concatenate([[0], np.cumsum(ln)])
    edges = edges[edges <= n]
    s0, e0 = edges[:-1], edges[1:]
    with _lib.Segmenter(0) as sg:
        sg.set_betas(data)
        b8 = sg.block_sums(s0, e0, mode=1)
        for s in range(len(data)):
            want = OB.trim(OB.block_sums(data[s], s0, e0), False)
            assert (b8[s] == want).all(), (general, s, int(np.flatnonzero((b8[s] != want).any(1))[0]))
        for i, K in enumerate(ks):                                      # the exact quotients themselves: (K, 255) for every saturated block
            sat = (e0 - s0) * 255 > 255
            assert (b8[i][sat, 0] == K).all() and (b8[i][sat, 1] == 255).all()


def test_bin_rows_of_a_corrupt_file_do_not_depend_on_the_kernel(monkeypatch):
    """the block reduction does not check meth <= cov (neither does beta_to_blocks.py:101-126).  Rows with meth > cov — a corrupt .beta —
    leave the domain the streaming kernel's integer rescale is proved on; it falls back to the float64 form there, so that the streaming kernel,
    the general kernel and numpy's float64 trim_to_uint8 restatement (utils_wgbs.py:277-290; wrap-around of the uint8 store included) agree."""
    n = 200000
    rng = np.random.default_rng(29)
    bad = synth.synth_betas(6, 0, 0, n).copy()
    where = rng.integers(0, n, 30000)
    bad[where, 0] = 255; bad[where, 1] = rng.integers(1, 40, where.size)          # meth far above cov: block quotients of 255 m / c up to several thousand
    worse = np.empty((n, 2), dtype=np.uint8); worse[:, 0] = 255; worse[:, 1] = 3    # every block: m = 85 c
    ln = np.where(rng.random(40000) < 0.7, rng.integers(2, 30, 40000), rng.integers(30, 1000, 40000))
    edges = np.concatenate([[0], np.cumsum(ln)]); edges = edges[edges <= n]
    s0, e0 = edges[:-1], edges[1:]
    res = []
    for general in (0, 1):
        monkeypatch.setenv('WGBSSEG_BLOCK_SUMS_GENERAL', str(general))
        with _lib.Segmenter(0) as sg:
            sg.set_betas([bad, worse])
            res.append(sg.block_sums(s0, e0, mode=1).copy())
    assert (res[0] == res[1]).all()
    for s, d in enumerate((bad, worse)):
        sums = OB.block_sums(d, s0, e0)
        assert (sums[:, 0] > sums[:, 1]).any()
        assert (res[0][s] == OB.trim(sums, False)).all()
