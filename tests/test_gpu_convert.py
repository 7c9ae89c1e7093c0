"""`wgbstools convert` on the GPU (k_convert behind wgbsseg_convert_regions) against the vectors captured from the reference's
Python and against the oracle; the inverse direction (--site_file -> wgbsseg_add_loci) and the single-region forms."""
import contextlib
import io
import os.path as op

import numpy as np
import pytest

from oracle import convert_oracle as OC
from wgbs_tools_amd import _lib, synth, wgbs_tools
from test_convert_cpu import cworld          # noqa: F401  (fixture: genome directory, BED files and goldens on disk)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['clean_shuffled', 'three_columns_sorted', 'overlaps_in_chr2', 'with_header'])
def test_cli_bed_to_cpgs_matches_reference(cworld, name, tmp_path):
    for drop in (False, True):
        rec = cworld['g']['bed'][name]['drop_empty' if drop else 'keep']
        out = str(tmp_path / ('o%d.bed' % drop))
        err = io.StringIO()
        with contextlib.redirect_stderr(err):
            rc = wgbs_tools.main(['wgbstools', 'convert', '-L', cworld['beds'][name], '--genome', cworld['ref'], '-o', out] + (['--drop_empty'] if drop else []))
        assert rc == 0 and open(out).read() == rec['text'] and err.getvalue() == rec['stderr']
    # an existing output is left alone without -f
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        assert wgbs_tools.main(['wgbstools', 'convert', '-L', cworld['beds'][name], '--genome', cworld['ref'], '-o', out]) == 0
    assert 'already exists. Skipping it.' in err.getvalue()


def test_round_trip_and_single_regions(cworld, tmp_path, capsys):
    """--site_file (CpG ranges -> BED rows) followed by -L gives the CpG ranges back; -r / -s print the reference's description"""
    rng = np.random.default_rng(3)
    total = int(sum(cworld['sizes']))
    cum = np.concatenate([[0], np.cumsum(cworld['sizes'])])
    s = np.sort(rng.integers(1, total, 4000))
    e = np.minimum(s + rng.integers(1, 300, 4000), cum[np.searchsorted(cum, s, 'left')] + 1)        # stay inside the chromosome
    sites = tmp_path / 'sites.txt'
    sites.write_text(''.join('%d\t%d\n' % (a, b) for a, b in zip(s, e)))
    bed = str(tmp_path / 'from_sites.bed')
    assert wgbs_tools.main(['wgbstools', 'convert', '--site_file', str(sites), '--genome', cworld['ref'], '-o', bed]) == 0
    back = str(tmp_path / 'back.bed')
    with contextlib.redirect_stderr(io.StringIO()):
        assert wgbs_tools.main(['wgbstools', 'convert', '-L', bed, '--genome', cworld['ref'], '-o', back]) == 0
    rows = [l.split('\t') for l in open(back).read().splitlines()]
    assert len(rows) == 4000
    # columns 4-5 are the join's answer, 6-7 the ranges the BED rows were made from
    got = np.array([[int(r[3]), int(r[4])] for r in rows]); src = np.array([[int(r[5]), int(r[6])] for r in rows])
    assert np.array_equal(got, src)
    capsys.readouterr()
    for r, rec in list(cworld['g']['regions'].items()) + list(cworld['g']['sites'].items()):
        flag = '-r' if r in cworld['g']['regions'] else '-s'
        with contextlib.redirect_stderr(io.StringIO()):
            rc = wgbs_tools.main(['wgbstools', 'convert', flag, r, '--genome', cworld['ref']])
        out = capsys.readouterr().out
        if 'error' in rec:
            assert rc == 1 and out == ''
        else:
            assert rc == 0 and out == rec['str'] + '\n'


def test_two_million_regions_against_oracle():
    """hg19-sized loci, 2 M random regions of both rule sets and every corner (edges on CpGs, empty, reversed, beyond the
    chromosome, unknown chromosome): the device join == the numpy oracle."""
    names, sizes = synth.genome_shape(synth.HG19_NR_SITES, 25)
    sizes = [int(x) for x in sizes]
    loci = synth.synth_loci(11, sizes)
    cum = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    L64 = loci.astype(np.int64)
    rng = np.random.default_rng(7)
    n = 2000000
    ci = rng.integers(0, 25, n)
    clo, chi = cum[ci], cum[ci + 1]
    anchor = L64[rng.integers(clo, chi)]
    start = anchor + rng.integers(-300, 300, n) * (rng.random(n) < 0.7)
    end = start + rng.integers(-5, 4000, n)
    on = rng.random(n) < 0.2
    end = np.where(on, L64[np.minimum(np.searchsorted(L64[:], 0) + rng.integers(clo, chi), chi - 1)], end)     # ends exactly on CpGs
    cbp = L64[chi - 1] + 10000
    slow = (rng.random(n) < 0.4).astype(np.uint8)
    unknown = rng.random(n) < 0.01
    clo = np.where(unknown, 0, clo); chi = np.where(unknown, 0, chi)
    with _lib.Segmenter(0) as seg:
        seg.set_loci(loci)
        s, e = seg.convert_regions(clo, chi, cbp, start, end, slow)
        ms = seg.last_block_sums_ms()
    ws = np.zeros(n, dtype=np.int64); we = np.zeros(n, dtype=np.int64)
    for c in range(25):
        lo, hi = int(cum[c]), int(cum[c + 1])
        for mode in (0, 1):
            rows = np.flatnonzero((clo == lo) & (chi == hi) & (slow == mode))
            if mode == 0:
                a, b = OC.fast_join(L64[lo:hi], lo, start[rows], end[rows])
            else:
                a, b = OC.slow_join(L64[lo:hi], lo, start[rows], end[rows], cbp[rows])
            ws[rows], we[rows] = a, b
    assert np.array_equal(s, ws) and np.array_equal(e, we)
    assert (s[unknown] == 0).all() and (s != 0).sum() > n // 2
    print('k_convert: %.3f ms for %d regions against %d loci' % (ms, n, loci.size))


def test_both_rule_sets_on_the_tabix_free_cross_check(cworld):
    """The 400 regions the reference answered twice (tests/golden/make_golden_convert.py: its `tabix | awk` pipeline on the stand-in AND its
    tabix-free pandas joins) through k_convert with either rule set: the device's answers are the reference's on both paths."""
    g = cworld['g']
    rows = g['shim_cross_check']['rows']
    cum = np.concatenate([[0], np.cumsum(cworld['sizes'])]).astype(np.int64)
    ci = np.array([cworld['names'].index(r[0]) for r in rows])
    start = np.array([r[1] for r in rows], dtype=np.int64); end = np.array([r[2] for r in rows], dtype=np.int64)
    from wgbs_tools_amd import genome as G
    gen = G.GenomeRefPaths(cworld['ref'])
    cbp = np.array([gen.get_chrom_size(r[0]) for r in rows], dtype=np.int64)
    with _lib.Segmenter(0) as seg:
        seg.set_loci(cworld['loci'])
        for slow, cs, ce in ((0, 5, 6), (1, 3, 4)):
            s, e = seg.convert_regions(cum[ci], cum[ci + 1], cbp, start, end, np.full(len(rows), slow, dtype=np.uint8))
            assert s.tolist() == [r[cs] for r in rows] and e.tolist() == [r[ce] for r in rows], 'rule set %d' % slow
