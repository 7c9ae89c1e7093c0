"""`convert` without a GPU: the oracle's join rules and the host logic of the mirror (BED parsing,
per-chromosome rule choice, NA / duplicate / order handling, text round trip, GenomicRegion) against vectors captured from the
reference's own Python (tests/golden/make_golden_convert.py)."""
import contextlib
import io
import json
import os.path as op

import numpy as np
import pytest

from oracle import convert_oracle as OC
from wgbs_tools_amd import convert as CV, genome as G, synth

HERE = op.dirname(op.abspath(__file__))


@pytest.fixture(scope='module')
def cworld(tmp_path_factory):
    g = json.load(open(op.join(HERE, 'golden', 'convert_cases.json')))
    names = [c for c, _ in g['chroms']]
    sizes = [s for _, s in g['chroms']]
    loci = synth.synth_loci(g['seed'], sizes)
    td = tmp_path_factory.mktemp('convert')
    ref = synth.write_genome(str(td / 'references' / 'synth'), names, sizes, loci)
    beds = {}
    for name, rec in g['bed'].items():
        p = str(td / (name + '.bed'))
        with open(p, 'w') as f:
            for r in rec['rows']:
                f.write('\t'.join(r) + '\n')
        beds[name] = p
    return dict(g=g, names=names, sizes=sizes, loci=loci, ref=ref, beds=beds, td=td)


class OracleLociEngine:
    """stands in for convert.LociEngine (same convert_regions contract): the numpy oracle instead of the GPU"""

    def __init__(self, loci):
        self.loci = loci.astype(np.int64)

    def convert_regions(self, clo, chi, cbp, start, end, slow):
        s = np.zeros(len(start), dtype=np.int64)
        e = np.zeros(len(start), dtype=np.int64)
        for lo, hi in sorted(set(zip(clo.tolist(), chi.tolist()))):
            if hi <= lo:
                continue
            for mode in (0, 1):
                rows = np.flatnonzero((clo == lo) & (chi == hi) & (slow == mode))
                if not rows.size:
                    continue
                L = self.loci[lo:hi]
                if mode == 0:
                    a, b = OC.fast_join(L, lo, start[rows], end[rows])
                else:
                    a, b = OC.slow_join(L, lo, start[rows], end[rows], cbp[rows])
                s[rows], e[rows] = a, b
        return s, e

    def close(self):
        pass


@pytest.mark.parametrize('name', ['clean_shuffled', 'three_columns_sorted', 'overlaps_in_chr2', 'with_header'])
@pytest.mark.parametrize('drop', [False, True])
def test_bed_to_cpgs_matches_reference(cworld, name, drop):
    rec = cworld['g']['bed'][name]['drop_empty' if drop else 'keep']
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        lines = CV.add_cpgs_to_bed(cworld['beds'][name], cworld['ref'], drop, engine=OracleLociEngine(cworld['loci']))
    text = '\n'.join(lines) + '\n'
    assert text == rec['text']
    assert err.getvalue() == rec['stderr']


def test_oracle_joins_against_reference_columns(cworld):
    """the oracle alone, row by row: regions of one chromosome, both rule sets, against the CpG columns the reference wrote"""
    g = cworld['g']
    cum = np.concatenate([[0], np.cumsum(cworld['sizes'])])
    loci = cworld['loci'].astype(np.int64)
    for name in ('three_columns_sorted', 'overlaps_in_chr2'):
        rows = [l.split('\t') for l in g['bed'][name]['keep']['text'].splitlines()]
        for ci, c in enumerate(cworld['names']):
            mine = [r for r in rows if r[0] == c]
            if not mine:
                continue
            st = np.array([int(r[1]) for r in mine]); en = np.array([int(r[2]) for r in mine])
            want_s = np.array([0 if r[3] == 'NA' else int(r[3]) for r in mine]); want_e = np.array([0 if r[4] == 'NA' else int(r[4]) for r in mine])
            L = loci[cum[ci]:cum[ci + 1]]
            uniq = np.unique(np.stack([st, en], 1), axis=0)
            if OC.has_overlaps(uniq[:, 0], uniq[:, 1]):
                bp = G.GenomeRefPaths(cworld['ref']).get_chrom_size(c)
                s, e = OC.slow_join(L, int(cum[ci]), st, en, bp)
            else:
                s, e = OC.fast_join(L, int(cum[ci]), st, en)
            assert np.array_equal(s, want_s) and np.array_equal(e, want_e), (name, c)


def test_genomic_region_matches_reference(cworld):
    gen = G.GenomeRefPaths(cworld['ref'])
    for r, rec in cworld['g']['regions'].items():
        if 'error' in rec:
            with pytest.raises(G.IllegalArgumentError) as ei:
                G.GenomicRegion(region=r, genome=gen)
            assert str(ei.value) == rec['error'], r
        else:
            gr = G.GenomicRegion(region=r, genome=gen)
            assert list(gr.sites) == rec['sites'] and str(gr) == rec['str'] and gr.region_str == rec['region_str'], r
    for s, rec in cworld['g']['sites'].items():
        err = io.StringIO()
        if 'error' in rec:
            with pytest.raises(G.IllegalArgumentError) as ei, contextlib.redirect_stderr(err):
                G.GenomicRegion(sites=s, genome=gen)
            assert str(ei.value) == rec['error'] and err.getvalue() == rec['stderr'], s
        else:
            gr = G.GenomicRegion(sites=s, genome=gen)
            assert list(gr.sites) == rec['sites'] and str(gr) == rec['str'] and gr.region_str == rec['region_str'], s


def test_column_round_trip_and_file_rules(tmp_path, cworld):
    assert CV.column_text(['1', '20', '+3']) == ['1', '20', '3']
    assert CV.column_text(['1', 'NA', '3']) == ['1.0', 'NA', '3.0']
    assert CV.column_text(['0.50', '1e3', '7', '']) == ['0.5', '1000.0', '7.0', 'NA']
    assert CV.column_text(['a', '0.50', 'NA']) == ['a', '0.50', 'NA']
    assert CV.column_text(['0.50', 'x'], raw=True) == ['0.50', 'x']
    p = tmp_path / 'empty.bed'
    p.write_text('# nothing\n\n')
    with pytest.raises(CV.IllegalArgumentError, match='Invalid bed file'):
        CV.load_bed(str(p))
    p.write_text('chr1\t5\t9\nchr1\t10\t20\textra\n')
    with pytest.raises(CV.IllegalArgumentError, match='Expected 3 fields in line 2, saw 4'):
        CV.load_bed(str(p))
    p.write_text('chr1\t5\t9\tx # trailing comment\n#whole line\nchr1\t10\t20\n')
    t = CV.load_bed(str(p))
    assert len(t) == 2 and t.extra == [['x ', '']] and t.end.tolist()[0] == 9
    out = tmp_path / 'o.bed'
    out.write_text('old')
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        assert CV.delete_or_skip(str(out), False) is False
    assert 'already exists. Skipping it. Use [-f] flag to force overwrite.' in err.getvalue()
    assert CV.delete_or_skip(str(out), True) is True and not out.exists()
    s, e = CV.load_site_file(str(_write(tmp_path / 's.txt', '5\n7\t9\n\n11 12\n')))
    assert s.tolist() == [5, 7, 11] and e.tolist() == [6, 9, 12]


def _write(p, text):
    p.write_text(text)
    return p
