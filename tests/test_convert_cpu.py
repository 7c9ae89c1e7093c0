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


def test_tabix_stand_in_cross_check_and_both_rule_sets(cworld):
    """the vectors of GenomicRegion (and of the BED rows that fall back to it) were generated through a `tabix` stand-in.  The
    fixture holds 400 random regions answered by the reference TWICE — its `tabix | awk` pipeline (stand-in) and chr_thread's pandas joins (no
    tabix) — which must agree but for the CpG exactly at a region's end (genomic_region.py:147-150 leaves it out, convert.py:169 keeps it).
    Held against both here: GenomicRegion and the oracle's two join rules."""
    g = cworld['g']
    flags = g['via_tabix_shim']
    assert flags['bed']['clean_shuffled'] is False and flags['regions'] is True and flags['shim_free_cross_check'] is False
    x = g['shim_cross_check']
    assert x['n'] == len(x['rows']) == x['agree'] >= 390 and x['end_on_a_cpg'] > 50
    gen = G.GenomeRefPaths(cworld['ref'])
    cum = np.concatenate([[0], np.cumsum(cworld['sizes'])])
    loci = cworld['loci'].astype(np.int64)
    for c, a, b, ps, pe, js, je, on in x['rows']:
        assert (ps, pe) == (js, je - on)                                            # the stand-in's path == the tabix-free path, by the reference's rules
        gr = G.GenomicRegion(region='%s:%d-%d' % (c, a, b), genome=gen)           # the mirror of the pipeline
        assert tuple(gr.sites) == (ps, pe), (c, a, b)
        ci = cworld['names'].index(c)
        L = loci[cum[ci]:cum[ci + 1]]
        st, en = np.array([a]), np.array([b])
        s1, e1 = OC.fast_join(L, int(cum[ci]), st, en)                             # chr_thread's rules
        assert (int(s1[0]), int(e1[0])) == (js, je), (c, a, b)
        s2, e2 = OC.slow_join(L, int(cum[ci]), st, en, np.array([gen.get_chrom_size(c)]))
        assert (int(s2[0]), int(e2[0])) == (ps, pe), (c, a, b)


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


# ---------------------------------------------------------------------------------------------------------
# the library's text path for `convert -L` (csrc/table_io.h: wgbsseg_bed_parse / _write_annotated) against the Python path
# ---------------------------------------------------------------------------------------------------------
def _python_text(path, cworld, drop):
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        lines = CV.add_cpgs_to_bed(path, cworld['ref'], drop, engine=OracleLociEngine(cworld['loci']))
    return '\n'.join(lines) + ('\n' if lines else ''), err.getvalue()


def _fast_text(path, cworld, drop, out):
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        done = CV.annotate_bed_fast(path, G.GenomeRefPaths(cworld['ref']), drop, out, engine=OracleLociEngine(cworld['loci']))
    return done, (open(out).read() if done is True else None), err.getvalue()


def test_fast_text_path_matches_python_path(cworld, tmp_path, capfd):
    g = cworld['g']
    names, sizes, loci = cworld['names'], cworld['sizes'], cworld['loci'].astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(sizes)])
    out = str(tmp_path / 'o.bed')
    taken = 0
    # the reference's fixtures: the fast path either declines (header ...) or writes the golden text
    for name, p in cworld['beds'].items():
        for drop in (False, True):
            done, text, err = _fast_text(p, cworld, drop, out)
            want, werr = _python_text(p, cworld, drop)
            if done is True:
                taken += 1
                assert text == want and err == werr, (name, drop)
                assert text == g['bed'][name]['drop_empty' if drop else 'keep']['text']
            else:
                assert done is False
    assert taken == 6                                           # the one with a header line too; not the one whose scores read 0.10, 1.00 (0.1, 1.0 on the way out)
    # tables made here: text columns, integer columns, NA in a text column, unknown chromosomes, regions without CpGs, overlaps
    # in one chromosome (the per-row rule set), no last newline, blank lines, gzip
    rng = np.random.default_rng(4)
    rows = []
    for i in range(4000):
        ci = int(rng.integers(0, len(names)))
        lo = int(loci[cum[ci]]); hi = int(loci[cum[ci + 1] - 1])
        a = int(rng.integers(max(1, lo - 500), hi + 500))
        b = a + int(rng.integers(1, 3000))
        chrom = names[ci] if rng.random() > 0.02 else 'chrUn_%d' % i
        rows.append([chrom, str(a), str(b), 'name%d' % i if i % 7 else 'NA', str(int(rng.integers(0, 1000))), '+-'[i % 2]])
    rows.append([names[0], '1', '2', 'tiny', '0', '+'])
    def write(path, rs, tail='\n', sep_blank=False):
        with open(path, 'w') as f:
            f.write(('\n\n' if sep_blank else '') + '\n'.join('\t'.join(r) for r in rs) + tail)
        return path
    files = {'six_columns': write(str(tmp_path / 'a.bed'), rows),
             'three_columns': write(str(tmp_path / 'b.bed'), [r[:3] for r in rows], tail=''),
             'sorted_no_overlap': write(str(tmp_path / 'c.bed'), sorted([r[:4] for r in rows if r[0] == names[1]][::25], key=lambda r: int(r[1])), sep_blank=True)}
    import gzip
    gz = str(tmp_path / 'a.bed.gz')
    with gzip.open(gz, 'wb') as f:
        f.write(open(files['six_columns'], 'rb').read())
    files['gz'] = gz
    for name, p in files.items():
        for drop in (False, True):
            done, text, err = _fast_text(p, cworld, drop, out)
            want, werr = _python_text(p, cworld, drop)
            assert done is True, name
            assert text == want and err == werr, (name, drop)
    # standard output
    capfd.readouterr()
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        assert CV.annotate_bed_fast(files['three_columns'], G.GenomeRefPaths(cworld['ref']), False, None, engine=OracleLociEngine(cworld['loci'])) is True
    assert capfd.readouterr().out == _python_text(files['three_columns'], cworld, False)[0]
    # what the fast path must decline: whatever a round trip through pandas re-prints, and the malformed
    base = [names[0], '100', '900']
    odd = {'header_only': 'chr\tstart\tend\n', 'header_then_float_start': 'chr\tstart\tend\n' + names[0] + '\t1.5\t9\n', 'comment': '\t'.join(base) + ' # c\n', 'comment_line': '# c\n' + '\t'.join(base) + '\n',
           'ragged_short': '\t'.join(base + ['x']) + '\n' + '\t'.join(base) + '\n', 'ragged_long': '\t'.join(base) + '\n' + '\t'.join(base + ['x']) + '\n',
           'float_column': '\t'.join(base + ['0.50']) + '\n' + '\t'.join(base + ['1']) + '\n', 'int_with_na': '\t'.join(base + ['5']) + '\n' + '\t'.join(base + ['NA']) + '\n',
           'plus_int': '\t'.join(base + ['+5']) + '\n', 'leading_zero_start': names[0] + '\t0100\t900\n', 'spaced_start': names[0] + '\t 100\t900\n',
           'empty_field_in_text': '\t'.join(base + ['abc']) + '\n' + '\t'.join(base + ['']) + '\n', 'nan_in_text': '\t'.join(base + ['abc']) + '\n' + '\t'.join(base + ['nan']) + '\n',
           'crlf': '\t'.join(base) + '\r\n', 'two_columns': names[0] + '\t5\n', 'empty': '', 'utf8': 'chré\t1\t2\n', 'inf_column': '\t'.join(base + ['inf']) + '\n',
           'form_feed': '\t'.join(base + ['a\x0cb']) + '\n', 'float_start': names[0] + '\t100.0\t900\n', 'exponent_like': '\t'.join(base + ['1e5']) + '\n' + '\t'.join(base + ['e10']) + '\n', 'underscore_number': '\t'.join(base + ['1_000']) + '\n', 'big_start': names[0] + '\t1234567890123456\t1234567890123457\n'}
    from wgbs_tools_amd import _lib
    for name, text in odd.items():
        assert _lib.bed_parse(text.encode('utf-8'), names) is None, name
    # ... while their neighbours are taken
    for name, text in {'header': 'chr\tstart\tend\tscore\n' + '\t'.join(base + ['0.50']) + '\n' + '\t'.join(base + ['+7']) + '\n', 'header_spaced_numbers_are_no_header': names[0] + '\t100\t900\n',
                       'na_in_text': '\t'.join(base + ['abc']) + '\n' + '\t'.join(base + ['NA']) + '\n', 'int_column': '\t'.join(base + ['5']) + '\n' + '\t'.join(base + ['0']) + '\n',
                       'numbers_in_text': '\t'.join(base + ['0.50']) + '\n' + '\t'.join(base + ['x']) + '\n', 'numeric_chrom': '1\t100\t900\n2\t5\t9\n', 'letters_and_digits': '\t'.join(base + ['n1', 'a2', '+']) + '\n' + '\t'.join(base + ['f3', 'NA', '-']) + '\n',
                       'zero_start': names[0] + '\t0\t900\n'}.items():
        p = str(tmp_path / (name + '.bed'))
        open(p, 'w').write(text)
        assert _lib.bed_parse(text.encode('utf-8'), names) is not None, name
        done, got, err = _fast_text(p, cworld, False, out)
        want, werr = _python_text(p, cworld, False)
        assert done is True and got == want and err == werr, name


def test_decimal_tokens_that_print_as_they_read():
    """canonical_float (the rule that lets a column of scores go back out verbatim): accepted => Python's repr(float(token)) is the
    token itself — on reprs of random doubles over the whole positional range, on their perturbations (a digit changed, zeros
    added, digits dropped, signs, exponents), and on hand-picked edges; and the reprs themselves are accepted wherever the rule's
    shape admits them (digits.digits, 1e-4 <= |v| < 1e16)."""
    from wgbs_tools_amd import _lib
    rng = np.random.default_rng(9)
    vals = np.concatenate([rng.random(60000), rng.random(20000) * 1000, np.exp(rng.uniform(np.log(1e-6), np.log(1e17), 60000)),
                           np.round(rng.random(20000), 2), np.round(rng.random(20000) * 100, 3), -rng.random(5000) * 50])
    toks = [repr(float(v)) for v in vals]
    extra = []
    for t in toks[:40000]:
        k = int(rng.integers(0, 6))
        if k == 0: extra.append(t + '0')
        elif k == 1: extra.append('0' + t)
        elif k == 2 and len(t) > 4: extra.append(t[:-1])
        elif k == 3 and len(t) > 4: extra.append(t[:-2] + str((int(t[-2]) + 1) % 10 if t[-2].isdigit() else 0) + t[-1])
        elif k == 4: extra.append('+' + t)
        else: extra.append(t + '1')
    hand = ['0.0', '-0.0', '0.5', '5.0', '5', '5.', '.5', '0.50', '00.5', '1e3', '1e-3', '0.0001', '0.00001', '0.00009999999999999999', '9999999999999998.0',
            '10000000000000000.0', '1000000000000000.5', '0.1', '0.30000000000000004', '0.3000000000000000444', '123456789.123456789', '-7.25', '--1.0', '1.0.0',
            '1_0.5', ' 1.5', '1.5 ', 'nan', 'inf', '', '.', '-', '-.5', '1.', '17.0', '0.1000000000000000055511151231257827', '4.35', '2.675', '1.005', '100.0']
    allt = toks + extra + hand
    ok = _lib.debug_canonical_float(allt)
    import re
    shape = re.compile(r'^-?\d+\.\d+$')
    n_ok = 0
    for t, a in zip(allt, ok.tolist()):
        try:
            same = repr(float(t)) == t
        except ValueError:
            same = False
        if a:
            n_ok += 1
            assert same, t                                             # never accept what would be re-printed
        elif same and shape.match(t):
            v = abs(float(t))
            assert not (v == 0.0 or 1e-4 <= v < 1e16), t                 # and do not turn down what fits the rule
    assert n_ok > 120000 and ok[len(toks) + len(extra)] and not ok[len(toks) + len(extra) + 4]


def test_score_columns_go_through_the_fast_path(cworld, tmp_path):
    """columns of decimals in their shortest form (what repr / to_csv print) pass, with gaps too; the reference's fixture, whose
    scores read 0.10 and 1.00, does not"""
    out = str(tmp_path / 'o.bed')
    assert _fast_text(cworld['beds']['clean_shuffled'], cworld, False, out)[0] is False
    names = cworld['names']
    rows = [[names[i % len(names)], str(1000 + 37 * i), str(1500 + 37 * i), 'r%d' % i, repr(round(i / 7.0, 3)) if i % 5 else 'NA'] for i in range(300)]
    p = str(tmp_path / 'scores.bed')
    open(p, 'w').write('\n'.join('\t'.join(r) for r in rows) + '\n')
    done, text, err = _fast_text(p, cworld, False, out)
    assert done is True and (text, err) == _python_text(p, cworld, False)
    rows[17][4] = '0.50'                                                   # one value that pandas would print as 0.5: the whole table to the Python path
    open(p, 'w').write('\n'.join('\t'.join(r) for r in rows) + '\n')
    assert _fast_text(p, cworld, False, out)[0] is False
    rows[17][4] = '3'                                                      # an integer among decimals: 3.0 in the output
    open(p, 'w').write('\n'.join('\t'.join(r) for r in rows) + '\n')
    assert _fast_text(p, cworld, False, out)[0] is False


def test_random_tables_fast_path_never_disagrees(cworld, tmp_path):
    """3,000 small random BED tables over an alphabet of tricky tokens (numbers in every spelling, missing-value spellings, signs,
    names that look like numbers, blanks): whenever the library's parser takes a table, the text it writes is the Python path's."""
    from wgbs_tools_amd import _lib
    names = cworld['names']
    gen = G.GenomeRefPaths(cworld['ref'])
    rng = np.random.default_rng(77)
    alphabet = ['a', 'x1', 'n1', 'NA', '', 'nan', 'NaN', 'N/A', '0', '1', '7', '007', '+5', '-3', '0.5', '0.50', '1.0', '1.', '.5', '1e3', '1E-2', 'inf', '-inf',
                'Infinity', '+', '-', '.', 'e', 'e5', '1_0', ' 1', '1 ', 'chr1', '12345678901234567890', '0.1', '0.30000000000000004', '2.5', '100.25', '-0.0',
                '1,5', 'a b', 'None', 'null', '#', 'x#y', '3.0', '0.0001', '0.00001']
    starts = ['100', '0', '1', '12', '012', '+3', '1.0', ' 4', '', 'NA', 'x', '99999', '1e2']
    out = str(tmp_path / 'o.bed')
    taken = 0
    for it in range(3000):
        width = int(rng.integers(3, 7))
        n_rows = int(rng.integers(1, 6))
        cols = [rng.choice(alphabet, size=int(rng.integers(1, 4)), replace=False).tolist() for _ in range(width)]
        rows = []
        for r in range(n_rows):
            w = width if rng.random() > 0.03 else int(rng.integers(1, 8))
            row = []
            for c in range(w):
                if c == 0:
                    row.append(str(rng.choice(names + ['chrZ', '1'])) if rng.random() > 0.1 else str(rng.choice(alphabet)))
                elif c in (1, 2):
                    row.append(str(rng.choice(starts)) if rng.random() < 0.15 else str(int(rng.integers(0, 30000))))
                else:
                    row.append(str(rng.choice(cols[c % width])))
            rows.append(row)
        text = '\n'.join('\t'.join(r) for r in rows) + ('\n' if rng.random() > 0.2 else '')
        if rng.random() < 0.05:
            text = 'chrom\tstart\tend' + '\tc' * (width - 3) + '\n' + text
        p = str(tmp_path / 't.bed')
        with open(p, 'w') as f:
            f.write(text)
        err = io.StringIO()
        with contextlib.redirect_stderr(err):
            done = CV.annotate_bed_fast(p, gen, False, out, engine=OracleLociEngine(cworld['loci']))
        if done is not True:
            continue
        taken += 1
        werr = io.StringIO()
        with contextlib.redirect_stderr(werr):
            lines = CV.add_cpgs_to_bed(p, cworld['ref'], False, engine=OracleLociEngine(cworld['loci']))
        want = '\n'.join(lines) + ('\n' if lines else '')
        assert open(out).read() == want and err.getvalue() == werr.getvalue(), (it, text)
    assert taken > 300, taken
