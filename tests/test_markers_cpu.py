"""find_markers without a GPU: the host logic of the mirror (parameters, groups, filters, quantiles,
tests, sorting, output text) with a numpy stand-in for the two device calls, against files captured from the reference's own
find_markers.py (tests/golden/make_golden_markers.py)."""
import contextlib
import io
import json
import os.path as op

import numpy as np
import pytest

import cases
from oracle import block_sums as OB
from wgbs_tools_amd import beta_to_blocks as B2B, find_markers as FM

HERE = op.dirname(op.abspath(__file__))


@pytest.fixture(scope='module')
def mworld(tmp_path_factory):
    g = json.load(open(op.join(HERE, 'golden', 'marker_cases.json')))
    td = str(tmp_path_factory.mktemp('markers'))
    w = cases.marker_world(td)
    assert json.loads(json.dumps(w['spec'])) == g['world']
    return dict(g=g, td=td, **w)


class OracleMarkerEngine:
    """stands in for BlockSumEngine: reduce(mode 3) and marker_stats by numpy (oracle/block_sums.py)"""

    def __init__(self, paths):
        self.data = [np.fromfile(p, dtype=np.uint8).reshape(-1, 2) for p in paths]
        self.table = None

    def reduce(self, t, mode=0, min_cov=1):
        assert mode == 3
        s0, e0 = B2B.block_site_ranges(t)
        self.table = np.array([OB.beta2vec(OB.block_sums(d, s0, e0), min_cov) for d in self.data])
        return self.table

    def marker_stats(self, tg, bg, n_blocks):
        assert self.table.shape[1] == n_blocks
        out = np.zeros((n_blocks, 8))
        for k, idx in enumerate((tg, bg)):
            v = self.table[np.asarray(idx)]
            ok = ~np.isnan(v)
            out[:, 4 * k] = ok.sum(axis=0)
            acc = np.zeros(n_blocks)
            for row in np.where(ok, v, 0.0):                       # sequential, in the order given
                acc = acc + row
            out[:, 4 * k + 1] = acc
            with np.errstate(all='ignore'):
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    out[:, 4 * k + 2] = np.nanmin(v, axis=0)
                    out[:, 4 * k + 3] = np.nanmax(v, axis=0)
        return out

    def close(self):
        pass


def run_case(mworld, extra, out_dir, engine_cls=None):
    argv = ['-b', mworld['blocks'], '-g', mworld['groups'], '--betas'] + mworld['betas'] + ['-o', out_dir] + extra
    params = FM.MFParams(FM.parse_args(argv))
    mf = FM.MarkerFinder(params)
    if engine_cls is not None:
        mf.engine = engine_cls(mf.paths)
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        mf.run()
    return err.getvalue()


@pytest.mark.parametrize('name', ['default', 'hypo_top', 'hyper_quants', 'two_targets_bg', 'mw_test', 'mvalue_test', 'single_sample_target'])
def test_markers_match_reference(mworld, name, tmp_path):
    rec = mworld['g']['cases'][name]
    od = str(tmp_path / 'out')
    err = run_case(mworld, rec['args'], od, OracleMarkerEngine)
    for fname, want in rec['files'].items():
        got = open(op.join(od, fname)).read()
        assert got == want, (name, fname)
    assert err.replace(od, '<OUT>') == rec['stderr'].replace('<TMP>/out_' + name, '<OUT>')


def test_parameters_and_errors(mworld, tmp_path):
    cfg = tmp_path / 'cfg.txt'
    cfg.write_text('# a comment\ndelta_means:0.45\ntargets:Liver Blood\nonly_hypo:True\ntop:NA\nmin_cov: 7\n')
    base = ['-b', mworld['blocks'], '-g', mworld['groups'], '--betas'] + mworld['betas'] + ['-o', str(tmp_path)]
    p = FM.MFParams(FM.parse_args(base + ['-p', str(cfg), '--delta_means', '0.5']))
    assert p.delta_means == 0.5 and p.targets == ['Liver', 'Blood'] and p.only_hypo is True and p.top is None and p.min_cov == 7
    assert p.bg_quant == 0.025 and p.test_type == 't' and p.chunk_size == 150000                    # defaults
    for bad, msg in ((['--pval', '1.5'], ''), (['--only_hyper', '--only_hypo'], ''), (['--sort_by', 'nope'], ''), (['--test_type', 'x'], ''),
                     (['--max_cpg', '0'], 'max_cpg must larger than 0')):
        with pytest.raises(FM.IllegalArgumentError) as ei, contextlib.redirect_stderr(io.StringIO()):
            FM.MFParams(FM.parse_args(base + bad))
        assert str(ei.value) == msg
    err = io.StringIO()
    with pytest.raises(FM.IllegalArgumentError), contextlib.redirect_stderr(err):
        FM.MarkerFinder(FM.MFParams(FM.parse_args(base + ['--targets', 'Livr'])))
    assert 'Invalid group: Livr' in err.getvalue() and 'Did you mean Liver?' in err.getvalue()
    assert FM.descending_order([3.0, np.nan, 5.0, 1.0]).tolist() == [2, 0, 3, 1]


def test_library_parsed_table_is_the_python_parsed_table(mworld, tmp_path, monkeypatch):
    """find_markers on the blocks table the library parsed (bytes + row offsets; bp columns as integers; annotation columns cut out
    for the markers' rows only) against the line-by-line Python table: same filters, same rows, same output files — with and
    without the two annotation columns, with a header, and for the sub-tables take() makes."""
    fast = B2B.load_blocks_file(mworld['blocks'], anno=True)
    assert fast.parsed is not None and fast.parsed.bp_start is not None
    monkeypatch.setenv('WGBSSEG_PY_TABLES', '1')
    slow = B2B.load_blocks_file(mworld['blocks'], anno=True)
    monkeypatch.delenv('WGBSSEG_PY_TABLES')
    assert slow.parsed is None and list(fast.extra) == list(slow.extra) and fast.columns == slow.columns
    assert fast.parsed.bp_start.tolist() == [int(x) for x in slow.start] and fast.parsed.bp_end.tolist() == [int(x) for x in slow.end]
    idx = np.array([0, 5, 17, len(slow) - 1, 3])
    sub_f, sub_s = fast.take(idx), slow.take(idx)
    assert sub_f.coords_of(np.arange(5)) == sub_s.coords_of(np.arange(5)) == [(slow.chr[i], slow.start[i], slow.end[i]) for i in idx]
    assert sub_f.extras_of([1, 4]) == sub_s.extras_of([1, 4])
    assert sub_f.chr == sub_s.chr and sub_f.startCpG.tolist() == sub_s.startCpG.tolist()
    for k in slow.extra:
        assert fast.extra[k] == slow.extra[k]
    # an annotated table (7 columns, a header, an NA row, a short row) through both parsers and through find_markers
    rows = open(mworld['blocks']).read().splitlines()
    anno = tmp_path / 'anno.bed'
    with open(anno, 'w') as f:
        f.write('chr\tstart\tend\tstartCpG\tendCpG\tanno\tgene\n')
        for i, r in enumerate(rows):
            tok = r.split('\t')[:5]
            f.write('\t'.join(tok + (['exon' if i % 3 else '', 'GENE%d' % i] if i % 11 else ['intron'])) + '\n')
        f.write('chr1\t5\t9\tNA\tNA\tx\ty')
    a = B2B.load_blocks_file(str(anno), anno=True)
    monkeypatch.setenv('WGBSSEG_PY_TABLES', '1')
    b = B2B.load_blocks_file(str(anno), anno=True)
    monkeypatch.delenv('WGBSSEG_PY_TABLES')
    assert a.parsed is not None and b.parsed is None and list(a.extra) == ['anno', 'gene'] == list(b.extra)
    assert a.extras_of(np.arange(len(a))) == {k: list(v) for k, v in b.extra.items()} and a.chr == b.chr and a.na.tolist() == b.na.tolist()
    outs = {}
    for mode in ('fast', 'python'):
        if mode == 'python':
            monkeypatch.setenv('WGBSSEG_PY_TABLES', '1')
        od = str(tmp_path / mode)
        w = dict(mworld, blocks=str(anno))
        run_case(w, ['--min_cpg', '1'], od, OracleMarkerEngine)
        outs[mode] = {f: open(op.join(od, f)).read() for f in sorted(__import__('os').listdir(od)) if f.startswith('Markers.')}
    monkeypatch.delenv('WGBSSEG_PY_TABLES')
    assert outs['fast'] == outs['python'] and outs['fast'] and any('GENE' in t for t in outs['fast'].values())
