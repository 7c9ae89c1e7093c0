"""Block reduction (beta_to_blocks / beta_to_table) without a GPU: the oracle and the host
logic of the mirrors against vectors captured from the reference's own Python (tests/golden/make_golden_blocks.py)."""
import base64
import hashlib
import io
import json
import os.path as op

import numpy as np
import pytest

from oracle import block_sums as OB
import cases
from wgbs_tools_amd import beta_to_blocks as B2B, beta_to_table as B2T, synth

HERE = op.dirname(op.abspath(__file__))


@pytest.fixture(scope='module')
def world(tmp_path_factory):
    g = json.load(open(op.join(HERE, 'golden', 'block_cases.json')))
    td = tmp_path_factory.mktemp('blocks')
    betas, data = [], []
    for s in g['samples']:
        d = synth.synth_betas(g['seed'], s, 0, g['n_sites'])
        p = str(td / ('smp%d.beta' % s))
        d.tofile(p)
        betas.append(p); data.append(d)
    assert synth.checksum(*[d.reshape(-1) for d in data]) == g['input_crc32']
    paths = {}
    for name, rec in g['tables'].items():
        p = str(td / (name + '.bed'))
        with open(p, 'w') as f:
            for c, s, e, a, b in rec['rows']:
                f.write('%s\t%d\t%d\t%s\t%s\n' % (c, s, e, 'NA' if a is None else a, 'NA' if b is None else b))
        paths[name] = p
    lbetas = []
    for s, d in zip(g['samples'], data):
        p = str(td / ('smp%d.lbeta' % s))
        cases.lbeta_twin(d).tofile(p)
        lbetas.append(p)
    gpath = str(td / 'groups.csv')
    with open(gpath, 'w') as f:
        f.write('name,group\nsmp0,A\nsmp1,B\nsmp2,A\nsmp3,B\n')
    return dict(g=g, td=td, betas=betas, lbetas=lbetas, data=data, blocks=paths, groups=gpath)


class OracleBlockEngine:
    """stands in for BlockSumEngine (same reduce() contract), numpy instead of the GPU"""

    def __init__(self, data):
        self.data = data                       # uint8 or uint16 [n, 2] arrays

    def reduce(self, df, mode=0, min_cov=1):
        s0, e0 = B2B.block_site_ranges(df)
        sums = [OB.block_sums(d, s0, e0) for d in self.data]
        if mode == 0:
            return np.array(sums).astype(np.uint32)
        if mode in (1, 2):
            return np.array([OB.trim(t, lbeta=(mode == 2)) for t in sums])
        return np.array([OB.beta2vec(t, min_cov) for t in sums])

    def close(self):
        pass


def _sha(text):
    return hashlib.sha1(text.encode()).hexdigest()


@pytest.mark.parametrize('name', ['nice', 'ragged'])
def test_oracle_and_host_logic_match_reference(world, name, tmp_path):
    g, rec = world['g'], world['g']['tables'][name]
    df = B2B.load_blocks_file(world['blocks'][name])
    assert B2B.is_block_file_nice(df) == (rec['is_nice'], rec['msg'])
    s0, e0 = B2B.block_site_ranges(df)
    eng = OracleBlockEngine(world['data'])
    for i, b in enumerate(world['betas']):
        key = op.basename(b)
        sums = OB.block_sums(world['data'][i], s0, e0)
        assert hashlib.sha1(np.ascontiguousarray(sums, dtype=np.int64).tobytes()).hexdigest() == rec['sums_sha1'][key]
        for lbeta, tag in ((False, 'bin'), (True, 'lbeta')):
            t = OB.trim(sums, lbeta)
            assert t.tobytes() == base64.b64decode(rec[tag][key]), (name, key, tag)
            assert B2B.trim_to_uint8(sums, lbeta).tobytes() == t.tobytes()
        # dump(): file names, .bin bytes and bedGraph text
        out = tmp_path / ('o%d' % i)
        out.mkdir()
        B2B.dump(df, OB.trim(sums, False), b, False, str(out), True)
        stem = str(out / op.splitext(key)[0])
        assert open(stem + '.bin', 'rb').read() == base64.b64decode(rec['bin'][key])
        bg = open(stem + '.bedGraph').read()
        assert bg[:600] == rec['bedgraph'][key]['head'] and len(bg) == rec['bedgraph'][key]['len']
        assert _sha(bg) == rec['bedgraph'][key]['sha1']
    # beta_to_table with the oracle behind the engine interface
    for tag, gfile, mc, dg in (('table_plain', None, 4, 2), ('table_groups', world['groups'], 10, 3)):
        gf = B2T.groups_load_wrap(gfile, world['betas'])
        t = B2T.get_table(df.copy(), gf, mc, engine=eng)
        buf = io.StringIO()
        B2T.dump(buf, t, True, dg)
        text = buf.getvalue()
        assert text[:600] == rec[tag]['head'], (name, tag)
        assert len(text) == rec[tag]['len'] and _sha(text) == rec[tag]['sha1']
        # in chunks, as the CLI walks the table
        buf = io.StringIO()
        for a in range(0, len(df), 257):
            B2T.dump(buf, B2T.get_table(df.rows(a, a + 257), gf, mc, engine=eng), a == 0, dg)
        assert buf.getvalue() == text
    # uint16 .lbeta INPUT files (utils_wgbs.py:311-319): the oracle's sums -> the reference's .bin / .lbeta bytes and table
    lrec = rec['lbeta_inputs']
    ldata = [cases.lbeta_twin(d) for d in world['data']]
    for i, lb in enumerate(world['lbetas']):
        key = op.basename(lb)
        sums = OB.block_sums(ldata[i], s0, e0)
        assert hashlib.sha1(OB.trim(sums, False).tobytes()).hexdigest() == lrec['bin_sha1'][key]
        assert hashlib.sha1(OB.trim(sums, True).tobytes()).hexdigest() == lrec['lbeta_sha1'][key]
    gf = B2T.groups_load_wrap(None, world['lbetas'])
    buf = io.StringIO()
    B2T.dump(buf, B2T.get_table(df.copy(), gf, 4, engine=OracleBlockEngine(ldata)), True, 3)
    text = buf.getvalue()
    assert text[:600] == lrec['table_plain']['head'] and _sha(text) == lrec['table_plain']['sha1']


def test_groups_file_rules(tmp_path):
    """groups csv: comments, `include` column, missing cells, unknown prefixes (dmb.py:24-79)."""
    for i in range(3):
        (tmp_path / ('a%d.beta' % i)).write_bytes(b'')
    betas = [str(tmp_path / ('a%d.beta' % i)) for i in range(3)]
    g = tmp_path / 'g.csv'
    g.write_text('# comment\nname,group,include\na0,X,True\na1,Y,False\na2,X,TRUE\n,Z,True\na1,,True\n')
    gf = B2T.groups_load_wrap(str(g), betas)
    assert gf.fname == ['a0', 'a2'] and gf.group == ['X', 'X'] and [op.basename(p) for p in gf.full_path] == ['a0.beta', 'a2.beta']
    g.write_text('name,group,include\na0,X,yes\n')
    with pytest.raises(B2B.IllegalArgumentError, match='Include column must be boolean'):
        B2T.groups_load_wrap(str(g), betas)
    g.write_text('name,grp\na0,X\n')
    with pytest.raises(B2B.IllegalArgumentError, match='column named "group"'):
        B2T.groups_load_wrap(str(g), betas)
    g.write_text('name,group\na0,X\nzz,Y\n')
    with pytest.raises(B2B.IllegalArgumentError, match='groups file mismatch'):
        B2T.groups_load_wrap(str(g), betas)
    gf = B2T.groups_load_wrap(None, betas + betas[:1])
    assert gf.fname == ['a0', 'a1', 'a2'] and gf.group == gf.fname


def test_blocks_file_rules(tmp_path):
    """load_blocks_file / is_block_file_nice corner cases (beta_to_blocks.py:24-91)."""
    def table(rows, header=None):
        p = str(tmp_path / ('t%d.bed' % len(rows)))
        with open(p, 'w') as f:
            if header:
                f.write(header + '\n')
            f.write('# a comment\n')
            for r in rows:
                f.write('\t'.join(map(str, r)) + '\n')
        return p
    good = [('chr1', 10, 20, 1, 3), ('chr1', 20, 40, 3, 7), ('chr1', 50, 60, 9, 10)]
    df = B2B.load_blocks_file(table(good, header='chr\tstart\tend\tstartCpG\tendCpG'))
    assert df.shape == (3, 5) and B2B.is_block_file_nice(df) == (True, '')
    for rows, msg in (([('chr1', 1, 2, 5, 5)] + good[2:], 'Some blocks are empty (startCpG==endCpG)'),
                      ([good[1], good[0], good[2]], 'startCpG is not monotonically increasing'),
                      ([good[0], good[0], good[1], good[2]][:3] + [good[2]], 'Some blocks are duplicated'),
                      ([('chr1', 10, 30, 1, 5), ('chr1', 20, 40, 3, 7)], 'Some blocks overlap')):
        assert B2B.is_block_file_nice(B2B.load_blocks_file(table(rows + [('chr9', 1, 2, 100 + len(rows), 200)]))) == (False, msg)
    with pytest.raises(B2B.IllegalArgumentError):
        B2B.load_blocks_file(table([('chr1', 10, 20)]))                       # fewer than 5 columns
    with pytest.raises(B2B.IllegalArgumentError):
        B2B.load_blocks_file(table([('chr1', 10, 20, 9, 3), ('chr1', 10, 20, 9, 13)]))    # endCpG < startCpG


# ---------------------------------------------------------------------------------------------------------
# the library's text paths (csrc/table_io.h) against the line-by-line Python they stand in for
# ---------------------------------------------------------------------------------------------------------
def _py_load(path, monkeypatch, **kw):
    monkeypatch.setenv('WGBSSEG_PY_TABLES', '1')
    try:
        return B2B.load_blocks_file(path, **kw)
    finally:
        monkeypatch.delenv('WGBSSEG_PY_TABLES')


def _same_table(a, b):
    assert len(a) == len(b) and a.chr == b.chr and a.start == b.start and a.end == b.end
    assert a.startCpG.tolist() == b.startCpG.tolist() and a.endCpG.tolist() == b.endCpG.tolist() and a.na.tolist() == b.na.tolist()
    assert a.columns == b.columns and a.cpg_text() == b.cpg_text()


def test_number_formatter_is_printf(monkeypatch):
    """wgbsseg_format_fixed == Python's '%.Nf' % v (== glibc printf): the exact binary value rounded half to even — random values,
    every k / 1000 and k / 256, exact ties at every digit count, the neighbours of ties, subnormals, and what falls outside the
    integer path (negative, > 1, infinities, -0.0, many digits)."""
    from wgbs_tools_amd import _lib
    rng = np.random.default_rng(3)
    ties = [k / 2.0 ** j for j in range(1, 12) for k in range(1, 2 ** j, 2)]
    near = [np.nextafter(t, 0) for t in ties[:200]] + [np.nextafter(t, 1) for t in ties[:200]]
    special = [0.0, 1.0, 0.005, 0.015, 0.995, 0.9949999999999999, 0.9950000000000001, 1e-300, 5e-324, 0.49999999999999994,
               float('nan'), -0.0, -0.5, 1.5, 1e22, float('inf'), float('-inf'), 2.675, 1.005, 123456.789, -1e-9]
    v = np.concatenate([rng.random(200000), rng.random(20000) * 1e-6, np.arange(1001) / 1000.0, np.arange(257) / 256.0, ties, near, special])
    for d in (0, 1, 2, 3, 4, 6, 9, 12):
        got = _lib.format_fixed(v, d)
        fmt = '%%.%df' % d
        want = ['NA' if x != x else fmt % x for x in v.tolist()]
        bad = [(x, g, w) for x, g, w in zip(v.tolist(), got, want) if g != w]
        assert not bad, (d, bad[:5])


def test_native_blocks_parser_and_writers_match_the_python_paths(world, tmp_path, monkeypatch, capfd):
    from wgbs_tools_amd import _lib
    # 1. plain tables: every fixture, and files with a header, comments, blank lines, NA spellings, more columns, no last newline
    files = dict(world['blocks'])
    extra = tmp_path / 'mixed.bed'
    extra.write_text('chr\tstart\tend\tstartCpG\tendCpG\n# c\n\n   \t \nchr1\t10\t20\t1\t3\nchr1\t20\t40\tNA\t7\nchr2\t5\t6\t\t\n'
                     'chr2\t7\t9\t10\tnan\tanno\tgene\nchrUn_gl000220\t000100\t200\t0012\t15\tx\nchr3\t1\t2\t100000000000000\t100000000000001')
    files['mixed'] = str(extra)
    import gzip
    gz = tmp_path / 'nice.bed.gz'
    gz.write_bytes(gzip.compress(open(world['blocks']['nice'], 'rb').read()))
    files['gz'] = str(gz)
    for name, p in files.items():
        a = B2B.load_blocks_file(p)
        assert a.parsed is not None, name                                     # the fast path took it
        b = _py_load(p, monkeypatch)
        assert b.parsed is None
        _same_table(a, b)
        assert B2B.is_block_file_nice(a) == B2B.is_block_file_nice(b)
        for nrows in (1, 3, 10 ** 9):
            _same_table(B2B.load_blocks_file(p, nrows=nrows), _py_load(p, monkeypatch, nrows=nrows))
        _same_table(a.rows(1, 4), b.rows(1, 4))
    # 2. what the fast path must decline (the Python parser then answers, with its messages)
    odd = {'crlf': 'chr1\t1\t2\t3\t4\r\nchr1\t2\t3\t4\t5\r\n', 'float': 'chr1\t1\t2\t3.0\t4\n', 'space': 'chr1\t1\t2\t 3\t4\n',
           'short_later': 'chr1\t1\t2\t3\t4\nchr1\t5\n', 'short_first': 'chr1\t1\t2\n', 'empty': '', 'only_comments': '# a\n# b\n',
           'utf8': 'chré\t1\t2\t3\t4\n', 'sci': 'chr1\t1\t2\t1e3\t2e3\n', 'neg': 'chr1\t1\t2\t-3\t4\n', 'long': 'chr1\t1\t2\t1234567890123456\t1234567890123457\n',
           'header_only': 'chr\tstart\tend\tstartCpG\tendCpG\n'}
    for name, text in odd.items():
        p = tmp_path / (name + '.bed')
        p.write_bytes(text.encode('utf-8'))
        assert _lib.blocks_parse(text.encode('utf-8')) is None, name
        res = []
        for py in (False, True):
            try:
                t = _py_load(str(p), monkeypatch) if py else B2B.load_blocks_file(str(p))
                res.append(('ok', t.chr, t.startCpG.tolist(), t.endCpG.tolist(), t.na.tolist()))
            except Exception as e:
                res.append((type(e).__name__, str(e)))
        assert res[0] == res[1], name
    capfd.readouterr()
    # 3. the table writer: file, chunks appended, standard output — against the Python loop (which the goldens pin above)
    df = B2B.load_blocks_file(str(extra))
    dfp = _py_load(str(extra), monkeypatch)
    rng = np.random.default_rng(1)
    for digits in (0, 2, 3, 5):
        vals = rng.random((len(df), 7))
        vals[rng.random(vals.shape) < 0.2] = np.nan
        vals[0, :4] = [0.0, 1.0, 0.125, 0.375]
        names = ['s%d' % i for i in range(7)]
        buf = io.StringIO()
        B2T.dump(buf, B2T.Table(dfp, names, vals), True, digits)
        want = buf.getvalue()
        out = str(tmp_path / ('t%d.tsv' % digits))
        B2T.dump(out, B2T.Table(df, names, vals), True, digits)
        assert open(out).read() == want
        for a in range(0, len(df), 2):                                        # in chunks, appended
            B2T.dump(out, B2T.Table(df.rows(a, a + 2), names, vals[a:a + 2]), a == 0, digits)
        assert open(out).read() == want
        B2T.dump(None, B2T.Table(df, names, vals), True, digits)              # standard output
        assert capfd.readouterr().out == want
    # a table large enough for several shards and threads
    n = 70001
    big = tmp_path / 'big.bed'
    with open(big, 'w') as f:
        for i in range(n):
            f.write('chr%d\t%d\t%d\t%d\t%d\n' % (1 + i % 22, 10 * i, 10 * i + 7, 1 + 2 * i, 3 + 2 * i))
    df = B2B.load_blocks_file(str(big))
    dfp = _py_load(str(big), monkeypatch)
    assert df.parsed is not None and B2B.is_block_file_nice(df) == B2B.is_block_file_nice(dfp) == (True, '')
    vals = np.round(rng.random((n, 3)), 3)
    vals[rng.random(vals.shape) < 0.1] = np.nan
    buf = io.StringIO()
    B2T.dump(buf, B2T.Table(dfp, ['a', 'b', 'c'], vals), True, 2)
    out = str(tmp_path / 'big.tsv')
    B2T.dump(out, B2T.Table(df, ['a', 'b', 'c'], vals), True, 2)
    assert open(out).read() == buf.getvalue()
    # 4. bedGraph rows from uint8 and uint16 pairs
    for dt in (np.uint8, np.uint16):
        top = np.iinfo(dt).max
        cov = rng.integers(0, top + 1, n)
        cov[rng.random(n) < 0.1] = 0
        meth = (cov * rng.random(n)).astype(np.int64)
        rows = np.stack([meth, cov], axis=1).astype(dt)
        pa, pb = str(tmp_path / 'a.bedGraph'), str(tmp_path / 'b.bedGraph')
        B2B.write_bedgraph(pa, df, rows)
        B2B.write_bedgraph(pb, dfp, rows)
        assert open(pa, 'rb').read() == open(pb, 'rb').read()


def test_random_block_tables_fast_path_never_disagrees(tmp_path, monkeypatch):
    """4,000 small random blocks tables over an alphabet of tricky fields: whatever the library's parser takes, it reads as the
    line-by-line parser does (rows, CpG columns, missing flags, annotation columns, the nice-table verdict, nrows); what it
    declines, the Python parser answers either way — same table or same exception from load_blocks_file."""
    rng = np.random.default_rng(12)
    cpg = ['1', '3', '7', '12', '007', '+5', '-3', '3.0', '1e1', ' 4', '4 ', 'NA', '', 'NaN', 'nan', 'N/A', 'NULL', 'None', 'x', '1234567890123456', '0']
    taken = 0
    for it in range(4000):
        n_rows = int(rng.integers(0, 6))
        rows = []
        s0 = 1
        for r in range(n_rows):
            w = int(rng.choice([5, 5, 5, 5, 6, 7, 7, 3, 8]))
            row = ['chr%d' % rng.integers(1, 4), str(int(rng.integers(1, 9999))), str(int(rng.integers(1, 9999)))]
            if rng.random() < 0.7:
                e0 = s0 + int(rng.integers(0, 5))
                row += [str(s0), str(e0)]
                s0 = e0 if rng.random() < 0.8 else e0 - 1
            else:
                row += [str(rng.choice(cpg)), str(rng.choice(cpg))]
            row += ['anno%d' % r, 'GENE', 'more'][:max(0, w - 5)]
            rows.append(row[:w])
        text = '\n'.join('\t'.join(r) for r in rows) + ('\n' if rng.random() > 0.2 else '')
        if rng.random() < 0.1:
            text = '# comment\n\n' + text
        if rng.random() < 0.1:
            text = 'chr\tstart\tend\tstartCpG\tendCpG' + ('\tanno\tgene' if rng.random() < 0.5 else '') + '\n' + text
        p = str(tmp_path / 't.bed')
        with open(p, 'w') as f:
            f.write(text)
        for anno in (False, True):
            res = []
            for py in (False, True):
                if py:
                    monkeypatch.setenv('WGBSSEG_PY_TABLES', '1')
                try:
                    t = B2B.load_blocks_file(p, anno=anno, nrows=None if it % 3 else 2)
                    res.append(('ok', t.chr, t.start, t.end, t.startCpG.tolist(), t.endCpG.tolist(), t.na.tolist(), t.columns,
                                {k: list(t.extra[k]) for k in t.extra}, B2B.is_block_file_nice(t) if len(t) else None))
                    if not py and t.parsed is not None:
                        taken += 1
                except Exception as e:
                    res.append((type(e).__name__, str(e)))
                if py:
                    monkeypatch.delenv('WGBSSEG_PY_TABLES')
            assert res[0] == res[1], (it, anno, text)
    assert taken > 1000, taken


def test_comments_inside_a_line_and_pandas_missing_values(tmp_path):
    """The reference reads a blocks table with pd.read_csv(sep='\\t', comment='#') (beta_to_blocks.py:50-91): a '#' ANYWHERE ends the
    parsed part of its line, and the CpG columns know pandas' whole vocabulary of missing values.  Checked against pandas itself."""
    pd = pytest.importorskip('pandas')
    from wgbs_tools_amd.genome import IllegalArgumentError
    text = ('chr1\t100\t200\t5\t9\tanno # trailing words\there\n'
            '# a whole line\n'
            'chr1\t300\t400\t9\t12# glued to the number\n'
            'chr1\t500\t600\t-nan\t14\n'
            'chr1\t700\t800\t14\t-NaN# a comment glued to a missing value\n'
            'chr2\t10\t20\t20\t25\n')
    p = tmp_path / 'c.bed'
    p.write_text(text)
    t = B2B.load_blocks_file(str(p))
    df = pd.read_csv(str(p), sep='\t', header=None, comment='#', usecols=range(5), names=['chr', 'start', 'end', 'startCpG', 'endCpG'])
    assert len(t) == len(df) == 5
    na = df['startCpG'].isna() | df['endCpG'].isna()
    assert t.na.tolist() == na.tolist() == [False, False, True, True, False]
    assert t.startCpG[~t.na].tolist() == df['startCpG'][~na].astype(int).tolist() == [5, 9, 20]
    assert t.endCpG[~t.na].tolist() == df['endCpG'][~na].astype(int).tolist() == [9, 12, 25]
    q = tmp_path / 'd.bed'
    q.write_text('chr1\t100\t200\tfive\t9\n')
    with pytest.raises(IllegalArgumentError):
        B2B.load_blocks_file(str(q))


def test_trim_rescale_is_an_integer_division():
    """utils_wgbs.py:277-290 trim_to_uint8 rescales a saturated pair with three float64 operations, trunc(fl(fl(m / c) * 255)).  Round 5's block reduction
    computes floor(255 m / c) in integers instead (csrc/seg_kernels.h wg_rescale_255: a float32 estimate pushed down by 1e-4, one exact correction).
    (1) the two are the same number: every exact quotient K / 255 (and K / 65535) survives the two roundings, and random / structured pairs agree;
    (2) the device's arithmetic, restated in numpy float32 with the reciprocal off by one ulp EITHER way (v_rcp_f32's accuracy), returns that floor."""
    # (1a) exact quotients: m / c = K / M as a real number => fl(m / c) = fl(K / M): 256 (65536) cases cover every such pair
    for M in (255, 65535):
        K = np.arange(M + 1, dtype=np.float64)
        assert np.array_equal(np.trunc(K / np.float64(M) * np.float64(M)), K)
    rng = np.random.default_rng(17)
    # (1b) + (2): random pairs over the whole range the block reduction can meet (c <= 65535 sites x 255), and the structured ones
    c = np.concatenate([rng.integers(256, 65535 * 255 + 1, 3000000), rng.integers(256, 300000, 3000000), rng.integers(256, 2000, 500000)]).astype(np.int64)
    m = (rng.random(c.size) * (c + 1)).astype(np.int64).clip(0, c)
    g = rng.integers(2, 65535, 400000).astype(np.int64)
    kq = rng.integers(0, 256, g.size).astype(np.int64)
    c = np.concatenate([c, 255 * g, 255 * g, g[g > 255], g[g > 255], 255 * g])            # exact quotients, m = 0, m = c, one off an exact quotient
    m = np.concatenate([m, kq * g, kq * g + (kq < 255), np.zeros((g > 255).sum(), dtype=np.int64), g[g > 255], np.maximum(kq * g - 1, 0)])
    want = np.trunc(m.astype(np.float64) / c.astype(np.float64) * np.float64(255.0)).astype(np.int64)
    assert np.array_equal(want, 255 * m // c)
    N = (255 * m).astype(np.uint32)
    Nf = N.astype(np.float32)
    cf = c.astype(np.float32)
    assert np.array_equal(cf.astype(np.int64), c)                                       # c < 2^24: exact in float
    rc0 = np.float32(1.0) / cf
    for rcv in (rc0, np.nextafter(rc0, np.float32(0)), np.nextafter(rc0, np.float32(1))):
        est = (Nf.astype(np.float64) * rcv.astype(np.float64) + np.float64(np.float32(-1.0e-4))).astype(np.float32)   # the fused multiply-add: one rounding
        q = np.where(est > 0, np.floor(est), 0).astype(np.int64)
        assert ((q == want) | (q == want - 1)).all()                                    # the biased estimate never overshoots
        r = N.astype(np.int64) - q * c
        assert (r >= 0).all()
        got = q + (r >= c)
        assert np.array_equal(got, want)
