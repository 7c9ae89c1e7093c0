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
