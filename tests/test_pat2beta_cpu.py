"""pat2beta without a GPU: the oracle's restatement against the reference's own stdin2beta binary
and the committed digests (tests/golden/make_golden_pat.py), the host-side chunking and file-name rules."""
import gzip
import hashlib
import json
import os.path as op

import numpy as np
import pytest

from oracle import pat2beta_oracle as OP
from wgbs_tools_amd import pat2beta as P2B, synth

HERE = op.dirname(op.abspath(__file__))


@pytest.fixture(scope='module')
def pat_golden():
    return json.load(open(op.join(HERE, 'golden', 'pat_cases.json')))


@pytest.mark.parametrize('name', ['small', 'sparse', 'deep'])
def test_restatement_matches_reference_digests(name, pat_golden):
    rec = pat_golden[name]
    spec = rec['spec']
    lines = synth.synth_pat_lines(spec['seed'], spec['n_sites'], spec['n_reads'])
    text = ('\n'.join(lines) + '\n').encode()
    assert hashlib.sha1(text).hexdigest() == rec['text_sha1'], 'synthetic pat generator drifted'
    arr = OP.counts(lines, 1, spec['n_sites'] + 1)
    assert int(arr[:, 1].max()) == rec['max_cov']
    for lbeta, tag in ((False, 'beta'), (True, 'lbeta')):
        b = OP.trim(arr, lbeta)
        assert hashlib.sha1(b.tobytes()).hexdigest() == rec[tag + '_sha1'] and b[:16].tolist() == rec[tag + '_head']
    if OP.have_ref():                                       # live, where the binary travelled
        assert np.array_equal(OP.ref_counts(text, 1, spec['n_sites'] + 1), arr)
        sub = OP.ref_counts(text, 100, 130) if spec['n_sites'] > 130 else None
        if sub is not None:
            assert np.array_equal(sub, OP.counts(lines, 100, 130))


def test_chunks_end_on_lines_and_names(tmp_path):
    lines = synth.synth_pat_lines(5, 500, 3000)
    text = ('\n'.join(lines)).encode()                      # no trailing newline
    p = tmp_path / 'x.pat'
    p.write_bytes(text)
    got = b''.join(P2B.pat_chunks(str(p), chunk_bytes=4096))
    assert got == text + b'\n'
    assert all(c.endswith(b'\n') for c in P2B.pat_chunks(str(p), chunk_bytes=777))
    g = tmp_path / 'y.pat.gz'
    with gzip.open(g, 'wb') as f:
        f.write(text + b'\n')
    assert b''.join(P2B.pat_chunks(str(g), chunk_bytes=5000)) == text + b'\n'
    assert P2B.splitextgz('a/b/s1.pat.gz') == ('a/b/s1', '.pat.gz') and P2B.splitextgz('s2.pat') == ('s2', '.pat')


def _bgzf_bytes(text, block=65280, level=6):
    """what `bgzip` writes (SAM specification 4.1): independent gzip members with a 'BC' extra subfield, and the empty EOF block"""
    import struct
    import zlib
    out = []
    for a in list(range(0, len(text), block)) + [None]:
        piece = b'' if a is None else text[a:a + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        payload = c.compress(piece) + c.flush()
        bsize = 12 + 6 + len(payload) + 8
        out.append(b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff' + struct.pack('<H', 6) + b'BC' + struct.pack('<HH', 2, bsize - 1) + payload +
                   struct.pack('<II', zlib.crc32(piece) & 0xffffffff, len(piece)))
    return b''.join(out)


def test_bgzf_pat_files_are_inflated_block_by_block(tmp_path, monkeypatch):
    """.pat.gz files are BGZF (bgzip): the blocks are inflated on a pool of threads; same text, in line-aligned chunks, as one gzip
    stream gives — small blocks, blocks that straddle the reader's buffer, an empty file, plain gzip and multi-member gzip (not
    BGZF: the gzip module's path), and a truncated file."""
    lines = synth.synth_pat_lines(7, 500, 40000)
    text = ('\n'.join(lines) + '\n').encode()
    assert len(text) > 600000
    for name, data in (('full', _bgzf_bytes(text)), ('small_blocks', _bgzf_bytes(text, block=1500)), ('empty', _bgzf_bytes(b''))):
        p = tmp_path / (name + '.pat.gz')
        p.write_bytes(data)
        assert gzip.open(p, 'rb').read() == (text if name != 'empty' else b'')            # it IS a gzip file
        for rb in (1 << 20, 70000, 4096 + 13):                                            # reader buffers smaller than a block too
            got = b''.join(b for b in P2B.bgzf_pieces(str(p), read_bytes=rb, threads=4))
            assert got == (text if name != 'empty' else b''), (name, rb)
        chunks = list(P2B.pat_chunks(str(p), chunk_bytes=100000))
        assert b''.join(chunks) == (text if name != 'empty' else b'') and all(c.endswith(b'\n') for c in chunks)
        monkeypatch.setenv('WGBSSEG_PY_GUNZIP', '1')
        assert b''.join(P2B.pat_chunks(str(p), chunk_bytes=100000)) == b''.join(chunks)
        monkeypatch.delenv('WGBSSEG_PY_GUNZIP')
    notail = tmp_path / 'notail.pat.gz'
    notail.write_bytes(_bgzf_bytes(text[:-1]))
    assert b''.join(P2B.pat_chunks(str(notail))) == text
    plain = tmp_path / 'plain.pat.gz'
    plain.write_bytes(gzip.compress(text[:300000]) + gzip.compress(text[300000:]))      # two ordinary members
    assert list(P2B.bgzf_pieces(str(plain))) == []
    assert b''.join(P2B.pat_chunks(str(plain), chunk_bytes=50000)) == text
    # BGZF blocks followed by ordinary gzip members (`cat` of files from different writers): the reference's `gunzip -cd` reads it
    # (pat2beta.py:30), so the rest goes to zlib's multi-member inflater instead of an error
    mixed = tmp_path / 'mixed.pat.gz'
    mixed.write_bytes(_bgzf_bytes(text[:200000], block=30000) + gzip.compress(text[200000:450000]) + gzip.compress(text[450000:]))
    for rb in (1 << 20, 70000):
        assert b''.join(P2B.bgzf_pieces(str(mixed), read_bytes=rb, threads=2)) == text, rb
    assert b''.join(P2B.pat_chunks(str(mixed), chunk_bytes=100000)) == text
    cut = tmp_path / 'cut.pat.gz'
    whole = _bgzf_bytes(text)
    cut.write_bytes(whole[:len(whole) // 2 + 7])
    with pytest.raises(Exception):
        b''.join(P2B.pat_chunks(str(cut)))


def test_truncated_or_corrupt_plain_gzip_member_behind_bgzf_blocks_is_an_error(tmp_path):
    """the multi-member inflater that takes over behind the BGZF blocks used to return a partial stream when its last
    member stopped in the middle, and let zlib.error escape on trailing junk; both are IllegalArgumentError now, like the pure-BGZF
    path's 'truncated or not BGZF'."""
    from wgbs_tools_amd.genome import IllegalArgumentError
    text = b''.join(b'chr1\t%d\tCCTT\t1\n' % i for i in range(60000))
    head = _bgzf_bytes(text[:200000], block=30000)
    member = gzip.compress(text[200000:])
    cut = tmp_path / 'cut_member.pat.gz'
    cut.write_bytes(head + member[:len(member) // 2])
    for rb in (1 << 20, 70000):
        with pytest.raises(IllegalArgumentError, match='truncated'):
            b''.join(P2B.bgzf_pieces(str(cut), read_bytes=rb, threads=2))
    junk = tmp_path / 'junk.pat.gz'
    junk.write_bytes(head + member + b'\x1f\x8b\x08\x00 this is not a deflate stream at all')
    with pytest.raises(IllegalArgumentError, match='Invalid gzip data'):
        b''.join(P2B.bgzf_pieces(str(junk), threads=2))
    whole = tmp_path / 'whole.pat.gz'                              # the stream may END between two members: no error
    whole.write_bytes(head + member)
    assert b''.join(P2B.bgzf_pieces(str(whole), threads=2)) == text
