"""pat2beta on the GPU (k_pat_count / k_pat_trim behind wgbsseg_patbeta_*) against the reference's stdin2beta binary, the
committed digests and the oracle."""
import gzip
import hashlib
import json
import os.path as op

import numpy as np
import pytest

from oracle import pat2beta_oracle as OP
from wgbs_tools_amd import _lib, synth, wgbs_tools
from test_pat2beta_cpu import pat_golden          # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['small', 'sparse', 'deep'])
def test_cli_matches_reference(name, pat_golden, tmp_path):
    rec = pat_golden[name]
    spec = rec['spec']
    lines = synth.synth_pat_lines(spec['seed'], spec['n_sites'], spec['n_reads'])
    ref = synth.write_genome(str(tmp_path / 'references' / 'g'), ['chr1'], [spec['n_sites']], (np.arange(spec['n_sites']) * 37 + 100).astype(np.uint32))
    pat = tmp_path / 'smp.pat.gz'
    with gzip.open(pat, 'wb') as f:
        f.write(('\n'.join(lines) + '\n\n').encode())                  # an empty line at the end: skipped
    for lbeta, tag in ((False, 'beta'), (True, 'lbeta')):
        assert wgbs_tools.main(['wgbstools', 'pat2beta', str(pat), '-o', str(tmp_path), '--genome', ref, '-f'] + (['-l'] if lbeta else [])) == 0
        got = open(str(tmp_path / ('smp.' + tag)), 'rb').read()
        assert hashlib.sha1(got).hexdigest() == rec[tag + '_sha1'], (name, tag)
    # the beta file feeds straight into segment's sanity check: 2 bytes per site
    assert op.getsize(str(tmp_path / 'smp.beta')) == 2 * spec['n_sites']


def test_ranges_chunks_and_malformed_lines():
    """sub-ranges (reads hanging over both ends), many small chunks, and the reference's failure cases"""
    lines = synth.synth_pat_lines(21, 5000, 40000)
    text = ('\n'.join(lines) + '\n').encode()
    for start, end in ((1, 5001), (1000, 1200), (4990, 5001), (1, 2)):
        with _lib.PatBeta(start, end) as pb:
            pos = 0
            while pos < len(text):                                     # ragged chunks cut at line ends
                cut = text.find(b'\n', min(len(text) - 1, pos + 3000)) + 1
                pb.feed(text[pos:cut])
                pos = cut
            got = pb.finish(lbeta=True)
        want = OP.ref_counts(text, start, end) if OP.have_ref() else OP.counts(lines, start, end)
        assert np.array_equal(got.astype(np.int64), OP.trim(want, True).astype(np.int64)), (start, end)
    for bad in (b'chr1\t5\tCT\n', b'chr1\t5\tCT\t\n', b'chr1\tx\tCT\t3\n', b'chr1\t5\tCT\t\t7\n'):
        with _lib.PatBeta(1, 100) as pb:
            pb.feed(b'chr1\t3\tCC\t2\n' + bad)
            with pytest.raises(_lib.SegmentorError, match='failed calculating beta: invalid line at byte offset 12'):
                pb.finish()
        if OP.have_ref():
            assert OP.ref_counts(b'chr1\t3\tCC\t2\n' + bad, 1, 100) is None
    with _lib.PatBeta(1, 100) as pb:
        with pytest.raises(_lib.SegmentorError, match='complete line'):
            pb.feed(b'chr1\t3\tCC\t2')
        pb.feed(b'chr1\t3\tCC\t2\textra\tcolumns\n\nchr1\t 4\tH.T\t+3\n')       # extra columns, empty line, stoi's leading blank and sign
        got = pb.finish()
    want = np.zeros((99, 2), dtype=np.uint8)
    want[2] = (2, 2); want[3] = (5, 5); want[5] = (0, 3)
    assert np.array_equal(got, want)


def test_lines_against_the_tiles_of_the_counting_kernel():
    """Round 5's k_pat_count stages 4096-byte tiles (+ 1024 bytes behind them) into LDS and parses one line per thread: lines longer than
    the staged bytes (reads of thousands of CpGs), lines across tile boundaries at every phase, runs of empty lines, two-byte-dense lines
    (a line start at every 8th byte), a chunk that ends inside a tile's last 16 bytes — against the reference binary / the oracle."""
    rng = np.random.default_rng(5)
    n_sites = 20000
    alphabet = np.frombuffer(b'CCCTTTH.', dtype=np.uint8)
    lines = []
    pos = 1
    for k in range(6000):
        kind = k % 7
        if kind == 0:
            ln = int(rng.integers(1100, 5200))               # longer than the tile's overhang, some longer than a tile
        elif kind in (1, 2):
            ln = 1
        else:
            ln = int(rng.integers(1, 40))
        pat = alphabet[rng.integers(0, 8, ln)].tobytes().decode()
        lines.append('c\t%d\t%s\t%d' % (pos, pat, int(rng.integers(1, 30))))
        if kind == 3:
            lines += [''] * int(rng.integers(1, 40))         # empty lines: skipped (stdin2beta.cpp:100)
        pos = int(min(n_sites - 5, pos + rng.integers(0, 9)))
    text = ('\n'.join(lines) + '\n').encode()
    assert len(text) > 40 * 4096
    want_counts = OP.ref_counts(text, 1, n_sites + 1) if OP.have_ref() else OP.counts([l for l in lines if l], 1, n_sites + 1)
    want = OP.trim(want_counts, True).astype(np.int64)
    for pieces in (1, 7):                                    # one chunk, and chunks cut at line ends (every chunk begins a new tile grid)
        with _lib.PatBeta(1, n_sites + 1) as pb:
            cuts = [0] + [text.find(b'\n', len(text) * q // pieces) + 1 for q in range(1, pieces)] + [len(text)]
            for a, b in zip(cuts[:-1], cuts[1:]):
                pb.feed(text[a:b])
            assert pb.kernel_ms() > 0
            got = pb.finish(lbeta=True)
        assert np.array_equal(got.astype(np.int64), want), pieces
    # a malformed line deep inside a later tile is reported with its byte offset in the whole input
    bad_at = text.find(b'\n', 9 * 4096 + 100) + 1
    broken = text[:bad_at] + b'c\t77\tCT\n' + text[bad_at:]
    with _lib.PatBeta(1, n_sites + 1) as pb:
        pb.feed(broken)
        with pytest.raises(_lib.SegmentorError, match='invalid line at byte offset %d ' % bad_at):
            pb.finish()
