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
