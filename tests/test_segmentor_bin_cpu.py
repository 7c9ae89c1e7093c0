"""The drop-in `segmentor` executable (wgbs_tools_amd/csrc/segmentor_main.cpp: the reference's per-chunk command line over the C ABI):
everything it decides BEFORE it needs a GPU — usage, mandatory options, the beta-file token rule, the loci count, the #meth <= #cov
check, ranges a file does not hold — beside the reference binary where that one is built (oracle/_ref/segmentor, build container)."""
import os
import os.path as op
import subprocess

import numpy as np
import pytest

from oracle import oracle
from wgbs_tools_amd import build as nbuild, synth

BIN = op.join(nbuild.CSRC, 'segmentor')


def run(binary, args, stdin=b''):
    return subprocess.run([binary] + [str(a) for a in args], input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)


@pytest.fixture(scope='module')
def world(tmp_path_factory):
    d = tmp_path_factory.mktemp('segbin')
    n = 500
    paths = []
    for s in range(2):
        p = str(d / ('s%d.beta' % s))
        synth.synth_betas(11, s, 0, n).tofile(p)
        paths.append(p)
    loci = synth.synth_loci(11, [n])
    return dict(dir=d, n=n, paths=paths, loci=loci, text=('\n'.join(str(int(x)) for x in loci) + '\n').encode())


def test_built_and_listed():
    nbuild.build()
    assert op.isfile(BIN) and os.access(BIN, os.X_OK)


def test_usage_line_and_status_like_the_reference():
    mine = run(BIN, ['a.beta', '-s', '0'])
    assert mine.returncode == 255 and mine.stdout == b''
    assert mine.stderr.decode() == 'Usage: segment BETA_PATH [BETA_PATH...] -s START -n NR_SITES  [-m max_cpg] [-ps PSEUDO_COUNT]\n'
    if oracle.have_ref():
        ref = run(oracle.REF_BIN, ['a.beta', '-s', '0'])
        assert (ref.returncode, ref.stdout, ref.stderr) == (mine.returncode, mine.stdout, mine.stderr)


@pytest.mark.parametrize('args,msg', [
    (['a.beta', '-n', '5', '-ps', '1', '-max_bp', '9'], 'start sites (-s) must be provided'),
    (['a.beta', '-s', '5', '-ps', '1', '-max_bp', '9'], 'number of sites (-n) must be provided'),
    (['a.beta', '-ps', '1', '-max_bp', '9', '-n', '4', '-s'], 'start sites (-s) must be provided'),        # an option that ends the line has no value
])
def test_mandatory_options(args, msg):
    r = run(BIN, args)
    assert r.returncode == 1 and r.stdout == b'' and r.stderr.decode().strip() == msg
    if oracle.have_ref():                                       # (the reference throws the same literal; nothing catches it)
        assert run(oracle.REF_BIN, args).returncode != 0


def test_empty_chunk_prints_its_start(world):
    args = world['paths'] + ['-s', 3, '-n', 0, '-max_cpg', 100, '-ps', 15, '-max_bp', 2000]
    r = run(BIN, args)
    assert (r.returncode, r.stdout) == (0, b'0 \n')
    if oracle.have_ref():
        ref = run(oracle.REF_BIN, args)
        assert (ref.returncode, ref.stdout) == (0, r.stdout)


def test_loci_count_must_match(world):
    args = world['paths'] + ['-s', 0, '-n', world['n'], '-max_cpg', 100, '-ps', 15, '-max_bp', 2000]
    short = b'\n'.join(world['text'].split(b'\n')[:-3]) + b'\n'
    r = run(BIN, args, short)
    want = 'Error: nr_sites != number of loci: %d != %d. Try different chunck size!' % (world['n'], world['n'] - 2)
    assert r.returncode == 1 and r.stdout == b'' and r.stderr.decode().strip() == want
    if oracle.have_ref():
        ref = run(oracle.REF_BIN, args, short)
        assert ref.returncode != 0 and ref.stderr.decode().splitlines()[0] == want


def test_meth_above_cov_is_reported_in_the_reference_s_words(world):
    bad = str(world['dir'] / 'bad.beta')
    rows = synth.synth_betas(11, 5, 0, world['n']).copy()
    rows[123] = (9, 4)
    rows.tofile(bad)
    args = [world['paths'][0], bad, '-s', 100, '-n', 200, '-max_cpg', 100, '-ps', 15, '-max_bp', 2000]
    text = b'\n'.join(world['text'].split(b'\n')[100:300]) + b'\n'
    r = run(BIN, args, text)
    lines = r.stderr.decode().splitlines()
    assert r.returncode == 1 and r.stdout == b'' and lines[:2] == ['invalid data, i = 23. data: 9, 4', 'beta path: ' + bad]
    if oracle.have_ref():
        ref = run(oracle.REF_BIN, args, text)
        assert ref.returncode != 0 and ref.stderr.decode().splitlines()[:2] == lines[:2]


def test_tokens_that_are_not_beta_files_are_not_read(world):
    """main.cpp:101-107: only arguments of six or more characters ending in .beta are files — `x.bet`, `.beta` (five characters) and
    option values are not; with no file left there is nothing to segment."""
    r = run(BIN, ['.beta', 'x.bet', '-s', 0, '-n', 5, '-ps', 1, '-max_bp', 9], b'1\n2\n3\n4\n5\n')
    assert r.returncode == 1 and 'no beta file' in r.stderr.decode()


def test_ranges_the_file_does_not_hold_and_max_bp_zero_are_refused(world):
    args = world['paths'] + ['-s', world['n'] - 10, '-n', 50, '-max_cpg', 100, '-ps', 15, '-max_bp', 2000]
    r = run(BIN, args, b'1\n' * 50)
    assert r.returncode == 1 and 'the file ends before site' in r.stderr.decode() and r.stdout == b''
    r = run(BIN, world['paths'] + ['-s', 0, '-n', 50, '-max_cpg', 100, '-ps', 15], b'1\n' * 50)
    assert r.returncode == 1 and '-max_bp must be at least 1' in r.stderr.decode()
    r = run(BIN, world['paths'] + ['-s', 'x', '-n', 50, '-max_cpg', 100, '-ps', 15], b'')
    assert r.returncode == 1 and 'invalid value for -s' in r.stderr.decode()
