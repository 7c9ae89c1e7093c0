"""The drop-in `segmentor` executable on the GPU: same command line, same stdin, same stdout bytes as the reference binary
(oracle/_ref/segmentor where it is built; the oracle's restatement otherwise) — what an unmodified segment.py:41-59 would parse."""
import os.path as op
import subprocess

import numpy as np
import pytest

from oracle import oracle
from wgbs_tools_amd import build as nbuild, synth

pytestmark = pytest.mark.gpu
BIN = op.join(nbuild.CSRC, 'segmentor')

# (seed, sites in the files, samples, start, n, max_cpg, max_bp, pcount, islands)
WORLDS = [
    (1, 9000, 3, 0, 9000, 1000, 2000, 15.0, False),
    (2, 12000, 1, 2500, 7000, 1000, 2000, 15.0, False),      # a range inside longer files, one sample
    (3, 6000, 4, 777, 4000, 40, 500, 1.0, False),            # the binary's default pseudo count
    (4, 8000, 2, 100, 6000, 1000, 5000, 15.0, True),         # islands: windows of several hundred sites
    (5, 3000, 2, 2999, 1, 1000, 2000, 15.0, False),          # one site
    (6, 5000, 5, 0, 5000, 300, 600, 0.5, False),             # pseudo count below 1: the guarded term forms
]


@pytest.mark.parametrize('seed,total,ns,start,n,max_cpg,max_bp,pc,islands', WORLDS)
def test_same_stdout_as_the_reference_binary(tmp_path, seed, total, ns, start, n, max_cpg, max_bp, pc, islands):
    nbuild.build()
    paths = []
    for s in range(ns):
        p = str(tmp_path / ('sample%d.beta' % s))
        synth.synth_betas(seed, s, 0, total).tofile(p)
        paths.append(p)
    loci = synth.synth_loci(seed, [total], islands=islands)[start:start + n]
    text = ('\n'.join(str(int(x)) for x in loci) + '\n').encode()
    opts = ['-s', str(start), '-n', str(n), '-max_cpg', str(max_cpg), '-ps', repr(pc), '-max_bp', str(max_bp)]
    # the reference finds options and files anywhere on the line: files first (segment.py:49-51) and options first
    for argv in (paths + opts, opts[:4] + paths + opts[4:]):
        got = subprocess.run([BIN] + argv, input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert got.returncode == 0, got.stderr.decode()[-2000:]
        if oracle.have_ref():
            ref = subprocess.run([oracle.REF_BIN] + argv, input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            assert ref.returncode == 0
            assert got.stdout == ref.stdout
        else:
            slices = [np.fromfile(p, dtype=np.uint8).reshape(-1, 2)[start:start + n] for p in paths]
            want = oracle.segment_chunk(slices, loci, pc, max_cpg, max_bp)
            assert got.stdout == (''.join('%d ' % b for b in want) + '\n').encode()
