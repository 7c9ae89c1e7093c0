"""Parity fuzz INSIDE the GPU suite (round 3; VERDICT r02 weak item 1: a round-1 defect survived 150,000 uniformly drawn chunks and
was found by an offline fuzz only).  Three parts:
  * a time-boxed slice of the boundary-aligned fuzz (tests/fuzzlib.py; tools/aligned_fuzz.py runs the same draws for as long as one
    likes): chunk starts / ends / lengths on and next to multiples of 16 / 64 / 128 / 256, window limits 59 .. 130 and 193 / 300 / 1000,
    random pseudo counts — against the oracle's C restatement;
  * 64 more seeds of the uniform random-parameter test (test_13 of test_gpu_parity.py);
  * a subset of the aligned draws against the REFERENCE BINARY itself (oracle/_ref/segmentor, `-s start -n len` inside a larger
    world, exactly as segment.py:48-53 runs it), not only its restatement.
WGBSSEG_FUZZ_SECONDS (default 60) sets the time box, WGBSSEG_FUZZ_FIRST (default 200000) the first seed of the slice."""
import os
import os.path as op
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import fuzzlib
import test_gpu_parity as T
from oracle import oracle
from wgbs_tools_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def seg():
    s = _lib.Segmenter(0)
    yield s
    s.close()


def test_aligned_fuzz_time_boxed(seg):
    seconds = float(os.environ.get('WGBSSEG_FUZZ_SECONDS', '60'))
    first = int(os.environ.get('WGBSSEG_FUZZ_FIRST', '200000'))
    done, chunks, bad = fuzzlib.run_aligned(seg, oracle, first, 1000000, seconds, os.cpu_count() or 1)
    print('aligned fuzz: seeds %d .. %d, %d chunks, %d differences' % (first, first + done - 1, chunks, len(bad)))
    assert not bad, '%d of %d boundary-aligned chunks differ from the oracle; first: %s' % (len(bad), chunks, bad[0])
    assert chunks >= 480, 'the time box of %.0f s covered only %d chunks' % (seconds, chunks)


@pytest.mark.parametrize('seed', range(16, 80))
def test_13c_more_uniform_seeds(seg, seed):
    T.test_13_random_parameters_and_adversarial_inputs_match_oracle(seg, seed)


@pytest.mark.skipif(not oracle.have_ref(), reason='oracle/_ref/segmentor not built (needs /root/reference at build time)')
@pytest.mark.parametrize('seed', range(300000, 300006))
def test_aligned_fuzz_against_the_reference_binary(seg, seed):
    """The same kind of draw, checked by the reference's own binary run the way the driver runs it: whole-world .beta files,
    `-s start0 -n len`, the chunk's loci on stdin."""
    rng, slices, loci = fuzzlib.aligned_world(seed)
    n = loci.size
    seg.set_betas(slices)
    seg.set_loci(loci)
    with tempfile.TemporaryDirectory(dir='/dev/shm' if op.isdir('/dev/shm') else None) as td:
        paths = []
        for i, s in enumerate(slices):
            p = op.join(td, 's%04d.beta' % i)
            np.ascontiguousarray(s, dtype=np.uint8).tofile(p)
            paths.append(p)
        for draw in range(2):
            pcount, max_cpg, max_bp, starts, lens = fuzzlib.aligned_draw(rng, n)
            got = seg.segment_chunks(starts, lens, pcount, max_cpg, max_bp)
            with ThreadPoolExecutor(max_workers=min(12, os.cpu_count() or 1)) as ex:
                want = list(ex.map(lambda sl: oracle.ref_segment_chunk(paths, sl[0], sl[1], loci[sl[0]:sl[0] + sl[1]], pcount, max_cpg, max_bp, timeout=300),
                                   zip(starts, lens)))
            for c, (a, b) in enumerate(zip(got, want)):
                assert a.tolist() == b.tolist(), 'seed %d draw %d samples %d pcount %r max_cpg %d max_bp %d chunk [%d,+%d) vs the reference binary: %s' % (
                    seed, draw, len(slices), pcount, max_cpg, max_bp, starts[c], lens[c], fuzzlib.first_diff(a, b))
