"""Parity fuzz INSIDE the GPU suite (round 3; VERDICT r02 weak item 1: a round-1 defect survived 150,000 uniformly drawn chunks and
was found by an offline fuzz only).  Three parts:
  * a time-boxed slice of the boundary-aligned fuzz (tests/fuzzlib.py; tools/aligned_fuzz.py runs the same draws for as long as one
    likes): chunk starts / ends / lengths on and next to multiples of 16 / 64 / 128 / 256, window limits 59 .. 130 and 193 / 300 / 1000,
    random pseudo counts — against the oracle's C restatement;
  * 64 more seeds of the uniform random-parameter test (test_13 of test_gpu_parity.py);
  * a subset of the aligned draws against the REFERENCE BINARY itself (oracle/_ref/segmentor, `-s start -n len` inside a larger
    world, exactly as segment.py:48-53 runs it), not only its restatement.
  * (round 4) a time-boxed slice of DEEP-mode draws — max_cpg 2000 .. 5000 on chunks of 8-12 k sites, 1 / 3 / 40 samples: the wide
    scoring tiles, the <15,32> recurrence and its ring, which the draws above (windows of a few hundred sites) never reach — against
    the oracle's many-thread restatement.
WGBSSEG_FUZZ_SECONDS (default 45) sets the time box of the aligned slice, WGBSSEG_DEEP_FUZZ_SECONDS (default 35) that of the deep one;
the first seeds follow the build round (fuzzlib.round_number(): every round's suite covers new seeds) unless WGBSSEG_FUZZ_FIRST names one."""
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
    seconds = float(os.environ.get('WGBSSEG_FUZZ_SECONDS', '45'))
    first = int(os.environ.get('WGBSSEG_FUZZ_FIRST', str(200000 + 10000 * fuzzlib.round_number())))
    done, chunks, bad = fuzzlib.run_aligned(seg, oracle, first, 1000000, seconds, os.cpu_count() or 1)
    print('aligned fuzz: seeds %d .. %d, %d chunks, %d differences' % (first, first + done - 1, chunks, len(bad)))
    assert not bad, '%d of %d boundary-aligned chunks differ from the oracle; first: %s' % (len(bad), chunks, bad[0])
    assert chunks >= 360, 'the time box of %.0f s covered only %d chunks' % (seconds, chunks)


def test_deep_fuzz_time_boxed(seg):
    import time
    seconds = float(os.environ.get('WGBSSEG_DEEP_FUZZ_SECONDS', '35'))
    first = int(os.environ.get('WGBSSEG_FUZZ_FIRST', str(1000 * fuzzlib.round_number())))
    t0, done, bad = time.time(), 0, []
    for seed in range(first, first + 1000):
        if time.time() - t0 > seconds:
            break
        slices, loci, pcount, max_cpg, max_bp, starts, lens = fuzzlib.deep_draw(seed)
        seg.set_betas(slices)
        seg.set_loci(loci)
        got = seg.segment_chunks(starts, lens, pcount, max_cpg, max_bp)
        for c, (st, ln) in enumerate(zip(starts, lens)):
            want = oracle.segment_chunk_mt([s[st:st + ln] for s in slices], loci[st:st + ln], pcount, max_cpg, max_bp)
            if got[c].tolist() != want.tolist():
                bad.append('deep seed %d samples %d pcount %r max_cpg %d max_bp %d chunk [%d,+%d): %s' % (
                    seed, len(slices), pcount, max_cpg, max_bp, st, ln, fuzzlib.first_diff(got[c], want)))
        done += 1
    print('deep fuzz: seeds %d .. %d, %d differences' % (first, first + done - 1, len(bad)))
    assert not bad, '%d deep-mode chunks differ from the oracle; first: %s' % (len(bad), bad[0])
    assert done >= 3, 'the time box of %.0f s covered only %d deep cases' % (seconds, done)


@pytest.mark.parametrize('seed', range(16 + 100 * fuzzlib.round_number(), 80 + 100 * fuzzlib.round_number()))
def test_13c_more_uniform_seeds(seg, seed):
    T.test_13_random_parameters_and_adversarial_inputs_match_oracle(seg, seed)


@pytest.mark.skipif(not oracle.have_ref(), reason='oracle/_ref/segmentor not built (needs /root/reference at build time)')
@pytest.mark.parametrize('seed', range(300000 + 100 * fuzzlib.round_number(), 300006 + 100 * fuzzlib.round_number()))
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
