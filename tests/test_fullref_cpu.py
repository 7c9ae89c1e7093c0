"""The whole-genome checker of the GPU suite (tests/fullref.py), itself checked on the CPU: its pool of reference processes
returns, for ranges of a [samples][pitch] byte tensor, what the oracle's restatement gives for the same ranges, and
`whole_genome_vs_reference` counts chunks / patches / trees and notices a planted difference.  The chunk engine here is the
oracle — this test is about the checker, not about the product (which has no CPU path)."""
import numpy as np
import pytest

import fullref
from oracle import oracle
from wgbs_tools_amd import synth

SEED = 20260926


class _OracleSeg:
    """the two calls of _lib.Segmenter that fullref uses, answered by the oracle's restatement"""

    def __init__(self, rows, loci):
        self.rows, self.loci = rows, loci

    def segment_chunks(self, st0, ln, pc, mc, mb):
        return oracle.segment_chunks(self.rows, self.loci, st0, ln, pc, mc, mb, threads=4)


def _world(n_sites, n_samples):
    import torch
    rows = [synth.synth_betas(SEED, s, 0, n_sites) for s in range(n_samples)]
    pitch = ((2 * n_sites + 255) // 256) * 256 + 256
    buf = torch.zeros((n_samples, pitch), dtype=torch.uint8)
    for s, r in enumerate(rows):
        buf[s, :2 * n_sites] = torch.from_numpy(r.reshape(-1))
    return rows, buf, synth.synth_loci(SEED, [n_sites])


@pytest.mark.skipif(not oracle.have_ref(), reason='oracle/_ref/segmentor not built')
def test_pool_of_reference_processes_equals_the_restatement():
    rows, buf, loci = _world(9000, 3)
    ranges = [(0, 2500), (2500, 2500), (5000, 4000), (2450, 100), (4950, 100), (8999, 1), (100, 37)]
    ref = fullref.ref_on_ranges(buf, loci, ranges, 15.0, 1000, 2000, procs=3)
    assert set(ref) == set(ranges)
    want = oracle.segment_chunks(rows, loci, [r[0] for r in ranges], [r[1] for r in ranges], 15.0, 1000, 2000)
    for r, w in zip(ranges, want):
        assert np.array_equal(ref[r], w.astype(np.int64)), r


@pytest.mark.skipif(not oracle.have_ref(), reason='oracle/_ref/segmentor not built')
def test_whole_genome_check_counts_and_catches_a_planted_difference():
    import reftree
    n = 10000
    rows, buf, loci = _world(n, 2)
    regions = [(1, 6001), (6001, n + 1)]
    chunk, pc, mc, mb = 2000, 15.0, 1000, 2000
    seg = _OracleSeg(rows, loci)
    res = []
    for a, e in regions:
        eng = fullref.Recorder(seg, pc, mc, mb)
        res.append(fullref.tree(eng.segment_many(fullref.grid(a, e, chunk), {}), eng))
    out = fullref.whole_genome_vs_reference(seg, buf, loci, regions, chunk, pc, mc, mb, res, procs=4)
    assert out['differences'] == 0
    assert (out['chunks_identical'], out['chunks']) == (5, 5)
    assert (out['chromosomes_identical'], out['chromosomes']) == (2, 2)
    assert out['patches_identical'] == out['patches'] >= 3
    # a stitched list that is off by one border, and an engine that is off on one range
    wrong = [res[0].copy(), res[1]]
    wrong[0] = np.delete(wrong[0], len(wrong[0]) // 2)
    out = fullref.whole_genome_vs_reference(seg, buf, loci, regions, chunk, pc, mc, mb, wrong, procs=4)
    assert out['differences'] == 1 and out['chromosomes_identical'] == 1

    class _Off(_OracleSeg):
        def segment_chunks(self, st0, ln, pc, mc, mb):
            r = super().segment_chunks(st0, ln, pc, mc, mb)
            return [np.delete(x, 1) if (s == 2000 and len(x) > 3) else x for s, x in zip(st0, r)]
    out = fullref.whole_genome_vs_reference(_Off(rows, loci), buf, loci, regions, chunk, pc, mc, mb, res, procs=4)
    assert out['differences'] >= 1 and out['chunks_identical'] == 4
