"""Random worlds and chunk lists for the parity fuzzers (tests/test_gpu_fuzz.py and tools/aligned_fuzz.py share them).

The ALIGNED draw puts chunk starts, ends and lengths on and next to the multiples of 16 / 64 / 128 / 256 where the kernels' units,
tiles, carry groups and batches begin, with window limits next to the tile-class boundary (60 / 64 / 128 sites): the cases a uniform
draw meets once in thousands of chunks (round 2: seed 5751 of the uniform fuzz, the carry defect of a wide tile that wants P[len]
alone)."""
import glob
import os.path as op
import re

import numpy as np


def round_number():
    """The build round, read off the newest profiles/rNN_* file (tracked, so the GPU box sees the same number): the fuzzers derive
    their first seed from it, so every round's suite covers ground no earlier round has."""
    here = op.dirname(op.dirname(op.abspath(__file__)))
    r = [int(m.group(1)) for f in glob.glob(op.join(here, 'profiles', 'r[0-9][0-9]_*')) for m in [re.match(r'r(\d\d)_', op.basename(f))] if m]
    return max(r) if r else 0


def fuzz_world(rng, n, n_samples):
    """Random loci (dense runs, equal positions, long gaps) and counts (zeros, saturated bytes, meth == cov)."""
    kind = rng.integers(0, 4, n)
    gap = np.where(kind == 0, 0, np.where(kind == 1, rng.integers(1, 12, n), np.where(kind == 2, rng.integers(2, 300, n), rng.integers(300, 9000, n))))
    loci = (np.cumsum(gap) + 1000).astype(np.uint32)
    slices = []
    for _ in range(n_samples):
        cov = rng.integers(0, 256, n)
        mode = rng.integers(0, 6, n)
        cov = np.where(mode == 0, 0, np.where(mode == 1, 255, cov))
        meth = np.minimum(cov, np.where(mode == 2, cov, np.where(mode == 3, 0, rng.integers(0, 256, n))))
        slices.append(np.stack([meth, cov], axis=1).astype(np.uint8))
    return slices, loci


def near(rng, lo, hi):
    """A point of [lo, hi] on or next to a multiple of 16 / 64 / 128 / 256."""
    m = int(rng.choice([16, 64, 64, 128, 256]))
    for _ in range(20):
        x = int(rng.integers(lo // m, hi // m + 1)) * m + int(rng.choice([-1, 0, 0, 0, 1]))
        if lo <= x <= hi:
            return x
    return int(rng.integers(lo, hi + 1))


def aligned_world(seed):
    """-> (rng, slices, loci) of aligned-fuzz seed `seed` (half of the worlds dense: the windows are what max_cpg says)."""
    rng = np.random.default_rng(50000 + seed)
    n = int(rng.integers(3000, 9000))
    n_samples = int(rng.choice([1, 1, 2, 3, 7, 33, 40]))
    slices, loci = fuzz_world(rng, n, n_samples)
    if rng.random() < 0.5:
        loci = (np.cumsum(rng.integers(0, 5, n)) + 1000).astype(np.uint32)
    return rng, slices, loci


def aligned_draw(rng, n, reference_safe=False):
    """One parameter set and 12 boundary-aligned chunks of a world of n sites -> (pcount, max_cpg, max_bp, starts, lens).
    reference_safe: only parameters the reference BINARY takes on its command line the way the driver passes them."""
    pcount = float(rng.choice([0.0, 0.25, 1.0, 3.9999998, 15.0, 100.0, 1e-3]))
    if rng.random() < 0.3:                                         # a pseudo count nobody chose (its own short-division check, its own table rows)
        pcount = float(np.float32(np.exp2(rng.uniform(-12, 12))))
    max_cpg = int(rng.choice([2, 17, 59, 60, 61, 64, 65, 127, 128, 129, 130, 193, 300, 1000]))
    max_bp = int(rng.choice([50, 700, 2000, 100000, 100000]))
    starts, lens = [], []
    for _ in range(12):
        kind = rng.integers(0, 3)
        if kind == 0:                                           # end on a boundary, length next to one
            end = near(rng, 1, n); ln = min(end, near(rng, 1, 2600)); st = end - ln
        elif kind == 1:                                         # start and end on boundaries
            st = near(rng, 0, n - 1); end = near(rng, st + 1, min(n, st + 2600)); ln = end - st
        else:                                                   # start on a boundary, length next to one
            st = near(rng, 0, n - 1); ln = min(n - st, near(rng, 1, 2600))
        starts.append(st); lens.append(ln)
    return pcount, max_cpg, max_bp, starts, lens


def first_diff(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return 'shape %s vs %s' % (a.shape, b.shape)
    d = np.flatnonzero(a != b)
    if d.size == 0:
        return None
    i = int(d[0])
    return '%d mismatches, first at %d: got %r want %r' % (d.size, i, a.flat[i], b.flat[i])


def run_aligned(seg, oracle, first, count, budget_s, threads, log=None, draws=4):
    """Aligned fuzz over seeds first .. first+count-1 (or until budget_s seconds have passed) against the oracle's C restatement.
    -> (seeds done, chunks compared, list of difference messages)."""
    import time
    t0 = time.time()
    done = chunks = 0
    bad = []
    for seed in range(first, first + count):
        if time.time() - t0 > budget_s:
            break
        done += 1
        rng, slices, loci = aligned_world(seed)
        n = loci.size
        seg.set_betas(slices)
        seg.set_loci(loci)
        for draw in range(draws):
            pcount, max_cpg, max_bp, starts, lens = aligned_draw(rng, n)
            got = seg.segment_chunks(starts, lens, pcount, max_cpg, max_bp)
            want = oracle.segment_chunks(slices, loci, starts, lens, pcount, max_cpg, max_bp, threads=threads)
            chunks += len(starts)
            for c, (a, b) in enumerate(zip(got, want)):
                if a.tolist() != b.tolist():
                    msg = 'seed %d draw %d samples %d pcount %r max_cpg %d max_bp %d chunk [%d,+%d): %s' % (
                        seed, draw, len(slices), pcount, max_cpg, max_bp, starts[c], lens[c], first_diff(a, b))
                    bad.append(msg)
                    if log:
                        log(msg)
    return done, chunks, bad


def deep_draw(seed):
    """One deep-mode case (windows of thousands of sites: wide scoring tiles, the 32-step recurrence with its ring, stages of the
    scored-block buffer): -> (slices, loci, pcount, max_cpg, max_bp, starts, lens).  8-12 k sites, 1 / 3 / 40 samples."""
    rng = np.random.default_rng(700000 + seed)
    n = int(rng.integers(8000, 12001))
    n_samples = int(rng.choice([1, 3, 40]))
    slices, loci = fuzz_world(rng, n, n_samples)
    shape = rng.integers(0, 3)
    if shape == 0:                                                # dense: every window is what max_cpg says
        loci = (np.cumsum(rng.integers(0, 4, n)) + 1000).astype(np.uint32)
    elif shape == 1:                                              # dense stretches between sparse ones
        gap = np.where((np.arange(n) // int(rng.integers(500, 3000))) % 2 == 0, rng.integers(0, 4, n), rng.integers(50, 4000, n))
        loci = (np.cumsum(gap) + 1000).astype(np.uint32)
    pcount = float(rng.choice([0.0, 0.5, 1.0, 15.0, 15.0, float(np.float32(np.exp2(rng.uniform(-6, 8))))]))
    max_cpg = int(rng.integers(2000, 5001))
    max_bp = int(rng.choice([4000, 20000, 1000000, 100000000]))
    ln = int(rng.integers(max(1, n // 2), n + 1))
    st = int(rng.integers(0, n - ln + 1))
    return slices, loci, pcount, max_cpg, max_bp, [0, st], [n, ln]
