# tools/aligned_fuzz.py [first_seed] [n_seeds] [seconds]: like tools/extra_fuzz.py, but the chunk starts, ends and lengths are drawn ON and
# next to the multiples of 16 / 64 / 128 / 256 where the kernels' tiles, units, carry groups and batches begin (the case a uniform draw
# meets once in thousands of chunks: seed 5751 of the uniform fuzz), window limits next to the tile-class boundary (60 / 64 / 128 sites).
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import test_gpu_parity as T
from wgbs_tools_amd import _lib
import oracle.oracle as oracle

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
seg = _lib.Segmenter(0)
t0 = time.time()
bad = done = chunks = 0


def near(rng, lo, hi):
    """A point of [lo, hi] on or next to a multiple of 16 / 64 / 128 / 256."""
    m = int(rng.choice([16, 64, 64, 128, 256]))
    for _ in range(20):
        x = int(rng.integers(lo // m, hi // m + 1)) * m + int(rng.choice([-1, 0, 0, 0, 1]))
        if lo <= x <= hi:
            return x
    return int(rng.integers(lo, hi + 1))


for seed in range(first, first + count):
    if time.time() - t0 > budget:
        break
    done += 1
    rng = np.random.default_rng(50000 + seed)
    n = int(rng.integers(3000, 9000))
    n_samples = int(rng.choice([1, 1, 2, 3, 7, 33, 40]))
    slices, loci = T._fuzz_world(rng, n, n_samples)
    if rng.random() < 0.5:                                          # dense: the windows are what max_cpg says
        loci = (np.cumsum(rng.integers(0, 5, n)) + 1000).astype(np.uint32)
    seg.set_betas(slices); seg.set_loci(loci)
    for draw in range(4):
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
        got = seg.segment_chunks(starts, lens, pcount, max_cpg, max_bp)
        want = oracle.segment_chunks(slices, loci, starts, lens, pcount, max_cpg, max_bp, threads=os.cpu_count() or 1)
        chunks += 12
        for c, (a, b) in enumerate(zip(got, want)):
            if a.tolist() != b.tolist():
                bad += 1
                print('seed %d draw %d samples %d pcount %r max_cpg %d max_bp %d chunk [%d,+%d): %s' % (
                    seed, draw, n_samples, pcount, max_cpg, max_bp, starts[c], lens[c], T._first_diff(a, b)), flush=True)
print('done: seeds %d .. %d, %d chunks, differences: %d' % (first, first + done - 1, chunks, bad), flush=True)
