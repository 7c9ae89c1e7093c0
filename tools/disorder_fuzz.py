# tools/disorder_fuzz.py [first_seed] [n_seeds] [seconds]: random worlds whose loci are put out of order in random places (restarts, descending runs,
# swaps, shuffled stretches, equal runs), random parameters, batches that mix chunks with and without such places — the plain path of
# csrc/plain_dp.h (and the split of a batch around it) against the oracle's restatement of segmentor.cpp:103-155.
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import fuzzlib
from wgbs_tools_amd import _lib
import oracle.oracle as oracle

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
seg = _lib.Segmenter(0)
t0, done, chunks, bad = time.time(), 0, 0, 0
for seed in range(first, first + count):
    if time.time() - t0 > budget:
        break
    rng = np.random.default_rng(900000 + seed)
    n = int(rng.integers(400, 4000))
    ns = int(rng.choice([1, 2, 3, 9, 33]))
    slices, loci = fuzzlib.fuzz_world(rng, n, ns)
    loci = np.minimum(loci.astype(np.int64), 2**31 - 1)
    for _ in range(int(rng.integers(1, 6))):                     # put a few places out of order
        kind = rng.integers(0, 5)
        at = int(rng.integers(1, n - 1))
        ln = int(rng.integers(2, max(3, min(400, n - at))))
        if kind == 0:
            loci[at:] -= loci[at] - int(rng.integers(1, 50))      # the positions start again
            loci = np.maximum(loci, 0)
        elif kind == 1:
            loci[at:at + ln] = loci[at:at + ln][::-1]
        elif kind == 2:
            loci[at], loci[at - 1] = loci[at - 1], loci[at]
        elif kind == 3:
            loci[at:at + ln] = loci[at:at + ln][rng.permutation(min(ln, n - at))]
        else:
            loci[at:at + ln] = loci[at]
    loci = loci.astype(np.uint32)
    seg.set_betas(slices)
    seg.set_loci(loci)
    pcount = float(rng.choice([0.0, 0.5, 1.0, 15.0, float(np.float32(np.exp2(rng.uniform(-8, 8))))]))
    max_cpg = int(rng.choice([2, 17, 60, 61, 130, 1000]))
    max_bp = int(rng.choice([50, 700, 2000, 100000]))
    starts, lens = [], []
    for _ in range(8):
        ln = int(rng.integers(1, n + 1)); st = int(rng.integers(0, n - ln + 1))
        starts.append(st); lens.append(ln)
    got = seg.segment_chunks(starts, lens, pcount, max_cpg, max_bp)
    want = oracle.segment_chunks(slices, loci, starts, lens, pcount, max_cpg, max_bp, threads=os.cpu_count() or 1)
    for c, (a, b) in enumerate(zip(got, want)):
        chunks += 1
        if a.tolist() != b.tolist():
            bad += 1
            print('seed %d samples %d pcount %r max_cpg %d max_bp %d chunk [%d,+%d): %s' % (seed, ns, pcount, max_cpg, max_bp, starts[c], lens[c], fuzzlib.first_diff(a, b)), flush=True)
    done += 1
print('done: seeds %d .. %d, %d chunks, differences: %d' % (first, first + done - 1, chunks, bad), flush=True)
