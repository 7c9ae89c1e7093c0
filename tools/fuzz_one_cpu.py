# tools/fuzz_one_cpu.py SEED DRAW CHUNK: oracle vs the reference binary on one chunk of tests/test_gpu_parity.py::test_13's draws (no GPU)
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import test_gpu_parity as T
import oracle.oracle as oracle
seed, want_draw, want_chunk = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(1000 + seed)
n = int(rng.integers(3000, 9000))
n_samples = int(rng.choice([1, 2, 3, 7, 33, 40]))
slices, loci = T._fuzz_world(rng, n, n_samples)
for draw in range(4):
    pcount = float(rng.choice([0.0, 0.25, 0.99999994, 1.0, 3.9999998, 15.0, 100.0, 1e-3, 1e-8, 1e30]))
    max_cpg = int(rng.choice([1, 2, 17, 64, 65, 129, 300, 1000]))
    max_bp = int(rng.choice([1, 2, 50, 700, 2000, 100000]))
    starts, lens = [], []
    for _ in range(12):
        ln = int(rng.integers(1, min(n, 2500))); st = int(rng.integers(0, n - ln + 1))
        starts.append(st); lens.append(ln)
    if draw != want_draw:
        continue
    st, ln = starts[want_chunk], lens[want_chunk]
    sl = [np.ascontiguousarray(s[2 * st:2 * (st + ln)]) if s.ndim == 1 else s[st:st + ln] for s in slices]
    a = oracle.segment_chunk(sl, loci[st:st + ln], pcount, max_cpg, max_bp)
    b = oracle.ref_segment_arrays(sl, loci[st:st + ln], pcount, max_cpg, max_bp)
    print('pcount %r max_cpg %d max_bp %d [%d,+%d) samples %d' % (pcount, max_cpg, max_bp, st, ln, n_samples))
    print('oracle   ', len(a), a[-5:].tolist())
    print('reference', len(b), np.asarray(b)[-5:].tolist())
    print('last sites meth/total:', [tuple(x) for x in np.asarray(sl[0]).reshape(-1, 2)[-6:].tolist()], 'loci', loci[st + ln - 6:st + ln].tolist())
