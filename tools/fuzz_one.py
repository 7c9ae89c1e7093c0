# tools/fuzz_one.py SEED: the draws of tests/test_gpu_parity.py::test_13 for one seed, chunk by chunk, with the first differing border
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import test_gpu_parity as T
from wgbs_tools_amd import _lib
import oracle.oracle as oracle
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
n = int(rng.integers(3000, 9000))
n_samples = int(rng.choice([1, 2, 3, 7, 33, 40]))
slices, loci = T._fuzz_world(rng, n, n_samples)
seg = _lib.Segmenter(0)
seg.set_betas(slices); seg.set_loci(loci)
bad = 0
for draw in range(4):
    pcount = float(rng.choice([0.0, 0.25, 0.99999994, 1.0, 3.9999998, 15.0, 100.0, 1e-3, 1e-8, 1e30]))
    max_cpg = int(rng.choice([1, 2, 17, 64, 65, 129, 300, 1000]))
    max_bp = int(rng.choice([1, 2, 50, 700, 2000, 100000]))
    starts, lens = [], []
    for _ in range(12):
        ln = int(rng.integers(1, min(n, 2500))); st = int(rng.integers(0, n - ln + 1))
        starts.append(st); lens.append(ln)
    got = seg.segment_chunks(starts, lens, pcount, max_cpg, max_bp)
    want = oracle.segment_chunks(slices, loci, starts, lens, pcount, max_cpg, max_bp, threads=os.cpu_count() or 1)
    for c, (a, b) in enumerate(zip(got, want)):
        if a.tolist() != b.tolist():
            bad += 1
            k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
            print('draw %d pcount %r max_cpg %d max_bp %d n_samples %d chunk %d [%d,+%d): %d vs %d borders, first difference at #%d: got %s want %s' % (
                draw, pcount, max_cpg, max_bp, n_samples, c, starts[c], lens[c], len(a), len(b), k, a[max(0,k-2):k+3].tolist(), b[max(0,k-2):k+3].tolist()))
            # alone
            alone = seg.segment_chunks([starts[c]], [lens[c]], pcount, max_cpg, max_bp)[0]
            print('   the same chunk alone: %s' % ('identical to the oracle' if alone.tolist() == b.tolist() else 'differs too'))
    print('draw %d: timings %s' % (draw, {k: v for k, v in seg.timings().items() if k in ('max_window', 'n_stages', 'div_short')}))
print('seed', seed, 'differences:', bad)
