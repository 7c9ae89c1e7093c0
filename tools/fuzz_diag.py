# tools/fuzz_diag.py SEED DRAW CHUNK [world]: one chunk of test_13's draws alone, intermediates (windows, cost matrix, back pointers) against the oracle's
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import test_gpu_parity as T
from wgbs_tools_amd import _lib
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
    if draw == want_draw:
        break
st, ln = starts[want_chunk], lens[want_chunk]
sl = [np.ascontiguousarray(s[st:st + ln]) for s in slices]
lo = np.ascontiguousarray(loci[st:st + ln])
os.environ['WGBSSEG_FORCE_STAGES'] = '1'
sg = _lib.Segmenter(0)
whole = len(sys.argv) > 4 and sys.argv[4] == 'world'
if whole:
    sg.set_betas(slices); sg.set_loci(loci)
    got = sg.segment_chunks([st], [ln], pcount, max_cpg, max_bp)[0]
else:
    sg.set_betas(sl); sg.set_loci(lo)
    got = sg.segment_chunks([0], [ln], pcount, max_cpg, max_bp)[0]
b, M, Tt, band = oracle.segment_chunk(sl, lo, pcount, max_cpg, max_bp, debug=True)
print('borders: got %d want %d, tails %s / %s' % (len(got), len(b), got[-4:].tolist(), b[-4:].tolist()))
W = T._numpy_windows(lo, max_cpg, max_bp)
gW = sg.debug_fetch('window', np.uint16, ln).astype(np.int64)
print('windows differ at', np.flatnonzero(gW != W)[:5].tolist(), 'max', int(W.max()))
cum = np.concatenate([[0], np.cumsum(W)[:-1]])
gcost = sg.debug_fetch('cost', np.float64, int(W.sum()))
want = np.empty(int(W.sum()), dtype=np.float64)
for k in range(ln):
    want[cum[k]:cum[k] + W[k]] = band[k, :W[k]]
bad = np.flatnonzero(gcost.view(np.uint64) != want.view(np.uint64))
print('cost entries that differ: %d of %d' % (bad.size, want.size))
for j in bad[:12]:
    k = int(np.searchsorted(cum, j, 'right') - 1)
    print('   start %d len %d (window %d): got %r want %r' % (k, j - cum[k] + 1, W[k], gcost[j], want[j]))
gback = sg.debug_fetch('back', np.uint16, ln).astype(np.int64)
wback = np.arange(1, ln + 1) - Tt[1:]
bd = np.flatnonzero(gback != wback)
print('back pointers that differ: %d; first %s' % (bd.size, [(int(i), int(gback[i]), int(wback[i])) for i in bd[:8]]))
for k in range(max(0, ln - 4), ln):
    print('   row %d (window %d): got %s want %s' % (k, W[k], gcost[cum[k]:cum[k] + min(W[k], 4)].tolist(), band[k, :min(W[k], 4)].tolist()))
print('M tail (oracle):', M[-5:].tolist())
