# tools/fuzz_diag2.py: where does the stray cost entry of seed 5751 (draw 3, chunk 0) come from?  Chunks around it, in the world and as a world of their own,
# with and without an earlier call that leaves other numbers in the cost buffer.
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import test_gpu_parity as T
from wgbs_tools_amd import _lib
import oracle.oracle as oracle
rng = np.random.default_rng(1000 + 5751)
n = int(rng.integers(3000, 9000))
n_samples = int(rng.choice([1, 2, 3, 7, 33, 40]))
slices, loci = T._fuzz_world(rng, n, n_samples)
pcount, max_cpg, max_bp = 3.9999998, 129, 100000
os.environ['WGBSSEG_FORCE_STAGES'] = '1'
print('world of %d sites, %d samples' % (n, n_samples))

def check(sg, sl, lo, st, ln, tag, pc=pcount, mc=max_cpg, mb=max_bp):
    got = sg.segment_chunks([st], [ln], pc, mc, mb)[0]
    s1 = [np.ascontiguousarray(s[st:st + ln]) for s in sl]
    l1 = np.ascontiguousarray(lo[st:st + ln])
    b, M, Tt, band = oracle.segment_chunk(s1, l1, pc, mc, mb, debug=True)
    W = T._numpy_windows(l1, mc, mb)
    cum = np.concatenate([[0], np.cumsum(W)[:-1]])
    gcost = sg.debug_fetch('cost', np.float64, int(W.sum()))
    want = np.empty(int(W.sum()), dtype=np.float64)
    for k in range(ln):
        want[cum[k]:cum[k] + W[k]] = band[k, :W[k]]
    bad = np.flatnonzero(gcost.view(np.uint64) != want.view(np.uint64))
    msg = ''
    for j in bad[:3]:
        k = int(np.searchsorted(cum, j, 'right') - 1)
        msg += ' [start %d len %d: got %r want %r]' % (k, j - cum[k] + 1, float(gcost[j]), float(want[j]))
    print('%-58s borders %s, cost entries off: %d%s' % (tag, 'ok ' if got.tolist() == b.tolist() else 'BAD', bad.size, msg), flush=True)

sg = _lib.Segmenter(0); sg.set_betas(slices); sg.set_loci(loci)
check(sg, slices, loci, 3263, 1345, 'world, [3263,+1345), first call')
check(sg, slices, loci, 3263, 1345, 'world, again')
check(sg, slices, loci, 0, 2500, 'world, [0,+2500)')
check(sg, slices, loci, 3263, 1345, 'world, [3263,+1345) after it')
for st, ln in [(3262, 1346), (3264, 1344), (3263, 1344), (3263, 1346), (3263, 1281), (3263, 1409), (3263, 65), (3263, 129), (3199, 1409), (4543, 65), (4607, 1), (4606, 2),
               (3259, 1345), (3260, 1345), (3261, 1345), (3262, 1345), (3264, 1345), (3265, 1345), (3266, 1345)]:
    check(sg, slices, loci, st, ln, 'world, [%d,+%d)' % (st, ln))
for mc in (64, 65, 128, 300):
    check(sg, slices, loci, 3263, 1345, 'world, [3263,+1345) max_cpg %d' % mc, mc=mc)
for pc in (0.0, 0.25, 1.0, 15.0):
    check(sg, slices, loci, 3263, 1345, 'world, [3263,+1345) pcount %r' % pc, pc=pc)
sg.close()
sl = [np.ascontiguousarray(s[3263:3263 + 1345]) for s in slices]; lo = np.ascontiguousarray(loci[3263:3263 + 1345])
sg = _lib.Segmenter(0); sg.set_betas(sl); sg.set_loci(lo)
check(sg, sl, lo, 0, 1345, 'own world, [0,+1345), first call')
check(sg, sl, lo, 0, 1300, 'own world, [0,+1300)')
check(sg, sl, lo, 0, 1345, 'own world, [0,+1345) after it')
sg.close()
# the same sites three sites further into a world (alignment of the first byte)
for sh in (1, 2, 3, 4):
    sl2 = [np.concatenate([s[:sh], s[3263:3263 + 1345], s[:7]]) for s in slices]; lo2 = np.concatenate([loci[:sh], loci[3263:3263 + 1345], loci[-7:]])
    sg = _lib.Segmenter(0); sg.set_betas(sl2); sg.set_loci(lo2)
    check(sg, sl2, lo2, sh, 1345, 'own world shifted by %d, [%d,+1345)' % (sh, sh))
    sg.close()
