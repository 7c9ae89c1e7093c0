"""Index-logic emulation of the HIP kernels in numpy/python (developer tool; no GPU needed).
Mirrors the lane/tile arithmetic of seg_kernels.h (not the DPP mechanics) to validate the algorithm against the oracle."""
import sys, os.path as op
import numpy as np
ROOT = op.dirname(op.dirname(op.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, op.join(ROOT, 'tests'))
import cases
from oracle import oracle

def windows(loci, max_cpg, max_bp):
    l = loci.astype(np.int64); i = np.arange(l.size)
    lo = np.maximum(np.searchsorted(l, l - max_bp, 'left'), i + 1 - max_cpg)
    return (i - lo + 1).astype(np.int64)

def group_start(start0, k):
    a = (start0 + k) & ~63
    return 0 if a <= start0 else a - start0

def emu_scan_carries(row, start0, ln):
    """k_scan for one (chunk, sample) row: carry[g] at absolute 64-multiples (g>=1) / chunk start (g=0)."""
    g0 = start0 >> 6
    nG = ((start0 + ln - 1) >> 6) - g0 + 1
    carry = np.full((nG, 2), -1, dtype=np.int64)
    carry[0] = 0
    n_total = row.shape[0]
    a_abs = start0 & ~7; head = start0 - a_abs; span = head + ln
    run = np.zeros(2, dtype=np.int64)
    for base in range(0, span, 512):
        tot = np.zeros((64, 2), dtype=np.int64)
        for lane in range(64):
            off = base + lane * 8; rel0 = off - head
            for j in range(8):
                a = a_abs + off + j
                m, c = (row[a] if (off < span and a < n_total) else (0, 0))
                rel = rel0 + j
                if not (0 <= rel < ln): m = c = 0
                tot[lane] += (m, c)
        incl = np.cumsum(tot, axis=0)
        for lane in range(64):
            off = base + lane * 8; rel0 = off - head; vabs = a_abs + off
            if (vabs & 63) == 0 and 0 < rel0 < ln:
                carry[(vabs >> 6) - g0] = run + incl[lane] - tot[lane]
        run += incl[63]
    return carry

def emu_stage_row(row, carry, start0, ln, A, cnt):
    """wg_stage_prefix_row: returns dst[0..cnt) (uint2)."""
    n_total = row.shape[0]
    dst = np.full((cnt, 2), -1, dtype=np.int64)
    abs0 = start0 + A
    run = carry[(abs0 >> 6) - (start0 >> 6)].copy()
    al = abs0 & ~3; hs = abs0 - al
    for p0 in range(0, cnt + hs, 256):
        tot = np.zeros((64, 2), dtype=np.int64); vals = np.zeros((64, 4, 2), dtype=np.int64)
        for lane in range(64):
            sidx = p0 + lane * 4
            for j in range(4):
                a = al + sidx + j
                m, c = (row[a] if (sidx < cnt + hs and a < n_total) else (0, 0))
                x = sidx + j - hs
                if not (x >= 0 and A + x < ln): m = c = 0
                vals[lane, j] = (m, c); tot[lane] += (m, c)
        incl = np.cumsum(tot, axis=0)
        for lane in range(64):
            e = run + incl[lane] - tot[lane]
            for j in range(4):
                x = p0 + lane * 4 + j - hs
                if 0 <= x < cnt: dst[x] = e
                e = e + vals[lane, j]
        run = run + incl[63]
    return dst

def emu_cost_tiles(W, n, S, stage, TI, KT, TK, Wmax, start0=0):
    """k_cost tile decomposition for one chunk: yields (pairs list of (i,k)) per tile; checks K/I array bounds."""
    s0 = stage * S; s1 = min(s0 + S, n)
    if s0 >= n: return
    KS = TK + 64 if KT > 1 else TI + Wmax + 64
    IS = TI + 64 if KT > 1 else 0
    ntile = (s1 - s0 + TI - 1) // TI * KT
    for local in range(ntile):
        it, kt = divmod(local, KT)
        ia = s0 + it * TI; ib = min(ia + TI, s1); ni = ib - ia
        kt_hi = ia + TI - (KT - 1 - kt) * TK if KT > 1 else ia + TI
        kt_lo = kt_hi - TK if KT > 1 else -(1 << 30)
        pairs = []; kmin = None
        for il in range(ni):
            i = ia + il; lo = i - W[i] + 1
            ks = max(lo, kt_lo); ke = min(i, kt_hi - 1)
            if ke >= ks:
                kmin = ks if kmin is None else min(kmin, ks)
                pairs += [(i, k) for k in range(ks, ke + 1)]
        if not pairs: continue
        kA = group_start(start0, kmin)
        assert kmin - kA <= 63
        assert kA >= 0
        if KT > 1:
            iA = group_start(start0, ia); Kcnt = kt_hi - kA; Icnt = ib + 1 - iA
            assert Kcnt <= KS and Icnt <= IS, (Kcnt, KS, Icnt, IS)
            for (i, k) in pairs: assert 0 <= k - kA < Kcnt and 0 <= i + 1 - iA < Icnt
        else:
            Kcnt = ib + 1 - kA
            assert Kcnt <= KS, (Kcnt, KS)
            for (i, k) in pairs: assert 0 <= k - kA < Kcnt and 0 <= i + 1 - kA < Kcnt
        assert len(pairs) <= 4096, len(pairs)
        yield pairs

def emu_dp(W, cum, cost, n, S, max_cpg):
    """k_dp over all stages; cost CSR (global cum). Returns back array."""
    ringN = 1
    while ringN < max(64, max_cpg): ringN <<= 1
    rmask = ringN - 1
    ring = np.zeros(ringN); back = np.zeros(n, dtype=np.int64)
    nst = (n + S - 1) // S
    for stage in range(nst):
        s0 = stage * S; s1 = min(s0 + S, n)
        mreg = np.zeros(64)
        for lane in range(64):
            k = s0 - ((s0 - lane) & 63)
            mreg[lane] = ring[k & rmask] if k >= 0 else 0.0
        for base in range(s0, s1, 64):
            for stp in range(min(64, s1 - base)):
                i = base + stp; w = int(W[i]); lo = i - w + 1
                if w <= 64:
                    v = np.full(64, -np.inf)
                    for lane in range(64):
                        j = (lane - lo) & 63
                        if j < w: v[lane] = mreg[lane] + cost[cum[i] + j]
                    vmax = v.max(); eq = 0
                    for lane in range(64):
                        if v[lane] == vmax: eq |= 1 << lane
                    rot = lo & 63
                    rm = ((eq >> rot) | (eq << (64 - rot))) & ((1 << 64) - 1) if rot else eq
                    kbest = lo + ((rm & -rm).bit_length() - 1)
                else:
                    best = np.full(64, -np.inf); bk = np.full(64, 2**32 - 1, dtype=np.int64)
                    for lane in range(64):
                        for jj in range(lane, w, 64):
                            k = lo + jj; vv = ring[k & rmask] + cost[cum[i] + jj]
                            if vv > best[lane]: best[lane] = vv; bk[lane] = k
                    vmax = best.max(); kbest = int(bk[best == vmax].min())
                mreg[(i + 1) & 63] = vmax; ring[(i + 1) & rmask] = vmax
                back[i] = i + 1 - kbest
    return back

def run_case(name, S=None, TIsel=None):
    spec = cases.CHUNK_CASES[name]
    slices, loci = cases.build_case(spec)
    n, max_cpg, max_bp = spec['n'], spec['max_cpg'], spec['max_bp']
    b, M, T, band = oracle.segment_chunk(slices, loci, spec['pcount'], max_cpg, max_bp, debug=True)
    W = windows(loci, max_cpg, max_bp); cum = np.concatenate([[0], np.cumsum(W)[:-1]]); Wmax = int(W.max())
    if Wmax <= 64: TI, KT, TK = 64, 1, 0
    elif Wmax <= 128: TI, KT, TK = 32, 1, 0
    elif Wmax <= 256: TI, KT, TK = 16, 1, 0
    else: TI, TK = 16, 256; KT = (Wmax - 1 + TI + TK - 1) // TK
    if S is None: S = ((n + 63) // 64) * 64
    # tile coverage: every (i,k) exactly once
    seen = np.zeros(int(W.sum()), dtype=np.int32)
    for stage in range((n + S - 1) // S):
        for pairs in emu_cost_tiles(W, n, S, stage, TI, KT, TK, Wmax):
            for (i, k) in pairs: seen[cum[i] + k - (i - W[i] + 1)] += 1
    assert (seen == 1).all(), 'tile coverage broken'
    # dp on oracle costs
    cost = np.empty(int(W.sum()))
    for i in range(n):
        ks = np.arange(i - W[i] + 1, i + 1); cost[cum[i]:cum[i] + W[i]] = band[ks, i - ks]
    back = emu_dp(W, cum, cost, n, S, max_cpg)
    assert (back == np.arange(1, n + 1) - T[1:]).all(), 'dp emulation differs'
    print(name, 'ok: Wmax', Wmax, 'TI', TI, 'KT', KT, 'S', S)

if __name__ == '__main__':
    # scan carries + staged prefix rows
    rng = np.random.default_rng(0)
    row = rng.integers(0, 200, (5000, 2)); row[:, 0] = np.minimum(row[:, 0], row[:, 1])
    for start0, ln in [(0, 1), (3, 700), (8, 512), (13, 4000), (4990, 10), (1, 64), (7, 129), (64, 200), (63, 2), (100, 1000)]:
        carry = emu_scan_carries(row, start0, ln)
        P = np.concatenate([[[0, 0]], np.cumsum(row[start0:start0 + ln], axis=0)])
        g0 = start0 >> 6
        for g in range(carry.shape[0]):
            pos = 0 if g == 0 else ((g0 + g) << 6) - start0
            if g > 0 and pos >= ln: continue
            assert (carry[g] == P[pos]).all(), (start0, ln, g, carry[g], P[pos])
        for k in list(range(0, ln, 37)) + [ln - 1]:
            A = group_start(start0, k)
            assert 0 <= k - A <= 63
            for cnt in (1, 5, 64, 65, 200, 300):
                dst = emu_stage_row(row, carry, start0, ln, A, cnt)
                for x in range(cnt):
                    want = P[min(A + x, ln)]
                    assert (dst[x] == want).all(), (start0, ln, A, cnt, x, dst[x], want)
    print('scan/stage emulation ok')
    for nm in ['tiny', 'n1', 'n2', 'n65', 'max_cpg2', 'max_cpg_binds', 'dense_w_gt_64', 'dense_small_bp', 'equal_loci']:
        run_case(nm)
    run_case('tiny', S=64); run_case('dense_w_gt_64', S=128); run_case('max_cpg_binds', S=192)
