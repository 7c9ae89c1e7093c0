"""Index-logic emulation of the HIP kernels in numpy/python (developer tool; no GPU needed).
Mirrors the lane/tile arithmetic of seg_kernels.h (not the DPP mechanics) to validate the algorithm against the oracle."""
import sys, os.path as op
import numpy as np
ROOT = op.dirname(op.dirname(op.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, op.join(ROOT, 'tests'))
import cases
from oracle import oracle

def windows(loci, max_cpg, max_bp):
    """forward window F_k"""
    l = loci.astype(np.int64); k = np.arange(l.size)
    hi = np.searchsorted(l, l + max_bp, 'right') - 1
    hi = np.minimum(np.minimum(hi, k + max_cpg - 1), l.size - 1)
    return (hi - k + 1).astype(np.int64)

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

def emu_stage_row(row, carry, start0, ln, A, cnt, x0=0):
    cnt0 = cnt; cnt = x0 + cnt0
    """wg_stage_prefix_row: returns dst[0..cnt) (uint2)."""
    n_total = row.shape[0]
    dst = np.full((cnt0, 2), -1, dtype=np.int64)
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
                if x0 <= x < cnt: dst[x - x0] = e
                e = e + vals[lane, j]
        run = run + incl[63]
    return dst

def emu_cost_tiles(F, n, S, stage, TI, KT, TK, Fmax, start0=0):
    """k_cost tile decomposition (start-major) for one chunk: yields pairs (k,i) per tile; checks E/S array bounds."""
    s0 = stage * S; s1 = min(s0 + S, n)
    if s0 >= n: return
    KS = TK + 1 if KT > 1 else TI + Fmax + 1
    IS = TI + 1 if KT > 1 else 0
    ntile = (s1 - s0 + TI - 1) // TI * KT
    for local in range(ntile):
        kti, et = divmod(local, KT)
        ka = s0 + kti * TI; kb = min(ka + TI, s1); nk = kb - ka
        et_lo = ka + et * TK if KT > 1 else 0
        et_hi = et_lo + TK if KT > 1 else (1 << 30)
        pairs = []; imin = None; imax = None
        for kl in range(nk):
            k = ka + kl
            is_ = max(k, et_lo); ie = min(k + F[k], et_hi) - 1
            if ie >= is_:
                imin = is_ if imin is None else min(imin, is_); imax = ie if imax is None else max(imax, ie)
                pairs += [(k, i) for i in range(is_, ie + 1)]
        if not pairs: continue
        eA = imin + 1 if KT > 1 else ka
        eG = group_start(start0, eA); assert 0 <= eA - eG <= 63
        Ecnt = imax + 2 - eA
        assert 0 < Ecnt <= KS, (Ecnt, KS)
        if KT > 1:
            sA = ka; Scnt = kb - sA
            assert Scnt <= IS
            for (k, i) in pairs: assert 0 <= i + 1 - eA < Ecnt and 0 <= k - sA < Scnt
        else:
            for (k, i) in pairs: assert 0 <= i + 1 - eA < Ecnt and 0 <= k - eA < Ecnt
        assert len(pairs) <= 4096, len(pairs)
        yield pairs

def emu_dp(F, cum, cost, n, S):
    """k_dp push form over all stages; cost start-major CSR. Returns back array."""
    Fmax = int(F.max()); wide = Fmax > 64
    ringN = 1
    while ringN < Fmax + 64: ringN <<= 1
    rmask = ringN - 1
    pendB = np.full(ringN, -np.inf); pendA = np.zeros(ringN, dtype=np.int64)
    best = np.full(64, -np.inf); arg = np.zeros(64, dtype=np.int64); back = np.zeros(n, dtype=np.int64)
    Mk = 0.0
    for k in range(n):
        f = int(F[k]); stp = k & 63
        for lane in range(64):
            j = (lane - stp) & 63
            if j < f:
                cand = Mk + cost[cum[k] + j]
                if cand > best[lane]: best[lane] = cand; arg[lane] = k
        if f > 64:
            for jj in range(64, f):
                sl = (k + jj) & rmask; cd = Mk + cost[cum[k] + jj]
                if cd > pendB[sl]: pendB[sl] = cd; pendA[sl] = k
        Mk = best[stp]
        back[k] = k + 1 - arg[stp]
        best[stp] = -np.inf
        if wide:
            sl = (k + 64) & rmask
            best[stp] = pendB[sl]; arg[stp] = pendA[sl]; pendB[sl] = -np.inf
    return back

def run_case(name, S=None, TIsel=None):
    spec = cases.CHUNK_CASES[name]
    slices, loci = cases.build_case(spec)
    n, max_cpg, max_bp = spec['n'], spec['max_cpg'], spec['max_bp']
    b, M, T, band = oracle.segment_chunk(slices, loci, spec['pcount'], max_cpg, max_bp, debug=True)
    F = windows(loci, max_cpg, max_bp); cum = np.concatenate([[0], np.cumsum(F)[:-1]]); Fmax = int(F.max())
    # the oracle band agrees with the forward windows: finite exactly on j < F_k
    for k in range(0, n, max(1, n // 200)):
        row = band[k, :min(max_cpg, n - k)]
        assert np.isfinite(row[:F[k]]).all() and not np.isfinite(row[F[k]:]).any()
    if Fmax <= 64: TI, KT, TK = 64, 1, 0
    elif Fmax <= 128: TI, KT, TK = 32, 1, 0
    elif Fmax <= 256: TI, KT, TK = 16, 1, 0
    else: TI, TK = 16, 256; KT = (Fmax - 1 + TI + TK - 1) // TK
    if S is None: S = ((n + 63) // 64) * 64
    seen = np.zeros(int(F.sum()), dtype=np.int32)
    for stage in range((n + S - 1) // S):
        for pairs in emu_cost_tiles(F, n, S, stage, TI, KT, TK, Fmax, start0=spec['a']):
            for (k, i) in pairs: seen[cum[k] + i - k] += 1
    assert (seen == 1).all(), 'tile coverage broken'
    cost = np.empty(int(F.sum()))
    for k in range(n): cost[cum[k]:cum[k] + F[k]] = band[k, :F[k]]
    back = emu_dp(F, cum, cost, n, S)
    assert (back == np.arange(1, n + 1) - T[1:]).all(), 'dp emulation differs'
    print(name, 'ok: Fmax', Fmax, 'TI', TI, 'KT', KT, 'S', S)

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
                x0 = k - A
                dst = emu_stage_row(row, carry, start0, ln, A, cnt, x0)
                for x in range(cnt):
                    want = P[min(k + x, ln)]
                    assert (dst[x] == want).all(), (start0, ln, A, cnt, x, dst[x], want)
    print('scan/stage emulation ok')
    for nm in ['tiny', 'n1', 'n2', 'n65', 'max_cpg2', 'max_cpg_binds', 'dense_w_gt_64', 'dense_small_bp', 'equal_loci', 'island_mix']:
        run_case(nm)
    run_case('tiny', S=64); run_case('dense_w_gt_64', S=128); run_case('max_cpg_binds', S=192)
