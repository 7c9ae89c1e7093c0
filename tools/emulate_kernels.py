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

CARRY_SHIFT = 7                       # WG_CARRY_SHIFT: a carry every 128 sites of absolute index
CARRY_G = 1 << CARRY_SHIFT

def group_start(start0, k):
    a = (start0 + k) & ~(CARRY_G - 1)
    return 0 if a <= start0 else a - start0

def emu_scan_carries(row, start0, ln):
    """k_scan for one (chunk, sample) row: carry[g] at absolute multiples of CARRY_G (g>=1) / chunk start (g=0).
    64 lanes x 16 sites per iteration; vectors past the row end are clamped onto the last readable one and blanked."""
    g0 = start0 >> CARRY_SHIFT
    nG = ((start0 + ln - 1) >> CARRY_SHIFT) - g0 + 1
    carry = np.full((nG, 2), -1, dtype=np.int64)
    carry[0] = 0
    n_total = row.shape[0]
    a_abs = start0 & ~15; head = start0 - a_abs; span = head + ln
    A6 = a_abs & (CARRY_G - 1)
    vlast = ((n_total - 1) >> 3) - (a_abs >> 3)
    run = np.zeros(2, dtype=np.int64)
    for base in range(0, span, 1024):
        tot = np.zeros((64, 2), dtype=np.int64)
        for lane in range(64):
            off = base + lane * 16; rel0 = off - head
            for h in range(2):
                vi = min((off >> 3) + h, vlast)           # clamped vector index (from a_abs)
                for j in range(8):
                    a = a_abs + vi * 8 + j
                    m, c = row[a] if a < n_total else (255, 0)      # garbage past the row: must be blanked
                    rel = rel0 + 8 * h + j
                    if not (0 <= rel < ln): m = c = 0
                    else: assert a == start0 + rel, 'a clamped vector reached a site inside the chunk'
                    tot[lane] += (m, c)
        incl = np.cumsum(tot, axis=0)
        for lane in range(64):
            off = base + lane * 16; rel0 = off - head
            if ((A6 + off) & (CARRY_G - 1)) == 0 and 0 < rel0 < ln:
                carry[(A6 + off) >> CARRY_SHIFT] = run + incl[lane] - tot[lane]
        run += incl[63]
    # round 3: the carries are staged in LDS and written after the loop by lanes that decide from the group number alone which entries
    # the loop has stored (k_scan's flush): that rule must select exactly the entries written above
    for g in range(1, nG):
        rel0 = (g << CARRY_SHIFT) - A6 - head
        assert (0 < rel0 < ln) == (carry[g, 0] >= 0), (start0, ln, g, rel0, carry[g])
    return carry

def emu_stage_row(row, carry, start0, ln, A, cnt, x0=0):
    cnt0 = cnt; cnt = x0 + cnt0
    """wg_stage_prefix_row: returns dst[0..cnt) (uint2)."""
    n_total = row.shape[0]
    dst = np.full((cnt0, 2), -1, dtype=np.int64)
    abs0 = start0 + A
    run = carry[(abs0 >> CARRY_SHIFT) - (start0 >> CARRY_SHIFT)].copy()
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

NARROW_WMAX = 60                      # WG_NARROW_WMAX: widest window of a narrow tile

def emu_stage_local_rows(rows, start0, ln, ka, nsite):
    """wg_stage_local_rows: tile-local packed prefixes L[x] = sum of sites ka..ka+x-1 (meth | cov << 16), x = 0..nsite,
    for a list of sample rows; two rows per pass (32-lane halves, 4 sites per lane).  Returns [len(rows)][nsite+1]."""
    n_total = rows[0].shape[0]
    out = np.full((len(rows), nsite + 1), -1, dtype=np.int64)
    abs0 = start0 + ka; al = abs0 & ~3; hs = abs0 - al
    assert nsite + hs <= 128, (nsite, hs)
    for r0 in range(0, len(rows), 2):
        for half in range(2):
            rr = r0 + half
            if rr >= len(rows): continue
            tot = np.zeros(32, dtype=np.int64); mt = np.zeros((32, 4), dtype=np.int64)
            for l5 in range(32):
                for j in range(4):
                    a = al + l5 * 4 + j; x = l5 * 4 + j - hs
                    if l5 * 4 < nsite + hs and a < n_total and 0 <= x < nsite:
                        m, c = rows[rr][a]
                        mt[l5, j] = int(m) | (int(c) << 16)
                tot[l5] = mt[l5].sum()
            incl = np.cumsum(tot)
            for l5 in range(32):
                e = incl[l5] - tot[l5]
                for j in range(4):
                    x = l5 * 4 + j - hs
                    if 0 <= x <= nsite: out[rr, x] = e
                    e += mt[l5, j]
    assert (out >= 0).all() and (out & 0xffff).max() < 32768 and (out >> 16).max() < 32768
    return out

def emu_tile_plan(F, n, S, stage, TI, WA, TK):
    """k_tile_count / k_tile_emit for one chunk: -> (narrow tiles [(ka, nk)], wide tiles [(k0, nk, et_lo)])"""
    s0 = stage * S; s1 = min(s0 + S, n)
    A, B = [], []
    if s0 >= n: return A, B
    umax = [int(F[u * 16:min(u * 16 + 16, n)].max()) for u in range((n + 15) // 16)]
    for ka in range(s0, s1, TI):
        kb = min(ka + TI, s1)
        units = [(k0, min(16, kb - k0), umax[k0 >> 4]) for k0 in range(ka, kb, 16)]
        if max(u[2] for u in units) <= WA:
            A.append((ka, kb - ka))
        else:
            for (k0, nk, um) in units:
                kt = (nk - 1 + um + TK - 1) // TK
                B += [(k0, nk, k0 + e * TK) for e in range(kt)]
    return A, B

def emu_cost_tiles(F, n, S, stage, TI, WA, TK, start0=0):
    """k_cost tile decomposition (start-major) for one chunk: yields pairs (k,i) per tile; checks E/S array bounds."""
    A, B = emu_tile_plan(F, n, S, stage, TI, WA, TK)
    for split, tiles in ((False, [(ka, nk, 0) for (ka, nk) in A]), (True, B)):
        KS = TK + 1 if split else TI + NARROW_WMAX + 1
        IS = 17 if split else 0
        for (ka, nk, et_lo) in tiles:
            kb = ka + nk
            if not split: et_lo = 0
            et_hi = et_lo + TK if split else (1 << 30)
            pairs = []; imin = None; imax = None
            for kl in range(nk):
                k = ka + kl
                is_ = max(k, et_lo); ie = min(k + F[k], et_hi) - 1
                if ie >= is_:
                    imin = is_ if imin is None else min(imin, is_); imax = ie if imax is None else max(imax, ie)
                    pairs += [(k, i) for i in range(is_, ie + 1)]
            if not pairs: continue
            eA = imin + 1 if split else ka
            eG = group_start(start0, eA); assert 0 <= eA - eG <= CARRY_G - 1
            Ecnt = imax + 2 - eA
            assert 0 < Ecnt <= KS, (Ecnt, KS)
            if not split:                                      # narrow: everything a window could reach is staged, cut to the chunk
                nsite = min(KS - 1, n - ka)
                assert Ecnt - 1 <= nsite and nsite + ((start0 + ka) & 3) <= 128
            if split:
                sA = ka; Scnt = kb - sA
                assert Scnt <= IS
                for (k, i) in pairs: assert 0 <= i + 1 - eA < Ecnt and 0 <= k - sA < Scnt
            else:
                for (k, i) in pairs: assert 0 <= i + 1 - eA < Ecnt and 0 <= k - eA < Ecnt
            assert len(pairs) <= 4096, len(pairs)
            yield pairs

NEG = -np.inf

def emu_dp(F, cum, cost, n, S, BL=32):
    """k_dp<NW, BL> over all stages: push form with two pending registers per lane, worker pushes for blocks > 128
    sites one batch behind the recurrence, ring merge at the start of the finishing batch.  Returns back array."""
    Fmax = int(F.max())
    ringN = 1
    while ringN < Fmax + 2 * BL: ringN <<= 1
    rmask = ringN - 1
    pendB = np.full(ringN, NEG); pendA = np.zeros(ringN, dtype=np.int64)
    bestA = np.full(64, NEG); argA = np.zeros(64, dtype=np.int64)
    bestB = np.full(64, NEG); argB = np.zeros(64, dtype=np.int64)
    Mring = np.full(128, np.nan)
    back = np.zeros(n, dtype=np.int64)
    Mk = 0.0
    def far(base, s1):
        # workers: far pushes (j >= 128) of the sources of batch [base, base+BL)
        ks = [k for k in range(base, min(base + BL, s1))]
        fm = max([int(F[k]) for k in ks] + [0])
        if fm <= 128: return []
        touched = []
        for t in range(base + 128, base + BL - 1 + fm):
            b_, a_ = pendB[t & rmask], pendA[t & rmask]
            for k in ks:
                j = t - k
                if j >= 128 and j < F[k]:
                    cand = Mring[k & 127] + cost[cum[k] + j]
                    if cand > b_: b_ = cand; a_ = k
            pendB[t & rmask] = b_; pendA[t & rmask] = a_; touched.append(t)
        return touched
    for s0 in range(0, n, S):
        s1 = min(s0 + S, n)
        nb = (s1 - s0 + BL - 1) // BL
        Mring[s0 & 127] = Mk
        # pend of batch 0's steps
        lanes0 = [(s0 + x) & 63 for x in range(BL)]
        pb = np.full(64, NEG); pa = np.zeros(64, dtype=np.int64)
        for x in range(BL):
            t = s0 + x; pb[t & 63] = pendB[t & rmask]; pa[t & 63] = pendA[t & rmask]; pendB[t & rmask] = NEG
        for b in range(nb):
            base = s0 + b * BL
            # merge
            for x in range(BL):
                l = (base + x) & 63
                if pb[l] >= bestA[l] and pb[l] > NEG: bestA[l] = pb[l]; argA[l] = pa[l]
            # prefetch the pend entries of batch b+1 (as early as the hardware could do it)
            npb = np.full(64, NEG); npa = np.zeros(64, dtype=np.int64); pre = []
            if b + 1 < nb:
                for x in range(BL):
                    t = base + BL + x; npb[t & 63] = pendB[t & rmask]; npa[t & 63] = pendA[t & rmask]; pendB[t & rmask] = NEG; pre.append(t & rmask)
            # workers: far work of batch b-1, concurrently
            if b >= 1:
                tt = far(base - BL, s1)
                assert not (set(x & rmask for x in tt) & set(pre)), 'far work races with the prefetch'
            # DP wave: BL steps
            wideb = max(int(F[k]) for k in range(base, min(base + BL, s1))) > 64
            Mfin = np.full(64, np.nan)
            for s in range(BL):
                k = base + s
                if k >= s1: break
                f = int(F[k]); stp = k & 63
                for lane in range(64):
                    j = (lane - k) & 63
                    if j < f:
                        cand = Mk + cost[cum[k] + j]
                        if cand > bestA[lane]: bestA[lane] = cand; argA[lane] = k
                    if wideb and j + 64 < f:
                        cand = Mk + cost[cum[k] + j + 64]
                        if cand > bestB[lane]: bestB[lane] = cand; argB[lane] = k
                Mk = bestA[stp]; back[k] = k + 1 - argA[stp]
                Mfin[stp] = Mk
                bestA[stp] = bestB[stp]; argA[stp] = argB[stp]; bestB[stp] = NEG
                Mring[(k + 1) & 127] = Mk
            pb, pa = npb, npa
        # far work of the stage's last batch
        far(s0 + (nb - 1) * BL, s1)
    return back

def plain_dp(F, cum, cost, n):
    """segmentor.cpp:142-154 as written (pull form, ascending k, strict '>')"""
    M = np.zeros(n + 1); back = np.zeros(n, dtype=np.int64)
    # pull form, ascending k strict >
    lo = np.zeros(n, dtype=np.int64)
    for i in range(n):
        best = NEG; arg = -1
        for k in range(max(0, i - 6000), i + 1):
            if i - k < F[k]:
                v = M[k] + cost[cum[k] + i - k]
                if v > best: best = v; arg = k
        M[i + 1] = best; back[i] = i + 1 - arg
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
    TI = TIsel or 64; WA = min(Fmax, NARROW_WMAX); TK = 128
    if S is None: S = ((n + 63) // 64) * 64
    seen = np.zeros(int(F.sum()), dtype=np.int32)
    for stage in range((n + S - 1) // S):
        for pairs in emu_cost_tiles(F, n, S, stage, TI, WA, TK, start0=spec['a']):
            for (k, i) in pairs: seen[cum[k] + i - k] += 1
    assert (seen == 1).all(), 'tile coverage broken'
    cost = np.empty(int(F.sum()))
    for k in range(n): cost[cum[k]:cum[k] + F[k]] = band[k, :F[k]]
    back = emu_dp(F, cum, cost, n, S, 64 if Fmax <= 64 else 32)
    assert (emu_dp(F, cum, cost, n, S, 32) == back).all()
    assert (back == np.arange(1, n + 1) - T[1:]).all(), 'dp emulation differs'
    print(name, 'ok: Fmax', Fmax, 'TI', TI, 'WA', WA, 'S', S)

if __name__ == '__main__':
    # scan carries + staged prefix rows
    rng = np.random.default_rng(0)
    row = rng.integers(0, 200, (5000, 2)); row[:, 0] = np.minimum(row[:, 0], row[:, 1])
    for start0, ln in [(0, 1), (3, 700), (8, 512), (13, 4000), (4990, 10), (1, 64), (7, 129), (64, 200), (63, 2), (100, 1000)]:
        carry = emu_scan_carries(row, start0, ln)
        P = np.concatenate([[[0, 0]], np.cumsum(row[start0:start0 + ln], axis=0)])
        g0 = start0 >> CARRY_SHIFT
        for g in range(carry.shape[0]):
            pos = 0 if g == 0 else ((g0 + g) << CARRY_SHIFT) - start0
            if g > 0 and pos >= ln: continue
            assert (carry[g] == P[pos]).all(), (start0, ln, g, carry[g], P[pos])
        for k in list(range(0, ln, 37)) + [ln - 1]:
            A = group_start(start0, k)
            assert 0 <= k - A <= CARRY_G - 1
            for cnt in (1, 5, 64, 65, 200, 300):
                x0 = k - A
                dst = emu_stage_row(row, carry, start0, ln, A, cnt, x0)
                for x in range(cnt):
                    want = P[min(k + x, ln)]
                    assert (dst[x] == want).all(), (start0, ln, A, cnt, x, dst[x], want)
    # tile-local packed staging of the narrow tiles
    rows = [rng.integers(0, 256, (5000, 2)) for _ in range(5)]
    for r in rows: r[:, 0] = np.minimum(r[:, 0], r[:, 1])
    for start0, ln, ka, nsite in [(0, 300, 0, 124), (3, 700, 64, 124), (13, 200, 128, 72), (4990, 10, 0, 10), (1, 64, 63, 1), (7, 129, 64, 65), (4872, 128, 0, 124)]:
        L = emu_stage_local_rows(rows, start0, ln, ka, nsite)
        for rr, r in enumerate(rows):
            P = np.concatenate([[[0, 0]], np.cumsum(r[start0 + ka:start0 + ka + nsite], axis=0)])
            assert ((L[rr] & 0xffff) == P[:, 0]).all() and ((L[rr] >> 16) == P[:, 1]).all(), (start0, ln, ka, nsite, rr)
    print('scan/stage emulation ok')
    # recurrence emulation on random windows with integer costs (ties everywhere), incl. stage boundaries
    rng = np.random.default_rng(1)
    for trial in range(12):
        n = int(rng.integers(1, 700)); Fm = [40, 200, 700][trial % 3]
        F = np.minimum(rng.integers(1, Fm + 1, n), n - np.arange(n))
        if trial % 3 == 1: F = np.maximum(np.where((np.arange(n) // 97) % 2 == 0, np.minimum(F, 30), F), 1)
        cum = np.concatenate([[0], np.cumsum(F)[:-1]])
        cost = -rng.integers(0, 4, int(F.sum())).astype(float)
        want = plain_dp(F, cum, cost, n)
        for S in (64 * ((n + 63) // 64), 64, 192):
            for BL in ((32, 64) if F.max() <= 64 else (32,)):
                assert (emu_dp(F, cum, cost, n, S, BL) == want).all(), (trial, n, S, BL)
    print('recurrence emulation ok')
    for nm in ['tiny', 'n1', 'n2', 'n65', 'max_cpg2', 'max_cpg_binds', 'dense_w_gt_64', 'dense_small_bp', 'equal_loci', 'island_mix']:
        run_case(nm)
    run_case('tiny', S=64); run_case('dense_w_gt_64', S=128); run_case('max_cpg_binds', S=192)
    run_case('island_mix', TIsel=32); run_case('dense_small_bp', TIsel=16, S=64); run_case('dense_bp500')
