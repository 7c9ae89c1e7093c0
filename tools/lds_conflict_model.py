#!/usr/bin/env python3
"""Where do the LDS bank conflicts of k_cost come from?  (`lds_conflict_frac` 0.31 of the LDS-array cycles, unexplained.)

A CPU model of the four LDS reads one (block, sample) evaluation of a NARROW scoring tile issues, with the service rule of
MI355X_MICROARCH.md (LDS): a wave64 access is served in fixed lane groups, one LDS cycle per group when conflict-free; only lanes of
the same group conflict, identical addresses broadcast, every further DISTINCT address on a busy bank adds a cycle.
    ds_read_b32   Ep[sl * KS]   packed prefix of the block's end         2 groups of 32 lanes, bank = (a / 4) mod 32
    ds_read_b32   Sp[sl * KS]   packed prefix of the block's start       (the same)
    ds_read_b128  iys0[ki]      log2f table entry {invc 2^-k, logc + k}  4 groups of 16 lanes, bank = (a / 4) mod 64, 4 banks per lane
    ds_read_b128  kys0[kiu]     fast-log2 table entry                    (the same)
The addresses are those the kernel forms (csrc/seg_kernels.h k_cost<64, 3, 0>, csrc/exact_log2.h wg_sample_term_pcpos_ks) on the
bench's synthetic genome: tiles of 64 starts, blocks flattened start-major, thread t of the workgroup takes blocks t, t + 256, ...;
p = fl(fl(nmeth + pc) / fl(ntotal + 2 pc)) in float32, the table indices from the bits of p and of 1 - (double)p.

    python tools/lds_conflict_model.py [n_sites] [n_samples]      -> one JSON object (also written to profiles/r03_lds_conflict_model.json)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wgbs_tools_amd import synth          # noqa: E402

SEED = 20260926
TI, WMAX, KS = 64, 60, 125
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]
B32_GROUPS = [list(range(0, 32)), list(range(32, 64))]


def cycles(addr, active, groups, banks, width_dwords):
    """LDS-array cycles of one wave access: per group, the largest number of distinct addresses on one bank (>= 1 if any lane is active)."""
    tot = 0
    for g in groups:
        a = [int(addr[l]) for l in g if active[l]]
        if not a:
            continue
        per_bank = {}
        for x in set(a):
            for w in range(width_dwords):
                per_bank.setdefault((x // 4 + w) % banks, set()).add(x)
        tot += max(len(v) for v in per_bank.values())
    return tot


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    pc = np.float32(15.0)
    pc2 = np.float32(pc + pc)
    loci = synth.synth_loci(SEED, [n]).astype(np.int64)
    betas = [synth.synth_betas(SEED, s, 0, n).astype(np.int64) for s in range(ns)]
    hi = np.searchsorted(loci, loci + 2000, 'right')
    F = np.minimum(np.minimum(hi - np.arange(n), 1000), n - np.arange(n))
    rows = 11
    TB = rows * 80 * 16
    iys0 = (rows - 1) * 256                       # byte address of the k = 0 row of the log2f table
    ky0 = rows * 256 + (rows - 1) * 1024
    base = {'Ep': 0, 'Sp': 0, 'iy': 0, 'ky': 0}
    extra = {'Ep': 0, 'Sp': 0, 'iy': 0, 'ky': 0}
    n_tiles = 0
    for ka in range(0, n - 200, TI):
        f = F[ka:ka + TI]
        if f.max() > WMAX:
            continue
        n_tiles += 1
        offs = np.concatenate([[0], np.cumsum(f)])
        Q = int(offs[-1])
        q = np.arange(Q)
        lo = np.searchsorted(offs, q, 'right') - 1
        i = ka + lo + (q - offs[lo])               # end site of block q
        # tile-local prefixes of every sample
        span = int(i.max()) + 2 - ka
        for sl in range(ns):
            mt = betas[sl][ka:ka + span - 1]
            Lm = np.concatenate([[0], np.cumsum(mt[:, 0])])
            Lt = np.concatenate([[0], np.cumsum(mt[:, 1])])
            nm = (Lm[i + 1 - ka] - Lm[lo]).astype(np.float32)
            nt = (Lt[i + 1 - ka] - Lt[lo]).astype(np.float32)
            p = ((nm + pc) / (nt + pc2)).astype(np.float32)
            ki = (p.view(np.uint32).astype(np.int64) - 0x3f330000) >> 19
            x = 1.0 - p.astype(np.float64)
            kiu = (x.view(np.uint64) >> np.uint64(32 + 14)).astype(np.int64) - (0x3fe60000 >> 14)
            a_ep = TB + 4 * ((i + 1 - ka) + sl * KS)
            a_sp = TB + 4 * (lo + sl * KS)
            a_iy = iys0 + 16 * ki
            a_ky = ky0 + 16 * kiu
            for r0 in range(0, Q, 256):            # rounds of the workgroup; its four wavefronts
                for w in range(4):
                    l0 = r0 + 64 * w
                    if l0 >= Q:
                        break
                    act = np.arange(l0, l0 + 64) < Q
                    idx = np.minimum(np.arange(l0, l0 + 64), Q - 1)
                    for name, arr, groups, banks, wd in (('Ep', a_ep, B32_GROUPS, 32, 1), ('Sp', a_sp, B32_GROUPS, 32, 1),
                                                         ('iy', a_iy, B128_GROUPS, 64, 4), ('ky', a_ky, B128_GROUPS, 64, 4)):
                        c = cycles(arr[idx], act, groups, banks, wd)
                        b = sum(1 for g in groups if any(act[l] for l in g))
                        base[name] += b
                        extra[name] += c - b
        if n_tiles >= 12:
            break
    tb, te = sum(base.values()), sum(extra.values())
    rec = {'model': 'LDS-array cycles of the four reads of one evaluation of k_cost<64,3,0>, %d narrow tiles of the bench genome x %d samples, pseudo count 15' % (n_tiles, ns),
           'conflict_free_cycles': base, 'extra_cycles': extra,
           'conflict_frac_by_read': {k: extra[k] / (base[k] + extra[k]) for k in base},
           'conflict_frac_all_reads': te / (tb + te),
           'share_of_extra_cycles': {k: extra[k] / te for k in extra},
           'measured': 'profiles/r02_pmc_cost_sq.json: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.31 over the whole kernel (staging and table copies included)'}
    out = os.path.join(ROOT, 'profiles', 'r03_lds_conflict_model.json')
    json.dump(rec, open(out, 'w'), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    main()
