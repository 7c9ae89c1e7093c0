"""Index logic of k_scan restated in numpy (tools/emulate_kernels.py): the carries a row stores are the prefixes at the group starts,
and the rule by which the kernel's flush decides — from the group number alone — which LDS-staged entries to write selects exactly the
entries its loop has stored (an entry it had not stored would be garbage in the carries of a wide tile)."""
import os.path as op
import sys

import numpy as np

sys.path.insert(0, op.join(op.dirname(op.dirname(op.abspath(__file__))), 'tools'))


def test_staged_carries_flush_rule_and_values():
    import emulate_kernels as E
    rng = np.random.default_rng(3)
    row = rng.integers(0, 200, (5000, 2))
    row[:, 0] = np.minimum(row[:, 0], row[:, 1])
    for start0 in (0, 1, 15, 16, 17, 112, 127, 128, 129, 1000, 1023, 1024):
        for ln in (1, 16, 113, 128, 129, 1024, 1025, 2100):
            carry = E.emu_scan_carries(row, start0, ln)                      # asserts the flush rule inside
            P = np.concatenate([[[0, 0]], np.cumsum(row[start0:start0 + ln], axis=0)])
            g0 = start0 >> E.CARRY_SHIFT
            for g in range(carry.shape[0]):
                pos = 0 if g == 0 else ((g0 + g) << E.CARRY_SHIFT) - start0
                if g == 0 or pos < ln:
                    assert (carry[g] == P[pos]).all(), (start0, ln, g)
