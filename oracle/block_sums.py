"""oracle/block_sums.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy) of the reference's block reduction, for checking the HIP path (k_block_sums):
    block_sums   beta_to_blocks.py:101-126  fast_method / slow_method: per block the sums of the (meth, cov) rows
                 data[startCpG-1 : endCpG-1]; NA rows give (0, 0) (slow_method :112-114)
    trim         utils_wgbs.py:277-290      trim_to_uint8: rows with cov > max -> (trunc(meth / cov * max), max)
    beta2vec     utils_wgbs.py:270-274      meth / cov in float64, NaN where cov < min_cov
Pinned by tests/test_blocks_cpu.py against tests/golden/block_cases.json, which tests/golden/make_golden_blocks.py
captured from the reference's own Python on seeded inputs.  Only tests/ may import this module.
"""
import numpy as np


def block_sums(data, start0, end0):
    """data: uint8 [n, 2]; start0/end0: 0-based half-open site ranges (empty range -> 0, 0).  -> int64 [n_blocks, 2]"""
    P = np.concatenate([np.zeros((1, 2), dtype=np.int64), np.cumsum(data.astype(np.int64), axis=0)])
    s = np.asarray(start0, dtype=np.int64)
    e = np.asarray(end0, dtype=np.int64)
    out = P[np.maximum(e, s)] - P[s]
    return out


def trim(table, lbeta=False):
    max_val = 65535 if lbeta else 255
    t = np.array(table, dtype=np.int64)
    big = t[:, 1] > max_val
    t[big, 0] = (t[big, 0] / t[big, 1] * max_val).astype(np.int64)          # float64 divide, multiply, truncate
    t[big, 1] = max_val
    return t.astype(np.uint16 if lbeta else np.uint8)


def beta2vec(table, min_cov=1):
    t = np.asarray(table, dtype=np.int64)
    cond = t[:, 1] >= min_cov
    vec = np.full(t.shape[0], np.nan)
    vec[cond] = t[cond, 0] / t[cond, 1]
    return vec
