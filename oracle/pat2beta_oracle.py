"""oracle/pat2beta_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Checkers for the pat -> beta counting (k_pat_count / k_pat_trim behind wgbsseg_patbeta_*):
  * ref_counts()    the reference's own stdin2beta binary (oracle/_ref/stdin2beta = src/pat2beta/stdin2beta.cpp compiled where it
                    lies by oracle/Makefile with setup.py:42's flags) fed the pat text on stdin, as pat2beta.py:35-38 does
  * counts()        a numpy restatement of stdin2beta.cpp:59-93 (proc_line) for where the binary is absent
  * trim()          utils_wgbs.py:277-290 (shared with the block reduction's oracle)
Parity: pinned by tests/test_pat2beta_cpu.py (restatement == reference binary on seeded pat text; digests committed in
tests/golden/pat_cases.json).  Only tests/ may import this module.
"""
import os
import os.path as op
import subprocess

import numpy as np

from .block_sums import trim          # noqa: F401

HERE = op.dirname(op.abspath(__file__))
REF_BIN = op.join(HERE, '_ref', 'stdin2beta')


def have_ref():
    return op.isfile(REF_BIN) and os.access(REF_BIN, os.X_OK)


def ref_counts(text, start, end):
    """int64 [end - start, 2] (#meth, #cov) as the reference binary prints them; None when it gives up (malformed line)."""
    r = subprocess.run([REF_BIN, str(start), str(end)], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out = r.stdout.decode()
    if not out.strip():
        return None
    return np.array(out.split(), dtype=np.int64).reshape(-1, 2)


def counts(lines, start, end):
    """stdin2beta.cpp:59-93 for an iterable of text lines"""
    n = end - start
    meth = np.zeros(n, dtype=np.int64)
    cov = np.zeros(n, dtype=np.int64)
    for line in lines:
        if not line:
            continue
        tok = line.split('\t')
        if len(tok) < 4:
            return None
        site, pat, cnt = int(tok[1]), tok[2], int(tok[3])
        if site + len(pat) - 1 < start or site >= end:
            continue
        for i, ch in enumerate(pat):
            x = site - start + i
            if 0 <= x < n and ch in 'TCH':
                cov[x] += cnt
                if ch != 'T':
                    meth[x] += cnt
    return np.stack([meth, cov], axis=1)
