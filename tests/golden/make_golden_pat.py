#!/usr/bin/env python3
"""Golden vectors of pat2beta from the REFERENCE ITSELF: seeded synthetic pat text through the
reference's stdin2beta binary (oracle/_ref/stdin2beta, built from its own source by oracle/Makefile) and the reference's own
trim_to_uint8 (imported from /root/reference/src/python/utils_wgbs.py).  Writes tests/golden/pat_cases.json: per case the
generator parameters and sha1 digests of the .beta and .lbeta bytes (+ the first rows in full)."""
import hashlib
import json
import os.path as op
import sys

import numpy as np

HERE = op.dirname(op.abspath(__file__))
ROOT = op.dirname(op.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference/src/python')

from oracle import pat2beta_oracle as OP          # noqa: E402
from wgbs_tools_amd import synth                   # noqa: E402

CASES = {'small': dict(seed=11, n_sites=3000, n_reads=20000),            # deep coverage: many sites beyond 255 and a few beyond 65535? (no: see 'deep')
         'sparse': dict(seed=12, n_sites=200000, n_reads=150000),
         'deep': dict(seed=13, n_sites=40, n_reads=60000)}               # coverage sums in the tens of thousands: both trims bite


def main():
    import utils_wgbs as ru
    assert OP.have_ref(), 'make -C oracle ref first'
    out = {}
    for name, spec in CASES.items():
        lines = synth.synth_pat_lines(spec['seed'], spec['n_sites'], spec['n_reads'])
        text = ('\n'.join(lines) + '\n').encode()
        arr = OP.ref_counts(text, 1, spec['n_sites'] + 1)
        rec = dict(spec=spec, text_sha1=hashlib.sha1(text).hexdigest(), max_cov=int(arr[:, 1].max()))
        for lbeta, tag in ((False, 'beta'), (True, 'lbeta')):
            b = ru.trim_to_uint8(arr.copy(), lbeta)
            rec[tag + '_sha1'] = hashlib.sha1(b.tobytes()).hexdigest()
            rec[tag + '_head'] = b[:16].tolist()
        out[name] = rec
        print(name, spec, 'max cov', rec['max_cov'])
    with open(op.join(HERE, 'pat_cases.json'), 'w') as f:
        json.dump(out, f, separators=(',', ':'))


if __name__ == '__main__':
    main()
