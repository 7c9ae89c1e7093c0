#!/usr/bin/env python3
"""disorder_cases.json: border lists printed by the REFERENCE `segmentor` (oracle/_ref/segmentor = the reference's
src/segment_betas/{main,segmentor}.cpp compiled where they lie) for loci that are not ascending inside the chunk
(tests/cases.py DISORDER_CASES): what segmentor.cpp:114-117 does with them.  Runs only in the build container.

Usage:  python tests/golden/make_golden_disorder.py
"""
import json
import os.path as op
import sys

HERE = op.dirname(op.abspath(__file__))
ROOT = op.dirname(op.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, op.join(ROOT, 'tests'))

import cases                                   # noqa: E402
from oracle import oracle                      # noqa: E402

oracle.build(ref=True)
assert oracle.have_ref(), 'reference binary not built'
out = {}
for name, spec in cases.DISORDER_CASES.items():
    slices, loci = cases.build_disorder_case(spec)
    b = oracle.ref_segment_arrays(slices, loci, spec['pcount'], spec['max_cpg'], spec['max_bp'])
    spec_j = dict(spec, disorder=list(spec['disorder']))
    out[name] = dict(spec=spec_j, input_crc32=cases.case_checksum(slices, loci), borders=b.tolist())
    print('%-16s n=%-5d N=%-3d borders=%d' % (name, spec['n'], len(spec['samples']), len(b)), flush=True)
with open(op.join(HERE, 'disorder_cases.json'), 'w') as f:
    json.dump(out, f, separators=(',', ':'))
