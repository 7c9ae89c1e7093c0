#!/usr/bin/env python3
"""Regenerate the golden vectors under tests/golden/ from the REFERENCE ITSELF.  Runs only in the build
container (needs /root/reference); the fixtures it writes are data (specs, input checksums, expected outputs).

  chunk_cases.json   border lists printed by the reference `segmentor` (oracle/_ref/segmentor = the reference's
                     src/segment_betas/{main,segmentor}.cpp compiled where they lie with setup.py:58's flags)
                     on the seeded inputs of tests/cases.py.
  driver_cases.json  outputs of the reference's own Python driver functions, imported from
                     /root/reference/src/python/segment.py (break_to_chunks, stitch_2_dfs, merge2, find_dups,
                     increase_patch, SegmentByChunks.run with its chunk subprocess replaced by the binary above
                     fed from our loci array instead of tabix).

  offset_cases.json  the same binary with `-s start0 -n len` on whole-world files: chunks placed on the kernels' boundaries
                     (tests/cases.py OFFSET_CASES).

Usage:  python tests/golden/make_golden.py [chunks] [driver] [offsets]
"""
import json
import os
import os.path as op
import sys
import tempfile

import numpy as np

HERE = op.dirname(op.abspath(__file__))
ROOT = op.dirname(op.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, op.join(ROOT, 'tests'))

import cases                                   # noqa: E402
from oracle import oracle                      # noqa: E402


def gen_chunk_cases():
    oracle.build(ref=True)
    assert oracle.have_ref(), 'reference binary not built'
    out = {}
    only = [a for a in sys.argv[1:] if a.startswith('case=')]
    if only:                                               # add / refresh single cases without touching the others
        with open(op.join(HERE, 'chunk_cases.json')) as f:
            out = json.load(f)
        for o in only:
            name = o[5:]
            spec = cases.CHUNK_CASES[name]
            slices, loci = cases.build_case(spec)
            b = oracle.ref_segment_arrays(slices, loci, spec['pcount'], spec['max_cpg'], spec['max_bp'])
            out[name] = dict(spec=spec, input_crc32=cases.case_checksum(slices, loci), borders=b.tolist())
            print('%-16s n=%-6d N=%-3d borders=%d' % (name, spec['n'], len(spec['samples']), len(b)), flush=True)
        with open(op.join(HERE, 'chunk_cases.json'), 'w') as f:
            json.dump(out, f, separators=(',', ':'))
        return
    for name, spec in cases.CHUNK_CASES.items():
        slices, loci = cases.build_case(spec)
        b = oracle.ref_segment_arrays(slices, loci, spec['pcount'], spec['max_cpg'], spec['max_bp'])
        out[name] = dict(spec=spec, input_crc32=cases.case_checksum(slices, loci), borders=b.tolist())
        print('%-16s n=%-6d N=%-3d borders=%d' % (name, spec['n'], len(spec['samples']), len(b)), flush=True)

    # chr21-shaped multi-chunk case: per-chunk border lists from the reference binary on whole-file betas
    spec = cases.CHR21
    slices, loci = cases.build_case(spec)
    n, step = spec['n'], spec['chunk']
    starts = list(range(0, n, step))
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for i, s in enumerate(slices):
            p = op.join(td, 's%02d.beta' % i)
            s.tofile(p)
            paths.append(p)
        per_chunk = []
        for st in starts:
            ln = min(step, n - st)
            b = oracle.ref_segment_chunk(paths, st, ln, loci[st:st + ln], spec['pcount'], spec['max_cpg'], spec['max_bp'])
            per_chunk.append(b.tolist())
            print('chr21 chunk @%d len %d borders=%d' % (st, ln, len(b)), flush=True)
    out['chr21'] = dict(spec=spec, input_crc32=cases.case_checksum(slices, loci), starts=starts, borders=per_chunk)
    with open(op.join(HERE, 'chunk_cases.json'), 'w') as f:
        json.dump(out, f, separators=(',', ':'))
    print('wrote chunk_cases.json')


def gen_offset_cases():
    """offset_cases.json: chunks inside a larger world, `segmentor ... -s start0 -n len` on the whole-world .beta files."""
    oracle.build(ref=True)
    assert oracle.have_ref(), 'reference binary not built'
    out = {}
    for name, spec in cases.OFFSET_CASES.items():
        slices, loci = cases.build_case(spec)
        chunks = cases.offset_chunks(spec)
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for i, s in enumerate(slices):
                p = op.join(td, 's%02d.beta' % i)
                s.tofile(p)
                paths.append(p)
            borders = [oracle.ref_segment_chunk(paths, st, ln, loci[st:st + ln], spec['pcount'], spec['max_cpg'], spec['max_bp']).tolist()
                       for st, ln in chunks]
        out[name] = dict(spec=spec, input_crc32=cases.case_checksum(slices, loci), chunks=chunks, borders=borders)
        print('%-16s world %d x %d, %d chunks, %d borders' % (name, spec['n'], len(spec['samples']), len(chunks), sum(map(len, borders))), flush=True)
    with open(op.join(HERE, 'offset_cases.json'), 'w') as f:
        json.dump(out, f, separators=(',', ':'))
    print('wrote offset_cases.json')


if __name__ == '__main__':
    what = [a for a in sys.argv[1:] if not a.startswith('case=')] or ['chunks', 'driver']
    if 'chunks' in what:
        gen_chunk_cases()
    if 'driver' in what:
        import make_golden_driver
        make_golden_driver.gen_driver_cases()
    if 'offsets' in what:
        gen_offset_cases()
