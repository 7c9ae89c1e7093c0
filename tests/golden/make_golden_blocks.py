#!/usr/bin/env python3
"""Golden vectors of the block reduction from the REFERENCE ITSELF: runs only in the build
container, imports /root/reference/src/python/{beta_to_blocks,beta_to_table}.py and records what they produce on
seeded inputs (tests/cases.py generators).  The fixture holds the blocks tables, and the expected .bin / .lbeta bytes,
bedGraph text and beta_to_table text.

Usage:  python tests/golden/make_golden_blocks.py
"""
import base64
import io
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
sys.path.insert(0, '/root/reference/src/python')

from wgbs_tools_amd import synth               # noqa: E402
import cases                                   # noqa: E402

N_SITES = 40000
SAMPLES = [0, 1, 2, 3]


def blocks_tables(rng):
    """name -> list of (chr, start, end, startCpG, endCpG) with None for NA"""
    out = {}
    # nice: sorted, disjoint, with gaps, short and long blocks (cov sums beyond 255 and, for a few, beyond 65535)
    rows, pos = [], 5
    while pos < N_SITES - 4000:
        r = rng.random()
        ln = int(rng.integers(1, 30)) if r < 0.9 else (int(rng.integers(30, 400)) if r < 0.995 else int(rng.integers(2300, 3500)))
        gap = int(rng.integers(0, 4)) if rng.random() < 0.5 else 0
        rows.append(('chr1', 1000 + 10 * pos, 1000 + 10 * (pos + ln), pos, pos + ln))
        pos += ln + gap
    rows.append(('chr1', 1000 + 10 * pos, 1000 + 10 * (N_SITES + 1), pos, N_SITES + 1))        # up to the last site
    out['nice'] = rows
    # not nice: NA rows, overlaps, duplicates, unsorted, an empty block
    rows2 = []
    for _ in range(600):
        a = int(rng.integers(1, N_SITES - 600))
        ln = int(rng.integers(1, 500))
        rows2.append(('chr1', 1000 + 10 * a, 1000 + 10 * (a + ln), a, a + ln))
    rows2[10] = ('chr1', 5, 50, None, None)
    rows2[11] = rows2[12]
    rows2[13] = ('chr1', 700, 700, 70, 70)
    out['ragged'] = rows2
    return out


def text_record(text):
    import hashlib
    return {'sha1': hashlib.sha1(text.encode()).hexdigest(), 'len': len(text), 'head': text[:600]}


def write_blocks(path, rows):
    with open(path, 'w') as f:
        for c, s, e, a, b in rows:
            f.write('%s\t%d\t%d\t%s\t%s\n' % (c, s, e, 'NA' if a is None else a, 'NA' if b is None else b))


def main():
    import hashlib
    import beta_to_blocks as rb
    import beta_to_table as rt
    rng = np.random.default_rng(20260926)
    tables = blocks_tables(rng)
    fixture = {'n_sites': N_SITES, 'samples': SAMPLES, 'seed': cases.SEED, 'tables': {}}
    with tempfile.TemporaryDirectory() as td:
        betas = []
        for s in SAMPLES:
            p = op.join(td, 'smp%d.beta' % s)
            synth.synth_betas(cases.SEED, s, 0, N_SITES).tofile(p)
            betas.append(p)
        fixture['input_crc32'] = synth.checksum(*[np.fromfile(b, dtype=np.uint8) for b in betas])
        lbetas = []
        for s in SAMPLES:                                       # uint16 twins of the samples: (meth, cov) * k(site), k <= 997
            p = op.join(td, 'smp%d.lbeta' % s)
            cases.lbeta_twin(synth.synth_betas(cases.SEED, s, 0, N_SITES)).tofile(p)
            lbetas.append(p)
        gpath = op.join(td, 'groups.csv')
        with open(gpath, 'w') as f:
            f.write('name,group\nsmp0,A\nsmp1,B\nsmp2,A\nsmp3,B\n')
        for name, rows in tables.items():
            bpath = op.join(td, name + '.bed')
            write_blocks(bpath, rows)
            df = rb.load_blocks_file(bpath)
            is_nice, msg = rb.is_block_file_nice(df)
            rec = {'rows': rows, 'is_nice': bool(is_nice), 'msg': msg, 'bin': {}, 'lbeta': {}, 'bedgraph': {}, 'sums_sha1': {}}
            for b in betas:
                key = op.basename(b)
                raw = rb.collapse_process(b, df.copy(), is_nice)
                import hashlib
                rec['sums_sha1'][key] = hashlib.sha1(np.ascontiguousarray(raw, dtype=np.int64).tobytes()).hexdigest()
                for lbeta in (False, True):
                    od = op.join(td, 'out_%s_%d' % (name, int(lbeta)))
                    os.makedirs(od, exist_ok=True)
                    err = io.StringIO()
                    stderr, sys.stderr = sys.stderr, err
                    try:
                        rb.collapse_process(b, df.copy(), is_nice, lbeta, od, True)
                    finally:
                        sys.stderr = stderr
                    stem = op.join(od, op.splitext(key)[0])
                    data = open(stem + ('.lbeta' if lbeta else '.bin'), 'rb').read()
                    rec['lbeta' if lbeta else 'bin'][key] = base64.b64encode(data).decode()
                    if not lbeta:
                        rec['bedgraph'][key] = text_record(open(stem + '.bedGraph').read())
            # beta_to_table: per-sample table (min_cov 4, 2 digits) and grouped table (min_cov 10, 3 digits)
            for tag, gfile, mc, dg in (('table_plain', None, 4, 2), ('table_groups', gpath, 10, 3)):
                t = rt.betas2table(betas, bpath, gfile, mc, threads=2)
                o = op.join(td, 't.tsv')
                rt.dump(o, t, True, dg)
                rec[tag] = text_record(open(o).read())
            # the same with uint16 .lbeta INPUT files (utils_wgbs.py:311-319): outputs as digests
            lrec = {'bin_sha1': {}, 'lbeta_sha1': {}}
            for lb in lbetas:
                key = op.basename(lb)
                for lbeta in (False, True):
                    od = op.join(td, 'lout_%s_%d' % (name, int(lbeta)))
                    os.makedirs(od, exist_ok=True)
                    stderr, sys.stderr = sys.stderr, io.StringIO()
                    try:
                        rb.collapse_process(lb, df.copy(), is_nice, lbeta, od, False)
                    finally:
                        sys.stderr = stderr
                    stem = op.join(od, op.splitext(key)[0])
                    data = open(stem + ('.lbeta' if lbeta else '.bin'), 'rb').read()
                    lrec['lbeta_sha1' if lbeta else 'bin_sha1'][key] = hashlib.sha1(data).hexdigest()
            t = rt.betas2table(lbetas, bpath, None, 4, threads=2)
            o = op.join(td, 'tl.tsv')
            rt.dump(o, t, True, 3)
            lrec['table_plain'] = text_record(open(o).read())
            rec['lbeta_inputs'] = lrec
            fixture['tables'][name] = rec
            print(name, 'rows', len(rows), 'nice', is_nice, msg)
    with open(op.join(HERE, 'block_cases.json'), 'w') as f:
        json.dump(fixture, f, separators=(',', ':'))
    print('wrote block_cases.json (%.0f KB)' % (op.getsize(op.join(HERE, 'block_cases.json')) / 1e3))


if __name__ == '__main__':
    main()
