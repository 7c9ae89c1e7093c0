#!/usr/bin/env python3
"""Golden vectors of `wgbstools convert` (BED <-> CpG-index join) from the REFERENCE ITSELF:
runs only in the build container, imports /root/reference/src/python/{convert,genomic_region}.py and records what they
produce on a seeded synthetic genome.  What is replaced, and only that:
  * GenomeRefPaths (both modules)  -> an object over our synthetic genome directory (the reference resolves names under
                                      its own source tree, which is read-only here)
  * convert.load_dict_section      -> the rows `tabix CpG.bed.gz <chrom>` would print, from our loci array
  * convert.Pool                   -> an in-process pool
  * `tabix` on PATH                -> a 25-line filter over the text dictionary (the image has no htslib), used by the
                                      reference's own shell pipelines in genomic_region.py (tabix | awk ...)
Everything else — load_bed, add_cpgs_to_bed, chr_thread (merge_asof joins, end rules, drops), slow_conversion,
GenomicRegion's parsing, range rules and __str__ — is the reference's own code.  Writes tests/golden/convert_cases.json.
Round 6: the fixture says which vectors went through the `tabix` stand-in (`via_tabix_shim`), and holds a cross-check of the stand-in
against the reference's own tabix-free path (`shim_cross_check`: 400 random regions through the pipeline AND through chr_thread's joins).
"""
import contextlib
import gzip
import io
import json
import os
import os.path as op
import stat
import sys
import tempfile

import numpy as np

HERE = op.dirname(op.abspath(__file__))
ROOT = op.dirname(op.dirname(HERE))
REF_PY = '/root/reference/src/python'
SEED = 20260928
CHROMS = [('chr1', 30011), ('chr2', 20002), ('chrX', 9000), ('chrM', 5)]

TABIX_SHIM = r'''#!/usr/bin/env python3
import gzip, sys
path, region = sys.argv[-2], sys.argv[-1]
chrom, lo, hi = region, None, None
if ':' in region:
    chrom, rng = region.split(':')
    lo, hi = (int(x) for x in rng.replace(',', '').split('-'))
col = 2 if path.split('/')[-1].startswith('rev.') else 1
try:
    with gzip.open(path, 'rt') as f:
        for line in f:
            t = line.rstrip('\n').split('\t')
            if t[0] != chrom:
                continue
            if lo is None or lo <= int(t[col]) <= hi:
                sys.stdout.write(line)
except BrokenPipeError:
    pass
'''


def make_world(td):
    sys.path.insert(0, ROOT)
    from wgbs_tools_amd import synth
    names = [c for c, _ in CHROMS]
    sizes = [s for _, s in CHROMS]
    loci = synth.synth_loci(SEED, sizes)
    ref = synth.write_genome(op.join(td, 'references', 'synth'), names, sizes, loci)
    shim = op.join(td, 'bin')
    os.makedirs(shim)
    with open(op.join(shim, 'tabix'), 'w') as f:
        f.write(TABIX_SHIM)
    os.chmod(op.join(shim, 'tabix'), os.stat(op.join(shim, 'tabix')).st_mode | stat.S_IEXEC)
    os.environ['PATH'] = shim + os.pathsep + os.environ['PATH']
    return names, sizes, loci, ref


def bed_cases(names, sizes, loci, rng):
    """name -> list of text rows"""
    cum = np.concatenate([[0], np.cumsum(sizes)])
    out = {}

    def chrom_rows(ci, n, overlap=False, extra=False):
        lo, hi = int(cum[ci]), int(cum[ci + 1])
        L = loci[lo:hi].astype(np.int64)
        rows, pos = [], int(L[0]) - 50
        for k in range(n):
            kind = rng.integers(0, 8)
            a = pos + int(rng.integers(0, 400))
            b = a + int(rng.integers(1, 3000))
            if kind == 0:                                   # start exactly on a CpG
                a = int(L[min(np.searchsorted(L, a), L.size - 1)])
                b = max(b, a + 1)
            elif kind == 1:                                 # end exactly on a CpG
                b = int(L[min(np.searchsorted(L, b), L.size - 1)])
                if b <= a:
                    b = a + 1
            elif kind == 2:                                 # a short region, likely without CpGs
                b = a + int(rng.integers(1, 4))
            elif kind == 3:                                 # zero-length
                b = a
            r = [names[ci], str(a), str(b)]
            if extra:
                r += ['name%d' % k, '%.2f' % rng.random()]
            rows.append(r)
            pos = (a - int(rng.integers(0, 1500))) if (overlap and k % 3 == 1) else b
        return rows
    clean = chrom_rows(0, 60, extra=True) + chrom_rows(1, 45, extra=True) + chrom_rows(2, 20, extra=True)
    last = int(loci[cum[1] - 1])
    clean += [['chr1', str(last + 10), str(last + 500), 'beyond', '0.5'],            # after the last CpG of chr1
              ['chr1', str(last - 5), str(last + 500), 'tail', '0.5'],               # runs past the last CpG
              ['chr9', '100', '5000', 'unknown_chrom', '0.1'],
              ['chrM', '1', '99999999', 'wholeM', '0.2'],
              clean[3], clean[70]]                                                     # duplicates
    order = rng.permutation(len(clean))
    out['clean_shuffled'] = [clean[i] for i in order]
    out['three_columns_sorted'] = [r[:3] for r in chrom_rows(0, 40) + chrom_rows(1, 10)]
    out['overlaps_in_chr2'] = [r[:3] for r in chrom_rows(0, 25)] + [r[:3] for r in chrom_rows(1, 40, overlap=True)] + \
                              [['chr2', '5', '3'], ['chr2', '0', '100']]             # end before start, start < 1: NA on the slow path
    out['with_header'] = [['chrom', 'chromStart', 'chromEnd']] + [r[:3] for r in chrom_rows(2, 12)]
    return out


def main():
    td = tempfile.mkdtemp()
    names, sizes, loci, refdir = make_world(td)
    import pandas as pd
    sys.path.insert(0, REF_PY)
    import convert as rc
    import genomic_region as rg
    total = int(sum(sizes))
    cum = np.concatenate([[0], np.cumsum(sizes)])

    class FakeGenome:
        def __init__(self, name=None):
            self.genome = 'synth'
            self.dict_path = op.join(refdir, 'CpG.bed.gz')
            self.revdict_path = op.join(refdir, 'rev.CpG.bed.gz')
            self.annotations = None
            self.ilmn2cpg_dict = None

        def get_chrom_cpg_size_table(self):
            return pd.DataFrame({'chr': names, 'size': sizes})

        def get_chrom_size_table(self):
            return pd.read_csv(op.join(refdir, 'chrome.size'), sep='\t', header=None, names=['chr', 'size'])

        def get_chroms(self):
            return tuple(names)

        def get_nr_sites(self):
            return total

    def load_dict_section(region, genome_name=None):
        ci = names.index(region)
        lo, hi = int(cum[ci]), int(cum[ci + 1])
        return pd.DataFrame({'chr': region, 'start': loci[lo:hi].astype(np.int64), 'idx': np.arange(lo + 1, hi + 1)})

    class FakePool:
        def __init__(self, n):
            pass

        def starmap(self, f, ps):
            return [f(*p) for p in ps]

        def close(self):
            pass

        def join(self):
            pass

    rc.GenomeRefPaths = FakeGenome
    rg.GenomeRefPaths = FakeGenome
    rc.load_dict_section = load_dict_section
    rc.Pool = FakePool

    rng = np.random.default_rng(SEED)
    fixture = {'seed': SEED, 'chroms': CHROMS, 'bed': {}, 'regions': {}, 'sites': {}}
    for name, rows in bed_cases(names, sizes, loci, rng).items():
        p = op.join(td, name + '.bed')
        with open(p, 'w') as f:
            for r in rows:
                f.write('\t'.join(r) + '\n')
        rec = {'rows': rows}
        for drop in (False, True):
            err = io.StringIO()
            with contextlib.redirect_stderr(err):
                r = rc.add_cpgs_to_bed(bed_file=p, genome='synth', drop_empty=drop, threads=1, add_anno=False)
            buf = io.StringIO()
            r.to_csv(buf, sep='\t', header=None, index=None, na_rep='NA')
            rec['drop_empty' if drop else 'keep'] = {'text': buf.getvalue(), 'stderr': err.getvalue()}
        fixture['bed'][name] = rec
        print(name, len(rows), 'rows ->', rec['keep']['text'].count('\n'), 'lines;', rec['keep']['text'].count('NA') // 2, 'NA;',
              repr(rec['keep']['stderr'][:60]))
    # single regions / site ranges through GenomicRegion (its tabix | awk pipelines run for real on the shim)
    L1 = loci[:sizes[0]].astype(np.int64)
    regs = ['chr1:%d-%d' % (L1[10], L1[20]), 'chr1:%d-%d' % (L1[10] + 1, L1[20] - 1), 'chr1:%d-%d' % (L1[100], L1[100] + 1),
            'chr2', 'chrM', 'chr1:%d' % L1[7], 'chr1:%d-%d' % (L1[5] - 3, L1[5] + 1), 'chr1:1-%d' % (L1[0] - 1), 'chr1:5-3', 'chr7:1-100',
            'chr1:%d-%d' % (L1[-1] - 10, L1[-1] + 9000), 'chrX:1,000-90,000']
    for r in regs:
        try:
            g = rg.GenomicRegion(region=r, genome_name='synth')
            fixture['regions'][r] = {'sites': list(g.sites), 'str': str(g), 'region_str': g.region_str}
        except rg.IllegalArgumentError as e:
            fixture['regions'][r] = {'error': str(e)}
    for s in ['1-2', '15-25', '30011-30012', '30011-30013', '30000-30011', str(total), '%d-%d' % (total, total + 1), '7', '0-5',
              '%d-%d' % (total, total + 2), '1,000-2,000']:
        err = io.StringIO()
        try:
            with contextlib.redirect_stderr(err):
                g = rg.GenomicRegion(sites=s, genome_name='synth')
            fixture['sites'][s] = {'sites': list(g.sites), 'str': str(g), 'region_str': g.region_str}
        except rg.IllegalArgumentError as e:
            fixture['sites'][s] = {'error': str(e), 'stderr': err.getvalue()}
    # Which vectors went through the `tabix` stand-in, and a cross-check of the stand-in that does not: the reference has TWO code paths from a region
    # to its sites — GenomicRegion's `tabix | awk` pipeline (the stand-in answers the tabix half) and chr_thread's merge_asof joins (pure pandas over
    # our loci array: no tabix anywhere).  For a region that holds a CpG they agree by the reference's own rules, except that a CpG exactly AT the
    # region's end is left out by the pipeline (genomic_region.py:147-150) and kept by the joins (convert.py:169): on 400 random regions the two paths
    # are compared here, and every vector is stored so that the suite can hold the product against both.
    fixture['via_tabix_shim'] = {'bed': {'clean_shuffled': False, 'three_columns_sorted': False, 'with_header': False,
                                         'overlaps_in_chr2': 'the rows of chr2 only (overlaps: slow_conversion -> GenomicRegion); its chr1 rows take chr_thread'},
                                 'regions': True, 'sites': True, 'shim_free_cross_check': False}
    rng2 = np.random.default_rng(SEED + 1)
    cf = pd.DataFrame({'chr': names, 'size': np.cumsum(sizes)})
    rows, agree, n_end_on = [], 0, 0
    for k in range(400):
        ci = int(rng2.integers(0, 3))
        L = loci[cum[ci]:cum[ci + 1]].astype(np.int64)
        i0 = int(rng2.integers(0, L.size - 40)); i1 = i0 + int(rng2.integers(1, 39))
        a = int(L[i0]) - int(rng2.integers(0, 2)) * int(rng2.integers(0, max(1, min(50, int(L[i0]) - (int(L[i0 - 1]) if i0 else 0) - 1))))
        end_on = bool(rng2.integers(0, 3) == 0)
        b = int(L[i1]) if end_on else int(L[i1]) + 1 + int(rng2.integers(0, max(1, int(L[i1 + 1]) - int(L[i1]) - 1)))
        if b <= a:
            continue
        g = rg.GenomicRegion(region='%s:%d-%d' % (names[ci], a, b), genome_name='synth')                    # the stand-in answers
        one = pd.DataFrame({'chr': [names[ci]], 'start': [a], 'end': [b]})
        t = rc.chr_thread(one.copy(), cf, 'synth')                                                           # no tabix
        fast = (int(t['startCpG'].values[0]), int(t['endCpG'].values[0]))
        slow = (int(g.sites[0]), int(g.sites[1]))
        n_end_on += end_on
        agree += slow == (fast[0], fast[1] - (1 if end_on else 0))
        rows.append([names[ci], a, b, slow[0], slow[1], fast[0], fast[1], int(end_on)])
    fixture['shim_cross_check'] = {'rows': rows, 'n': len(rows), 'agree': agree, 'end_on_a_cpg': n_end_on,
                                   'columns': ['chr', 'start', 'end', 'pipeline_startCpG', 'pipeline_endCpG', 'joins_startCpG', 'joins_endCpG', 'end_on_a_cpg']}
    print('tabix stand-in against the reference\'s own tabix-free joins: %d / %d regions agree (%d end on a CpG: pipeline = joins - 1 there)' % (agree, len(rows), n_end_on))
    assert agree == len(rows)
    with open(op.join(HERE, 'convert_cases.json'), 'w') as f:
        json.dump(fixture, f, separators=(',', ':'))
    print('regions:', {k: v.get('sites', v.get('error')) for k, v in fixture['regions'].items()})
    print('sites:', {k: v.get('region_str', v.get('error')) for k, v in fixture['sites'].items()})
    print('wrote convert_cases.json (%.0f KB)' % (op.getsize(op.join(HERE, 'convert_cases.json')) / 1e3))


if __name__ == '__main__':
    main()
