"""`wgbstools convert` on MI355X: genomic loci <-> CpG indexes, the step users run right before
`segment -L` and `beta_to_blocks`.

Drop-in for the reference's src/python/convert.py (same flags, table layout, messages), written against the formats and
the reference's join rules:

    -L BED      tab-separated, >= 3 columns chr, start, end (+ anything), optional header line, '#' comments.  Output: chr,
                start, end, startCpG, endCpG, then the other input columns; NA where a region holds no CpG (or its
                chromosome is unknown); --drop_empty removes those rows.  Input order and duplicate rows are kept.
                Per chromosome the reference uses one of two rule sets (convert.py:147-185 / :133-145): as-of joins against
                the CpG dictionary when the chromosome's regions do not overlap, one GenomicRegion per row when they do; both are
                evaluated on the GPU against the resident loci (wgbsseg_convert_regions, include/wgbsseg.h).
    --site_file one or two columns startCpG [endCpG] -> chr, start, end, startCpG, endCpG: the reference pipes the file through
                its add_loci binary (convert.py:242-248); here wgbsseg_add_loci, which restates it.
    -r / -s     one region / site range -> its description (genomic_region.py:239-247), or the bare range with -p.

The reference needs tabix (and optionally bedtools) for all of this; here the loci column of the dictionary is one array,
resident on the device for the joins.  Annotations (bedtools + an annotation track) are not produced.  No CPU fallback.
"""
import argparse
import os
import os.path as op
import re
import sys

import numpy as np

from .genome import GenomeRefPaths, GenomicRegion, IllegalArgumentError, eprint, write_bed
from .cliutil import NA_TOKENS, add_threads_option, add_where_options

_INT = re.compile(r'^[+-]?\d+$')


def delete_or_skip(output_file, force):
    """utils_wgbs.py:435-454: False iff the output exists and must not be overwritten."""
    if output_file is None or output_file is sys.stdout or output_file == '/dev/stdout':
        return True
    if op.isfile(output_file):
        if not force:
            eprint(f'File {output_file} already exists. Skipping it. Use [-f] flag to force overwrite.')
            return False
        for f in (output_file, output_file + '.csi'):
            if op.isfile(f):
                os.remove(f)
    return True


def _float_text(tok):
    v = float(tok)
    if v != v:
        return 'NA'
    return repr(v)


def column_text(tokens, raw=False):
    """How a column of a table read with pandas.read_csv and written back with to_csv(na_rep='NA') prints: an all-integer column
    as integers, a numeric column with decimals or gaps as floats (shortest round-trip form), anything else as it came;
    missing values as NA.  raw: the table had a header line, so every column was read as text."""
    miss = [t in NA_TOKENS for t in tokens]
    if raw:
        return ['NA' if m else t for t, m in zip(tokens, miss)]
    present = [t for t, m in zip(tokens, miss) if not m]
    if present and all(_INT.match(t) for t in present):
        if not any(miss):
            return [str(int(t)) for t in tokens]
        return ['NA' if m else repr(float(int(t))) for t, m in zip(tokens, miss)]
    try:
        vals = ['NA' if m else _float_text(t) for t, m in zip(tokens, miss)]
        if present:
            return vals
    except ValueError:
        pass
    return ['NA' if m else t for t, m in zip(tokens, miss)]


class BedTable:
    def __init__(self, chrom, start, end, extra, raw):
        self.chr, self.start, self.end, self.extra, self.raw = chrom, start, end, extra, raw

    def __len__(self):
        return len(self.chr)


def load_bed(bed_path):
    """convert.py:77-89: the table of a BED file ('-' / a file object: standard input); a first line whose 2nd and 3rd fields are
    not numbers is a header and is ignored (with the reference's note)."""
    if hasattr(bed_path, 'read'):
        lines = bed_path.read().splitlines()
    else:
        import gzip
        opener = gzip.open if str(bed_path).endswith('.gz') else open
        with opener(bed_path, 'rt') as f:
            lines = f.read().splitlines()
    rows = []
    for line in lines:
        line = line.split('#', 1)[0]                               # pandas comment='#': the rest of the line is not parsed
        if line.strip():
            rows.append(line.split('\t'))
    if not rows:
        eprint('[wt convert] ERROR: empty bed file')
        raise IllegalArgumentError('Invalid bed file')
    width = len(rows[0])
    for i, r in enumerate(rows):
        if len(r) > width:
            raise IllegalArgumentError(f'Invalid input file.\nError tokenizing data. Expected {width} fields in line {i + 1}, saw {len(r)}')
        if len(r) < width:
            r.extend([''] * (width - len(r)))
    if width < 3:
        raise IllegalArgumentError('Invalid bed file')
    raw = False
    if not (rows[0][1].strip().isdigit() and rows[0][2].strip().isdigit()):
        eprint('[wt convert] Header line detected. Ignoring first line of input')
        rows = rows[1:]
        raw = True
    try:
        start = np.array([int(r[1]) for r in rows], dtype=np.int64)
        end = np.array([int(r[2]) for r in rows], dtype=np.int64)
    except ValueError:
        raise IllegalArgumentError('Invalid bed file')
    chrom = [r[0] for r in rows]
    extra = [[r[c] for r in rows] for c in range(3, width)]
    return BedTable(chrom, start, end, extra, raw)


class LociEngine:
    """the genome's loci resident on one GPU"""

    def __init__(self, genome, device=0):
        from . import _lib                     # raises NativeLibraryError if libwgbsseg.so is not built
        self._seg = _lib.Segmenter(device)
        self._seg.set_loci(genome.loci())

    def convert_regions(self, *a):
        return self._seg.convert_regions(*a)

    def kernel_ms(self):
        return self._seg.last_block_sums_ms()

    def close(self):
        self._seg.close()


def regions_to_cpgs(table, genome, engine=None, device=0):
    """(startCpG, endCpG) int64 arrays for the rows of a BedTable, 0 = NA: the reference's add_cpgs_to_bed (convert.py:188-219)
    with its per-chromosome choice of rules; the searches themselves run on the device."""
    names, sizes = genome.get_chrom_cpg_sizes()
    cum = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    index = {c: i for i, c in enumerate(names)}
    if getattr(table, 'chrom_idx', None) is not None:          # (the library's parser has looked the names up already)
        ci = np.asarray(table.chrom_idx, dtype=np.int64)
        present = sorted(names[i] for i in np.unique(ci[ci >= 0]).tolist())
    else:
        ci = np.array([index.get(c, -1) for c in table.chr], dtype=np.int64)
        present = sorted(set(table.chr) & set(names))
    known = ci >= 0
    slow = np.zeros(len(table), dtype=np.uint8)
    for c in present:                                          # the reference walks the chromosomes in sorted order (its warnings too)
        rows = np.flatnonzero(ci == index[c])
        st, en = table.start[rows], table.end[rows]
        o = np.lexsort((en, st))                               # by start, then end; duplicates do not count as overlaps
        st, en = st[o], en[o]
        first = np.ones(st.size, dtype=bool)
        first[1:] = (st[1:] != st[:-1]) | (en[1:] != en[:-1])
        st, en = st[first], en[first]
        if (st[1:] - en[:-1] < 0).any():
            if st.size > 30:
                eprint(f'[wt convert] [{c}] WARNING: Found overlaps in the input bed file. Conversion may be slow.\n'
                       '             Install bedtools for better performance')
            slow[rows] = 1
    cidx = np.where(known, ci, 0)
    clo = np.where(known, cum[cidx], 0)
    chi = np.where(known, cum[cidx + 1], 0)
    bp = np.array([genome.get_chrom_size(c) for c in names], dtype=np.int64)
    cbp = np.where(known, bp[cidx], 0)
    eng = engine or LociEngine(genome, device)
    try:
        return eng.convert_regions(clo, chi, cbp, table.start, table.end, slow)
    finally:
        if engine is None:
            eng.close()


def add_cpgs_to_bed(bed_file, genome, drop_empty, threads=1, add_anno=False, engine=None, device=0):
    """-> the output lines of `convert -L` (chr, start, end, startCpG, endCpG, the other columns)."""
    table = load_bed(bed_file)
    g = genome if isinstance(genome, GenomeRefPaths) else GenomeRefPaths(genome)
    s, e = regions_to_cpgs(table, g, engine, device)
    cols = [column_text(table.chr, table.raw), [str(v) for v in table.start.tolist()], [str(v) for v in table.end.tolist()],
            ['NA' if v == 0 else str(v) for v in s.tolist()], ['NA' if v == 0 else str(v) for v in e.tolist()]]
    cols += [column_text(x, table.raw) for x in table.extra]
    keep = (s != 0) if drop_empty else np.ones(len(table), dtype=bool)
    return ['\t'.join(c[i] for c in cols) for i in np.flatnonzero(keep).tolist()]


class FastBedTable:
    """A BED table the library's parser has read (include/wgbsseg.h: wgbsseg_bed_parse): start, end and the chromosome index of
    every row as arrays, the text left where it is — the annotated rows are written from it by wgbsseg_bed_write_annotated."""

    def __init__(self, parsed):
        self.parsed, self.start, self.end, self.chrom_idx = parsed, parsed.start, parsed.end, parsed.chrom_idx

    def __len__(self):
        return len(self.parsed)


def load_bed_fast(bed_file, genome):
    """-> FastBedTable, or None when the table needs the line-by-line parser (see wgbsseg_bed_parse for what qualifies), the
    library is not built, or WGBSSEG_PY_TABLES=1 asks for the Python path (A/B tests)."""
    if os.environ.get('WGBSSEG_PY_TABLES', '0') not in ('', '0'):
        return None
    try:
        from . import _lib
        _lib.load()
    except Exception:
        return None
    if hasattr(bed_file, 'read'):
        raw = getattr(bed_file, 'buffer', None)
        if raw is None:
            return None                                          # a text stream without its bytes (tests): the Python parser
        data = raw.read()
    elif str(bed_file).endswith('.gz'):
        import gzip
        with gzip.open(bed_file, 'rb') as f:
            data = f.read()
    else:
        with open(bed_file, 'rb') as f:
            data = f.read()
    names, _ = genome.get_chrom_cpg_sizes()
    p = _lib.bed_parse(data, list(names))
    if p is None:
        return None if not hasattr(bed_file, 'read') else data   # (standard input cannot be read twice: hand the bytes back)
    if p.header:
        eprint('[wt convert] Header line detected. Ignoring first line of input')
    return FastBedTable(p)


def annotate_bed_fast(bed_file, genome, drop_empty, out_path, engine=None, device=0):
    """`convert -L` without a Python object per row: the library parses the table, the device joins, the library prints (same
    bytes as add_cpgs_to_bed's lines: tests/test_convert_cpu.py).  -> True when done; False (or the bytes read from a stream)
    when the table needs the line-by-line path."""
    fast = load_bed_fast(bed_file, genome)
    if not isinstance(fast, FastBedTable):
        return fast if isinstance(fast, (bytes, bytearray)) else False
    from . import _lib
    s, e = regions_to_cpgs(fast, genome, engine, device)
    keep = (s != 0) if drop_empty else None
    to_stdout = out_path is None or out_path is sys.stdout
    if to_stdout:
        sys.stdout.flush()
    _lib.bed_write_annotated(None if to_stdout else out_path, fast.parsed, s, e, keep)
    return True


def convert_bed_file(args):
    """convert.py:44-74"""
    out_path = sys.stdout if args.out_path is None else args.out_path
    if not delete_or_skip(out_path, args.force):
        return
    bed_file = sys.stdin if args.bed_file == '-' else args.bed_file
    g = args.genome if isinstance(args.genome, GenomeRefPaths) else GenomeRefPaths(args.genome)
    done = annotate_bed_fast(bed_file, g, args.drop_empty, out_path, device=args.device)
    if done is True:
        return
    if isinstance(done, (bytes, bytearray)):                     # standard input, already consumed: its bytes for the parser below
        import io
        bed_file = io.StringIO(done.decode('utf-8'))
    lines = add_cpgs_to_bed(bed_file, args.genome, args.drop_empty, args.threads, device=args.device)
    text = '\n'.join(lines) + ('\n' if lines else '')
    if out_path is sys.stdout:
        sys.stdout.write(text)
    else:
        with open(out_path, 'w') as f:
            f.write(text)


def load_site_file(site_file):
    """one or two whitespace-separated integer columns: startCpG [endCpG] (a single site s stands for [s, s+1): add_loci.cpp:34-36)"""
    f = sys.stdin if site_file == '-' else open(site_file)
    try:
        s, e = [], []
        for line in f:
            tok = line.split()
            if not tok:
                continue
            s.append(int(tok[0]))
            e.append(int(tok[1]) if len(tok) > 1 else int(tok[0]) + 1)
    except ValueError:
        raise IllegalArgumentError(f'Invalid site file: {site_file}')
    finally:
        if f is not sys.stdin:
            f.close()
    return np.array(s, dtype=np.int64), np.array(e, dtype=np.int64)


def convert_site_file(args):
    """convert.py:228-248"""
    out_path = sys.stdout if args.out_path is None else args.out_path
    if not delete_or_skip(out_path, args.force):
        return
    s, e = load_site_file(args.site_file)
    try:
        write_bed(GenomeRefPaths(args.genome), s, e, None if out_path is sys.stdout else out_path)
    except RuntimeError as err:                                 # the reference's add_loci prints this and exits non-zero
        eprint('[ add_loci ] Failed! exception:')
        eprint(err)
        raise


def convert_single_region(args):
    gr = GenomicRegion(args)
    if gr.is_whole():
        print('Whole genome')
    elif args.parsable:
        print(gr.region_str if args.sites else '{}-{}'.format(*gr.sites))
    else:
        print(gr)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description=main.__doc__)
    region_or_sites = add_where_options(parser, bed_file=True)
    parser.add_argument('--no_anno', help='Do not print genomic annotations', action='store_true')
    region_or_sites.add_argument('--site_file',
                                 help='text file with a single CpG indexes column, or <startCpG, endCpG> columns.\n'
                                      'if "-" is passed, the file is read from stdin.')
    parser.add_argument('--out_path', '-o', help='Output path for bed file [stdout]')
    parser.add_argument('-d', '--debug', action='store_true')
    parser.add_argument('-p', '--parsable', action='store_true', help='Output a parsing friendly format')
    parser.add_argument('--drop_empty', action='store_true', help='Drop empty regions (without CpGs)')
    parser.add_argument('-f', '--force', action='store_true', help='Overwrite existing files if existed')
    add_threads_option(parser)
    parser.add_argument('--device', type=int, default=0, help='HIP device index [0]')
    return parser.parse_args(argv)


def main(argv=None):
    """
    Convert genomic region to CpG index range and vise versa
    """
    args = parse_args(argv)
    if args.bed_file:
        convert_bed_file(args)
    elif args.site_file:
        convert_site_file(args)
    else:
        convert_single_region(args)


if __name__ == '__main__':
    main()
