"""`wgbstools beta_to_blocks` on the GPU: collapse beta files to a blocks table.

Drop-in for the reference's src/python/beta_to_blocks.py: same flags, file formats (.bin / .lbeta / .bedGraph), messages
and entry-point names (load_blocks_file, is_block_file_nice, reduce_data, collapse_process, dump, main), written against
the FORMATS rather than the reference's pandas code:

    blocks table   tab-separated text, >= 5 columns chr, start, end, startCpG, endCpG (+ optional anno, gene), optional
                   header line, '#' comment lines, 'NA'/empty CpG fields allowed          (beta_to_blocks.py:52-91)
    nice table     no NA, no empty block, both CpG columns ascending, no duplicate rows, no overlaps   (:24-49)
    reduction      per block the sums of (#meth, #cov) over sites [startCpG-1, endCpG-1); NA rows -> (0, 0)   (:101-126)
    .bin / .lbeta  the sums as uint8 / uint16 pairs; a row whose coverage exceeds the type's maximum M becomes
                   (trunc(meth / cov * M), M)                                             (utils_wgbs.py:277-290)
    .bedGraph      chr, start, end, beta = meth/cov of the TRIMMED row (%.2f, -1 when cov == 0), coverage   (:160-165)

The reduction runs in ONE kernel launch over all the given beta files (wgbsseg_block_sums, include/wgbsseg.h) instead of
one numpy pass per file in a process pool; input files may be uint8 `.beta` / `.bin` or uint16 `.lbeta`
(utils_wgbs.py:311-319).  No CPU fallback.
"""
import argparse
import os
import re
import os.path as op
import sys

import numpy as np

from .genome import IllegalArgumentError, eprint
from .cliutil import NA_TOKENS

COORDS_COLS5 = ['chr', 'start', 'end', 'startCpG', 'endCpG']


def b2b_log(*args, **kwargs):
    print('[ wt beta_to_blocks ]', *args, file=sys.stderr, **kwargs)


class _LazyExtra(dict):
    """The extra columns ('anno', 'gene') of a table the library parsed: the names are known, the lists of strings are cut out of
    the file's text the first time somebody asks for one (find_markers only ever prints them for the markers it found)."""

    def __init__(self, parsed, names):
        super().__init__()
        self._parsed, self._names, self._cut = parsed, list(names), False

    def _fill(self):
        if not self._cut:
            for k, col in zip(self._names, self._parsed.fields(None, 5, len(self._names))):
                dict.__setitem__(self, k, col)
            self._cut = True

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)

    def __contains__(self, k):
        return k in self._names

    def keys(self):
        return list(self._names)

    def values(self):
        self._fill()
        return [dict.__getitem__(self, k) for k in self._names]

    def items(self):
        self._fill()
        return [(k, dict.__getitem__(self, k)) for k in self._names]


class BlocksTable:
    """A blocks table in memory, column-wise.  `chr`, `start`, `end`: the file's own text (they are only ever written back);
    `startCpG`, `endCpG`: int64 with `na` marking rows whose CpG fields are missing; `extra`: {'anno': [...], 'gene': [...]}.
    A table read by the library's parser (`parsed`: _lib.ParsedBlocks) keeps the file's bytes and row offsets instead of lists
    of strings: the writers in the library print straight from them; `chr` / `start` / `end` / the extra columns are cut out on
    first use (coords_of / extras_of: for a few rows only)."""

    def __init__(self, chrom, start, end, start_cpg, end_cpg, na, extra=None, parsed=None):
        self.parsed = parsed
        self._coords = None if parsed is not None else (list(chrom), list(start), list(end))
        self.startCpG = np.asarray(start_cpg, dtype=np.int64)
        self.endCpG = np.asarray(end_cpg, dtype=np.int64)
        self.na = np.asarray(na, dtype=bool)
        self.extra = extra if isinstance(extra, _LazyExtra) else dict(extra or {})

    def _text_columns(self):
        if self._coords is None:
            self._coords = self.parsed.coords()
        return self._coords

    chr = property(lambda self: self._text_columns()[0])
    start = property(lambda self: self._text_columns()[1])
    end = property(lambda self: self._text_columns()[2])

    def coords_of(self, idx):
        """[(chr, start, end) text of row i for i in idx] without cutting out the whole table"""
        idx = np.asarray(idx, dtype=np.int64)
        if self._coords is None:
            c, s, e = self.parsed.coords(idx)
            return list(zip(c, s, e))
        c, s, e = self._coords
        return [(c[i], s[i], e[i]) for i in idx.tolist()]

    def extras_of(self, idx):
        """{name: [text of row i for i in idx]} for the extra columns"""
        idx = np.asarray(idx, dtype=np.int64)
        if isinstance(self.extra, _LazyExtra) and not self.extra._cut:
            names = list(self.extra)
            return dict(zip(names, self.parsed.fields(idx, 5, len(names))))
        return {k: [v[i] for i in idx.tolist()] for k, v in self.extra.items()}

    @property
    def columns(self):
        return COORDS_COLS5 + list(self.extra)

    @property
    def shape(self):
        return (len(self), len(self.columns))

    def __len__(self):
        return int(self.startCpG.size)

    def take(self, idx):
        """rows idx (a slice or an index array) as a new table"""
        if self.parsed is not None:
            p = self.parsed.take(idx)
            return BlocksTable(None, None, None, self.startCpG[idx], self.endCpG[idx], self.na[idx],
                               _LazyExtra(p, list(self.extra)) if len(self.extra) else None, parsed=p)
        rows = range(*idx.indices(len(self))) if isinstance(idx, slice) else np.asarray(idx).tolist()
        c, s, e = self._coords
        return BlocksTable([c[i] for i in rows], [s[i] for i in rows], [e[i] for i in rows], self.startCpG[idx], self.endCpG[idx],
                           self.na[idx], {k: [v[i] for i in rows] for k, v in self.extra.items()})

    def rows(self, a, b):
        """rows [a, b) as a new table (beta_to_table walks the table in chunks)"""
        return self.take(slice(a, b))

    def copy(self):
        return self.rows(0, len(self))

    def cpg_text(self):
        """the two CpG columns as they print: integers, NA where missing"""
        s = ['NA' if n else str(v) for v, n in zip(self.startCpG.tolist(), self.na.tolist())]
        e = ['NA' if n else str(v) for v, n in zip(self.endCpG.tolist(), self.na.tolist())]
        return s, e


def _opener(path):
    if path.endswith('.gz'):
        import gzip
        return gzip.open(path, 'rt')
    return open(path, 'r')


_MIDLINE_COMMENT = re.compile(rb'[^\n]#')


def _load_blocks_native(blocks_path, nrows, anno=False):
    """The library's one-pass parser (include/wgbsseg.h: wgbsseg_blocks_parse) on the file's bytes -> BlocksTable, or None when
    the library is not built or the file is not a plain table (the line-by-line parser below then handles it and owns the
    messages).  WGBSSEG_PY_TABLES=1 turns it off (A/B tests)."""
    if os.environ.get('WGBSSEG_PY_TABLES', '0') not in ('', '0'):
        return None
    try:
        from . import _lib
        _lib.load()
    except Exception:
        return None
    if blocks_path.endswith('.gz'):
        import gzip
        with gzip.open(blocks_path, 'rb') as f:
            data = f.read()
    else:
        with open(blocks_path, 'rb') as f:
            data = f.read()
    if _MIDLINE_COMMENT.search(data):          # pandas comment='#' cuts a line at a '#' anywhere: the line-by-line parser does that
        return None
    p = _lib.blocks_parse(data, nrows)
    if p is None:
        return None
    extra = _LazyExtra(p, ['anno', 'gene']) if (anno and p.first_fields >= 7) else None
    return BlocksTable(None, None, None, p.start_cpg, p.end_cpg, p.na, extra, parsed=p)


def load_blocks_file(blocks_path, anno=False, nrows=None):
    """Parse a blocks table (format above).  Fewer than 5 columns, or endCpG < startCpG in a complete row, are errors;
    a file without any row gives an empty table (after the reference's 'Empty blocks file.' note)."""
    if not op.isfile(blocks_path):
        raise IllegalArgumentError(f'Invalid file: {blocks_path}')
    t = _load_blocks_native(blocks_path, nrows, anno)
    if t is not None:
        ok = ~t.na
        if (t.endCpG[ok] < t.startCpG[ok]).any():
            raise IllegalArgumentError(f'Invalid CpG columns in blocks file {blocks_path}')
        return t
    want = 7 if anno else 5
    chrom, start, end, scpg, ecpg, na = [], [], [], [], [], []
    extra = None
    first = True
    with _opener(blocks_path) as f:
        for line in f:
            line = line.split('#', 1)[0]                         # pd.read_csv(comment='#'): the rest of the line is not parsed
            if not line.strip():
                continue
            tok = line.rstrip('\n').rstrip('\r').split('\t')
            if first:
                first = False
                if len(tok) < 5:
                    msg = f'Invalid blocks file: {blocks_path}. less than {want} columns.\n'
                    msg += f'Run wgbstools convert -L {blocks_path} -o OUTPUT_REGION_FILE to add the CpG columns'
                    raise IllegalArgumentError(msg)
                if anno and len(tok) >= 7:
                    extra = {'anno': [], 'gene': []}
                if not tok[1].isdigit():                         # a header line: the second field of a data row is a position
                    continue
            if len(tok) < 5:
                eprint(f'Invalid input file.\nrow with {len(tok)} fields: {line.strip()[:80]}')
                return BlocksTable([], [], [], [], [], [])
            chrom.append(tok[0]); start.append(tok[1]); end.append(tok[2])
            miss = tok[3] in NA_TOKENS or tok[4] in NA_TOKENS
            na.append(miss)
            try:
                scpg.append(0 if miss else int(float(tok[3])))
                ecpg.append(0 if miss else int(float(tok[4])))
            except (ValueError, OverflowError):
                raise IllegalArgumentError(f'Invalid CpG columns in blocks file {blocks_path}: {tok[3]!r}, {tok[4]!r}')
            if extra is not None:
                extra['anno'].append(tok[5] if len(tok) > 5 else '')
                extra['gene'].append(tok[6] if len(tok) > 6 else '')
            if nrows is not None and len(chrom) >= nrows:
                break
    if not chrom:
        eprint('Empty blocks file.\nNo columns to parse from file')
        return BlocksTable([], [], [], [], [], [])
    t = BlocksTable(chrom, start, end, scpg, ecpg, na, extra)
    ok = ~t.na
    if (t.endCpG[ok] < t.startCpG[ok]).any():
        raise IllegalArgumentError(f'Invalid CpG columns in blocks file {blocks_path}')
    return t


def is_block_file_nice(t):
    """(True, '') for a nice table (definition above), else (False, reason): the first failing rule in the reference's
    order of checks (beta_to_blocks.py:24-49)."""
    n = len(t)
    s, e = t.startCpG, t.endCpG
    if t.na.any():
        return False, 'Some blocks are empty (NA)'
    if (e <= s).any():
        return False, 'Some blocks are empty (startCpG==endCpG)'
    if n > 1 and (s[1:] < s[:-1]).any():
        return False, 'startCpG is not monotonically increasing'
    if n > 1 and (e[1:] < e[:-1]).any():
        return False, 'endCpG is not monotonically increasing'
    # (both CpG columns are in order here: rows that repeat each other have equal CpG columns and stand next to each other)
    same = np.flatnonzero((s[1:] == s[:-1]) & (e[1:] == e[:-1])) if n > 1 else np.zeros(0, dtype=np.int64)
    if same.size:
        idx = np.unique(np.concatenate([same, same + 1]))
        txt = t.coords_of(idx)
        ext = [t.extra[k] for k in t.extra]
        rows = [txt[j] + (int(s[i]), int(e[i])) + tuple(x[i] for x in ext) for j, i in enumerate(idx.tolist())]
        if len(set(rows)) != len(rows):
            return False, 'Some blocks are duplicated'
    if n > 1 and (s[1:] < e[:-1]).any():
        return False, 'Some blocks overlap'
    return True, ''


def block_site_ranges(t):
    """0-based half-open site ranges of the rows: [startCpG-1, endCpG-1); NA rows -> the empty range (sums 0, 0)."""
    s0 = np.where(t.na, 0, t.startCpG - 1).astype(np.int64)
    e0 = np.where(t.na, 0, t.endCpG - 1).astype(np.int64)
    return s0, e0


class BlockSumEngine:
    """The given beta files resident on one GPU: uint8 `.beta` / `.bin`, or uint16 `.lbeta` (not mixed)."""

    def __init__(self, beta_paths, device=0):
        from . import _lib                     # raises NativeLibraryError if libwgbsseg.so is not built
        suffs = {op.splitext(b)[1] for b in beta_paths}
        for b in beta_paths:
            if not (op.isfile(b) and op.splitext(b)[1] in ('.beta', '.lbeta', '.bin')):
                raise IllegalArgumentError(f'Invalid beta file:\n{b}')
        wide = '.lbeta' in suffs
        if wide and len(suffs) > 1:
            raise IllegalArgumentError('uint16 (.lbeta) and uint8 (.beta / .bin) files cannot be reduced together')
        self._seg = _lib.Segmenter(device)
        maps = [np.memmap(b, dtype=np.uint16 if wide else np.uint8, mode='r') for b in beta_paths]
        if len({m.size for m in maps}) != 1:
            raise IllegalArgumentError('beta files of different sizes')
        self.nr_sites = maps[0].size // 2
        (self._seg.set_lbetas if wide else self._seg.set_betas)(maps)

    def reduce(self, t, mode=0, min_cov=1):
        s0, e0 = block_site_ranges(t)
        if s0.size and (e0.max() > self.nr_sites):
            raise IllegalArgumentError('blocks table reaches beyond the beta file')
        return self._seg.block_sums(s0, e0, mode=mode, min_cov=min_cov)

    def marker_stats(self, tg, bg, n_blocks):
        """per-block statistics of two sets of resident samples over the table the last mode-3 reduce() left on the device
        (find_markers; include/wgbsseg.h wgbsseg_marker_stats)"""
        return self._seg.marker_stats(tg, bg, n_blocks)

    def kernel_ms(self):
        return self._seg.last_block_sums_ms()

    def close(self):
        self._seg.close()


def reduce_data(beta_path, df, is_nice=None, engine=None):
    """int64 array [n_blocks, 2] of the (#meth, #cov) sums of one beta file (beta_to_blocks.py:119-126)."""
    eng = engine or BlockSumEngine([beta_path])
    try:
        return eng.reduce(df, mode=0)[0].astype(np.int64)
    finally:
        if engine is None:
            eng.close()


def trim_to_uint8(data, lbeta=False):
    """Host form of the .bin / .lbeta row rule (format above) for callers that hold the sums; the device applies the same
    rule in modes 1 / 2 of wgbsseg_block_sums."""
    top = 65535 if lbeta else 255
    t = np.array(data, dtype=np.int64)
    over = t[:, 1] > top
    t[over, 0] = (t[over, 0] / t[over, 1] * top).astype(np.int64)
    t[over, 1] = top
    return t.astype(np.uint16 if lbeta else np.uint8)


def write_bedgraph(path, t, bin_table):
    """chr, start, end, beta (%.2f; -1 for 0/0), coverage — from the trimmed rows, like the reference's in-place trim."""
    if t.parsed is not None and bin_table.dtype in (np.uint8, np.uint16) and len(t):
        from . import _lib
        _lib.blocks_write_bedgraph(path, t.parsed, bin_table)
        return
    meth = bin_table[:, 0].astype(np.int64)
    cov = bin_table[:, 1].astype(np.int64)
    with open(path, 'w') as f:
        for c, s, e, m, v in zip(t.chr, t.start, t.end, meth.tolist(), cov.tolist()):
            beta = '%.2f' % (m / v) if v else '-1'               # 0/0 is the table's missing value: -1
            f.write(f'{c}\t{s}\t{e}\t{beta}\t{v}\n')


def dump(df, bin_table, beta_path, lbeta, out_dir, bedGraph):
    """Write <out_dir>/<beta name>.bin|.lbeta (and .bedGraph) for one beta file; bin_table = its trimmed table."""
    stem = op.join(out_dir, op.splitext(op.basename(beta_path))[0])
    target = stem + ('.lbeta' if lbeta else '.bin')
    np.ascontiguousarray(bin_table).tofile(target)
    b2b_log(target)
    if bedGraph:
        write_bedgraph(stem + '.bedGraph', df, bin_table)


def collapse_process(beta_path, df, is_nice=None, lbeta=False, out_dir=None, bedGraph=False, engine=None):
    """One beta file: the sums (out_dir None) or its output files; failures are logged, not raised (beta_to_blocks.py:129-139)."""
    try:
        if out_dir is None:
            return reduce_data(beta_path, df, is_nice, engine)
        eng = engine or BlockSumEngine([beta_path])
        try:
            table = eng.reduce(df, mode=2 if lbeta else 1)[0]
        finally:
            if engine is None:
                eng.close()
        return dump(df, table, beta_path, lbeta, out_dir, bedGraph)
    except Exception as e:
        b2b_log('Failed with beta', beta_path)
        b2b_log('Exception:', e)


def output_stem(beta, out_dir):
    base = op.basename(beta)
    if base.endswith('.gz'):
        base = base[:-3]
    return op.join(out_dir, op.splitext(base)[0])


def filter_existing_files(files, out_dir, lbeta):
    """the input files whose output does not exist yet; the others are named on stderr"""
    suff = '.lbeta' if lbeta else '.bin'
    todo = [b for b in files if not op.isfile(output_stem(b, out_dir) + suff)]
    for b in files:
        if b not in todo:
            b2b_log(f'Skipping {b}. Use -f flag to overwrite')
    return todo


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description=main.__doc__)
    parser.add_argument('input_files', nargs='+', help='one or more beta files')
    parser.add_argument('-b', '--blocks_file', help='blocks path', required=True)
    parser.add_argument('-o', '--out_dir', help='output directory. Default is "."', default='.')
    parser.add_argument('-l', '--lbeta', action='store_true', help='Use lbeta file (uint16) instead of bin (uint8)')
    parser.add_argument('--bedGraph', action='store_true', help='output a text file in addition to binary file')
    parser.add_argument('--force', '-f', action='store_true', help='Overwrite existing files if existed')
    parser.add_argument('--debug', '-d', action='store_true')
    parser.add_argument('-@', '--threads', type=int, default=1, help='kept for compatibility; the GPU batches the files')
    parser.add_argument('--device', type=int, default=0, help='GPU ordinal')
    return parser.parse_args(argv)


def main(argv=None):
    """
    Collapse beta file to blocks binary file, of the same beta format
    """
    args = parse_args(argv)
    missing = [f for f in args.input_files if not op.isfile(f)]
    if missing:
        raise IllegalArgumentError(f'Invalid file: {missing[0]}')
    if not op.isdir(args.out_dir):
        raise IllegalArgumentError(f'Invalid output dir: {args.out_dir}')
    files = args.input_files if args.force else filter_existing_files(args.input_files, args.out_dir, args.lbeta)
    table = load_blocks_file(args.blocks_file)
    nice, why = is_block_file_nice(table)
    if not nice:
        b2b_log(why)
    if not files:
        return
    eng = BlockSumEngine(files, device=args.device)
    try:
        rows = eng.reduce(table, mode=2 if args.lbeta else 1)
    finally:
        eng.close()
    for beta, tbl in zip(files, rows):
        dump(table, tbl, beta, args.lbeta, args.out_dir, args.bedGraph)


if __name__ == '__main__':
    main()
