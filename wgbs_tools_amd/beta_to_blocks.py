"""`wgbstools beta_to_blocks` on the GPU: collapse beta files to a blocks table.

Mirror of the reference's src/python/beta_to_blocks.py — same entry points, flags, file formats and messages:
    load_blocks_file / is_block_file_nice      beta_to_blocks.py:24-91
    reduce_data / collapse_process             beta_to_blocks.py:101-139   -> wgbsseg_block_sums (include/wgbsseg.h)
    dump (.bin / .lbeta / .bedGraph)           beta_to_blocks.py:149-165, utils_wgbs.py:277-290 trim_to_uint8
The reduction itself — per block, the sums of (#meth, #cov) over its CpGs, for every sample — runs in one kernel launch
over all the given beta files at once instead of one numpy pass per file in a process pool.  No CPU fallback.
"""
import argparse
import os.path as op
import sys

import numpy as np
import pandas as pd

from .genome import IllegalArgumentError, eprint

COORDS_COLS5 = ['chr', 'start', 'end', 'startCpG', 'endCpG']


def b2b_log(*args, **kwargs):
    print('[ wt beta_to_blocks ]', *args, file=sys.stderr, **kwargs)


def is_block_file_nice(df):
    """beta_to_blocks.py:24-49: (True, '') when the table has no NAs or empty blocks, is sorted, has no duplicates and no
    overlaps; else (False, the first reason in the reference's order)."""
    msg = ''
    if df[['startCpG', 'endCpG']].isna().values.sum() > 0:
        msg = 'Some blocks are empty (NA)'
    elif not (df['endCpG'] - df['startCpG'] > 0).all():
        msg = 'Some blocks are empty (startCpG==endCpG)'
    elif not np.all(np.diff(df['startCpG'].values) >= 0):
        msg = 'startCpG is not monotonically increasing'
    elif not np.all(np.diff(df['endCpG'].values) >= 0):
        msg = 'endCpG is not monotonically increasing'
    elif df.shape[0] != df.drop_duplicates().shape[0]:
        msg = 'Some blocks are duplicated'
    elif not (df['startCpG'][1:].values - df['endCpG'][:df.shape[0] - 1].values >= 0).all():
        msg = 'Some blocks overlap'
    return (False, msg) if msg else (True, '')


def load_blocks_file(blocks_path, anno=False, nrows=None):
    """beta_to_blocks.py:52-91: 5-column (optionally 7 with anno, gene) tab-separated table, header line optional,
    '#' comments skipped, NA allowed in the CpG columns; empty DataFrame on parser errors."""
    if not op.isfile(blocks_path):
        raise IllegalArgumentError(f'Invalid file: {blocks_path}')
    try:
        peek_df = pd.read_csv(blocks_path, sep='\t', nrows=1, header=None, comment='#')
        header = None if str(peek_df.iloc[0, 1]).isdigit() else 0
        names = COORDS_COLS5.copy()
        if anno:
            names += ['anno', 'gene']
        if len(peek_df.columns) < len(COORDS_COLS5):
            msg = f'Invalid blocks file: {blocks_path}. less than {len(names)} columns.\n'
            msg += f'Run wgbstools convert -L {blocks_path} -o OUTPUT_REGION_FILE to add the CpG columns'
            raise IllegalArgumentError(msg)
        elif len(peek_df.columns) < len(names):
            names = COORDS_COLS5
        dtypes = {'startCpG': 'Int64', 'endCpG': 'Int64'}
        df = pd.read_csv(blocks_path, sep='\t', usecols=range(len(names)), dtype=dtypes, header=header, names=names,
                         nrows=nrows, comment='#')
        dfnona = df.dropna()
        if not ((dfnona['endCpG'] - dfnona['startCpG']) >= 0).all():
            raise IllegalArgumentError(f'Invalid CpG columns in blocks file {blocks_path}')
        if dfnona.shape[0] == df.shape[0]:
            df['startCpG'] = df['startCpG'].astype(int)
            df['endCpG'] = df['endCpG'].astype(int)
    except pd.errors.ParserError as e:
        eprint(f'Invalid input file.\n{e}')
        return pd.DataFrame()
    except pd.errors.EmptyDataError as e:
        eprint(f'Empty blocks file.\n{e}')
        return pd.DataFrame()
    return df


def block_site_ranges(df):
    """0-based half-open site ranges of the table's rows: [startCpG-1, endCpG-1); NA rows -> empty range (sum 0, 0:
    what the reference's slow_method writes for them, beta_to_blocks.py:112-114)."""
    s = pd.to_numeric(df['startCpG'], errors='coerce').astype('float64').values
    e = pd.to_numeric(df['endCpG'], errors='coerce').astype('float64').values
    na = np.isnan(s) | np.isnan(e)
    s0 = np.where(na, 0, s - 1).astype(np.int64)
    e0 = np.where(na, 0, e - 1).astype(np.int64)
    return s0, e0


class BlockSumEngine:
    """The given beta files resident on one GPU (uint8 .beta / .bin rows; .lbeta is not read by this library)."""

    def __init__(self, beta_paths, device=0):
        from . import _lib                     # raises NativeLibraryError if libwgbsseg.so is not built
        for b in beta_paths:
            if not (op.isfile(b) and op.splitext(b)[1] in ('.beta', '.bin')):
                raise IllegalArgumentError(f'Invalid beta file:\n{b}')
        self._seg = _lib.Segmenter(device)
        maps = [np.memmap(b, dtype=np.uint8, mode='r') for b in beta_paths]
        if len({m.size for m in maps}) != 1:
            raise IllegalArgumentError('beta files of different sizes')
        self.nr_sites = maps[0].size // 2
        self._seg.set_betas(maps)

    def reduce(self, df, mode=0, min_cov=1):
        s0, e0 = block_site_ranges(df)
        if s0.size and (e0.max() > self.nr_sites):
            raise IllegalArgumentError('blocks table reaches beyond the beta file')
        return self._seg.block_sums(s0, e0, mode=mode, min_cov=min_cov)

    def kernel_ms(self):
        return self._seg.last_block_sums_ms()

    def close(self):
        self._seg.close()


def reduce_data(beta_path, df, is_nice=None, engine=None):
    """beta_to_blocks.py:119-126: int array [n_blocks, 2] of (#meth, #cov) sums of one beta file."""
    own = engine is None
    eng = BlockSumEngine([beta_path]) if own else engine
    try:
        return eng.reduce(df.reset_index(drop=True), mode=0)[0].astype(np.int64)
    finally:
        if own:
            eng.close()


def trim_to_uint8(data, lbeta=False):
    """utils_wgbs.py:277-290 (host form, for callers that hold the sums): rows with cov > max become
    (trunc(meth / cov * max), max).  The device applies the same rule in modes 1 / 2 of wgbsseg_block_sums."""
    max_val = 65535 if lbeta else 255
    data = np.array(data, dtype=np.int64)
    big = data[:, 1] > max_val
    data[big, 0] = (data[big, 0] / data[big, 1] * max_val).astype(np.int64)
    data[big, 1] = max_val
    return data.astype(np.uint16 if lbeta else np.uint8)


def dump(df, bin_table, beta_path, lbeta, out_dir, bedGraph):
    """beta_to_blocks.py:149-165; bin_table = the trimmed table of this beta (what the reference's in-place
    trim_to_uint8 leaves in reduced_data before the bedGraph is computed from it)."""
    name = op.splitext(op.basename(beta_path))[0]
    suff = '.lbeta' if lbeta else '.bin'
    prefix = op.join(out_dir, name)
    bin_table.tofile(prefix + suff)
    b2b_log(prefix + suff)
    if bedGraph:
        df = df.copy()
        with np.errstate(divide='ignore', invalid='ignore'):
            df['beta'] = bin_table[:, 0].astype(np.int64) / bin_table[:, 1].astype(np.int64)
        df['coverage'] = bin_table[:, 1].astype(np.int64)
        df[['chr', 'start', 'end', 'beta', 'coverage']].to_csv(prefix + '.bedGraph', sep='\t', index=None, header=None,
                                                             na_rep=-1, float_format='%.2f')


def collapse_process(beta_path, df, is_nice=None, lbeta=False, out_dir=None, bedGraph=False, engine=None):
    """beta_to_blocks.py:129-139 for one beta file."""
    try:
        if out_dir is None:
            return reduce_data(beta_path, df, is_nice, engine)
        own = engine is None
        eng = BlockSumEngine([beta_path]) if own else engine
        try:
            table = eng.reduce(df.reset_index(drop=True), mode=2 if lbeta else 1)[0]
        finally:
            if own:
                eng.close()
        return dump(df, table, beta_path, lbeta, out_dir, bedGraph)
    except Exception as e:
        b2b_log('Failed with beta', beta_path)
        b2b_log('Exception:', e)


def filter_existing_files(files, out_dir, lbeta):
    files_to_process = []
    suff = '.lbeta' if lbeta else '.bin'
    for beta in files:
        base = op.basename(beta)
        stem = base[:-3] if base.endswith('.gz') else base
        prefix = op.join(out_dir, op.splitext(stem)[0])
        if not op.isfile(prefix + suff):
            files_to_process.append(beta)
        else:
            b2b_log(f'Skipping {beta}. Use -f flag to overwrite')
    return files_to_process


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
    files = args.input_files
    for f in files:
        if not op.isfile(f):
            raise IllegalArgumentError(f'Invalid file: {f}')
    if not op.isdir(args.out_dir):
        raise IllegalArgumentError(f'Invalid output dir: {args.out_dir}')
    if not args.force:
        files = filter_existing_files(files, args.out_dir, args.lbeta)
    df = load_blocks_file(args.blocks_file)
    is_nice, msg = is_block_file_nice(df)
    if not is_nice:
        b2b_log(msg)
    if not files:
        return
    eng = BlockSumEngine(files, device=args.device)
    try:
        tables = eng.reduce(df.reset_index(drop=True), mode=2 if args.lbeta else 1)
    finally:
        eng.close()
    for beta, table in zip(files, tables):
        dump(df, table, beta, args.lbeta, args.out_dir, args.bedGraph)


if __name__ == '__main__':
    main()
