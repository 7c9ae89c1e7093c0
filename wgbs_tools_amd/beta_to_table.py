"""`wgbstools beta_to_table` on the GPU: a text table of per-block average methylation per sample or group.

Mirror of the reference's src/python/beta_to_table.py (flags, table layout, NA / rounding rules):
    groups_load_wrap / load_gfile_helper        beta_to_table.py:39-57, dmb.py:24-38
    get_table                                   beta_to_table.py:73-106: block sums -> beta2vec -> nanmean per group
    dump                                        beta_to_table.py:116-128
The block sums and the meth/cov ratios (NaN below --min_cov, utils_wgbs.py:270-274) of ALL beta files come from one
wgbsseg_block_sums call per chunk of blocks (mode 3); grouping and text formatting stay on the host.
"""
import argparse
import os.path as op
import sys
import warnings

import numpy as np
import pandas as pd

from .beta_to_blocks import BlockSumEngine, load_blocks_file
from .genome import IllegalArgumentError, eprint


def drop_dup_keep_order(lst):
    seen = set()
    return [x for x in lst if not (x in seen or seen.add(x))]


def pretty_name(fpath):
    base = op.basename(fpath)
    if base.endswith('.gz'):
        base = base[:-3]
    return op.splitext(base)[0]


def load_gfile_helper(groups_file):
    """dmb.py:24-38"""
    gf = pd.read_csv(groups_file, index_col=False, comment='#')
    if 'group' not in gf.columns:
        raise IllegalArgumentError('gropus file must have a column named "group"')
    if 'include' in gf.columns:
        if gf['include'].dtype != bool:
            eprint('Invalid group file')
            raise IllegalArgumentError('Invalid group file. Include column must be boolean')
        gf = gf[gf['include']]
    gf = gf.rename(columns={gf.columns[0]: 'fname'})
    return gf[['fname', 'group']].dropna().reset_index(drop=True)


def match_prefix_to_bin(prefixes, bins, suff):
    """dmb.py:41-79 (exact-name form): the path of `prefix + suff` among `bins`, for every prefix."""
    full_paths, missing = [], []
    for prefix in prefixes:
        results = [f for f in bins if op.basename(f) == prefix + suff]
        if not results:
            missing.append(prefix)
        else:
            full_paths.append(results[0])
    if missing:
        eprint(f'Error: {len(missing)} prefixes from groups file were not found in input bins:')
        for p in missing:
            eprint(p)
        raise IllegalArgumentError('groups file mismatch binary files')
    return full_paths


def groups_load_wrap(groups_file, betas):
    if groups_file is not None:
        if not op.isfile(groups_file):
            raise IllegalArgumentError(f'Invalid file: {groups_file}')
        gf = load_gfile_helper(groups_file)
    else:
        betas = drop_dup_keep_order(list(betas))
        fnames = [pretty_name(b) for b in betas]
        gf = pd.DataFrame(columns=['fname'], data=fnames)
        gf['group'] = gf['fname']
    suff = '.lbeta' if betas[0].endswith('.lbeta') else '.beta'
    gf['full_path'] = match_prefix_to_bin(gf['fname'], betas, suff)
    return gf


def get_table(blocks_df, gf, min_cov, threads=8, verbose=False, group=True, engine=None):
    """beta_to_table.py:73-106 for one chunk of blocks."""
    if verbose:
        eprint(f'[wt table] reducing to {blocks_df.shape[0]:,} blocks')
    betas = drop_dup_keep_order(gf['full_path'])
    own = engine is None
    eng = BlockSumEngine(betas) if own else engine
    try:
        vecs = eng.reduce(blocks_df.reset_index(drop=True), mode=3, min_cov=min_cov)          # [n_betas][n_blocks]
    finally:
        if own:
            eng.close()
    dres = {pretty_name(b): vecs[i] for i, b in enumerate(betas)}
    blocks_df = blocks_df.reset_index(drop=True)
    if not group:
        return pd.concat([blocks_df, pd.DataFrame(dres)[gf['fname'].tolist()]], axis=1)
    ugroups = drop_dup_keep_order(gf['group'])
    with warnings.catch_warnings():
        warnings.filterwarnings('ignore', category=RuntimeWarning)
        cols = {}
        for ugroup in ugroups:
            members = gf['fname'][gf['group'] == ugroup]
            cols[ugroup] = np.nanmean(np.concatenate([dres[k][None, :] for k in members]), axis=0).T
    return pd.concat([blocks_df, pd.DataFrame(cols, index=blocks_df.index)[ugroups]], axis=1)


def betas2table(betas, blocks, groups_file, min_cov, threads=8, verbose=False):
    gf = groups_load_wrap(groups_file, betas)
    return get_table(load_blocks_file(blocks), gf, min_cov, threads, verbose)


def dump(outpath, df, first=True, digits=3):
    """beta_to_table.py:116-128"""
    if outpath is None:
        outpath = sys.stdout
    df.to_csv(outpath, na_rep='NA', float_format=f'%.{digits}f', index=None, sep='\t',
              mode='w' if first else 'a', header=True if first else None)


def beta2table_generator(betas, blocks, groups_file, min_cov, threads, chunk_size=None, verbose=False, device=0):
    if not op.isfile(blocks):
        raise IllegalArgumentError(f'Invalid file: {blocks}')
    gf = groups_load_wrap(groups_file, betas)
    blocks_df = load_blocks_file(blocks)
    if chunk_size is None:
        chunk_size = blocks_df.shape[0]
    eng = BlockSumEngine(drop_dup_keep_order(gf['full_path']), device=device)    # the files go to the device once
    try:
        for start in range(0, blocks_df.shape[0], chunk_size):
            yield get_table(blocks_df.iloc[start:start + chunk_size].copy(), gf, min_cov, threads, verbose, engine=eng)
    finally:
        eng.close()


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('blocks', help='Blocks file with no header and with >= 5 columns')
    parser.add_argument('--output', '-o', help='specify output path for the table [Default is stdout]')
    parser.add_argument('--groups_file', '-g', help='groups csv file with at least 2 columns: name, group. beta files belong to the same group are averaged')
    parser.add_argument('--betas', nargs='+', help='beta files', required=True)
    parser.add_argument('--verbose', '-v', action='store_true')
    parser.add_argument('-c', '--min_cov', type=int, default=4, help='Minimal coverage to be considered. blocks with less than MIN_COV site observations are considered as missing. [4]')
    parser.add_argument('--digits', type=int, default=2, help='float percision (number of digits) [2]')
    parser.add_argument('--chunk_size', type=int, default=200000, help='Number of blocks to load on each step [200000]')
    parser.add_argument('-@', '--threads', type=int, default=1, help='kept for compatibility; the GPU batches the files')
    parser.add_argument('--device', type=int, default=0, help='GPU ordinal')
    return parser.parse_args(argv)


def main(argv=None):
    """
    build a text table from beta files
    Optionally collapse samples with groups file
    """
    args = parse_args(argv)
    first_chunk = True
    for chunk in beta2table_generator(args.betas, args.blocks, args.groups_file, args.min_cov, args.threads,
                                      args.chunk_size, args.verbose, args.device):
        dump(args.output, chunk, first_chunk, args.digits)
        first_chunk = False


if __name__ == '__main__':
    main()
