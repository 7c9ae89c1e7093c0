"""`wgbstools beta_to_table` on the GPU: a text table of per-block average methylation per sample or per group.

Drop-in for the reference's src/python/beta_to_table.py (flags, table layout, NA and rounding rules), written against the
formats:

    groups file    csv with a header; first column = file name prefix, a column named `group`, optional boolean column
                   `include`; '#' comment lines; rows with an empty name or group are ignored            (dmb.py:24-38)
    table          the blocks table's columns, then one column per sample (no groups file) or per group: the mean over the
                   group's samples of meth/cov per block, samples with coverage < --min_cov counting as missing; missing
                   values print as NA, floats with --digits decimals; header line once                  (beta_to_table.py:73-128)

The block sums and the meth/cov ratios of ALL beta files come from one wgbsseg_block_sums call per chunk of blocks
(mode 3: ratios as float64, NaN below min_cov — utils_wgbs.py:270-274); grouping and text formatting stay on the host.
"""
import argparse
import csv
import os.path as op
import sys
import warnings

import numpy as np

from .beta_to_blocks import BlockSumEngine, load_blocks_file
from .genome import IllegalArgumentError, eprint


def drop_dup_keep_order(lst):
    return list(dict.fromkeys(lst))


def pretty_name(fpath):
    base = op.basename(fpath)
    if base.endswith('.gz'):
        base = base[:-3]
    return op.splitext(base)[0]


class Groups:
    """fname[i] belongs to group[i] and lives in full_path[i]."""

    def __init__(self, fname, group, full_path=None):
        self.fname, self.group, self.full_path = list(fname), list(group), list(full_path or [])

    def __getitem__(self, key):                                    # gf['fname'] / gf['group'] / gf['full_path']
        return getattr(self, key)


def load_gfile_helper(groups_file):
    """Parse a groups file (format above) -> Groups without paths."""
    with open(groups_file, newline='') as f:
        rows = [r for r in csv.reader(line.split('#', 1)[0] for line in f) if r]      # (pandas comment='#': a '#' anywhere ends the parsed part of a line)
    if not rows:
        raise IllegalArgumentError('gropus file must have a column named "group"')
    header, body = [h.strip() for h in rows[0]], rows[1:]
    if 'group' not in header:
        raise IllegalArgumentError('gropus file must have a column named "group"')
    gi = header.index('group')
    keep = [True] * len(body)
    if 'include' in header:
        ii = header.index('include')
        vals = [(r[ii].strip() if ii < len(r) else '') for r in body]
        if any(v.lower() not in ('true', 'false') for v in vals):
            eprint('Invalid group file')
            raise IllegalArgumentError('Invalid group file. Include column must be boolean')
        keep = [v.lower() == 'true' for v in vals]
    fname, group = [], []
    for r, k in zip(body, keep):
        name = r[0].strip() if r else ''
        grp = r[gi].strip() if gi < len(r) else ''
        if k and name and grp:
            fname.append(name); group.append(grp)
    return Groups(fname, group)


def match_prefix_to_bin(prefixes, bins, suff):
    """the path among `bins` whose file name is prefix + suff, for every prefix; all missing ones are reported together"""
    by_name = {}
    for f in bins:
        by_name.setdefault(op.basename(f), f)
    missing = [p for p in prefixes if p + suff not in by_name]
    if missing:
        eprint(f'Error: {len(missing)} prefixes from groups file were not found in input bins:')
        for p in missing:
            eprint(p)
        raise IllegalArgumentError('groups file mismatch binary files')
    return [by_name[p + suff] for p in prefixes]


def groups_load_wrap(groups_file, betas):
    if groups_file is not None:
        if not op.isfile(groups_file):
            raise IllegalArgumentError(f'Invalid file: {groups_file}')
        gf = load_gfile_helper(groups_file)
    else:                                                          # every file its own group, named after the file
        betas = drop_dup_keep_order(list(betas))
        names = [pretty_name(b) for b in betas]
        gf = Groups(names, names)
    suff = '.lbeta' if betas[0].endswith('.lbeta') else '.beta'
    gf.full_path = match_prefix_to_bin(gf.fname, betas, suff)
    return gf


class Table:
    """blocks + value columns, ready to print"""

    def __init__(self, blocks, names, values):
        self.blocks, self.names, self.values = blocks, list(names), values       # values: float64 [n_blocks][len(names)]

    @property
    def shape(self):
        return (len(self.blocks), self.blocks.shape[1] + len(self.names))


def get_table(blocks_df, gf, min_cov, threads=8, verbose=False, group=True, engine=None):
    """One chunk of blocks -> Table (per sample when group is False, else per group: nanmean over the group's samples)."""
    if verbose:
        eprint(f'[wt table] reducing to {len(blocks_df):,} blocks')
    betas = drop_dup_keep_order(gf.full_path)
    eng = engine or BlockSumEngine(betas)
    try:
        vecs = eng.reduce(blocks_df, mode=3, min_cov=min_cov)                     # [n_betas][n_blocks] float64
    finally:
        if engine is None:
            eng.close()
    by_name = {pretty_name(b): vecs[i] for i, b in enumerate(betas)}
    if not group:
        names = list(gf.fname)
        cols = [by_name[k] for k in names]
    else:
        names = drop_dup_keep_order(gf.group)
        cols = []
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', category=RuntimeWarning)             # all-NaN rows: the mean is NaN, printed NA
            for g in names:
                members = [f for f, gg in zip(gf.fname, gf.group) if gg == g]
                cols.append(np.nanmean(np.stack([by_name[k] for k in members]), axis=0))
    vals = np.stack(cols, axis=1) if cols else np.zeros((len(blocks_df), 0))
    return Table(blocks_df, names, vals)


def betas2table(betas, blocks, groups_file, min_cov, threads=8, verbose=False):
    gf = groups_load_wrap(groups_file, betas)
    return get_table(load_blocks_file(blocks), gf, min_cov, threads, verbose)


def dump(outpath, table, first=True, digits=3):
    """Append (or start, with the header) the text of a Table: tab-separated, NA for missing, %.<digits>f floats."""
    own = not hasattr(outpath, 'write') and outpath is not None
    b = table.blocks
    if b.parsed is not None and not b.extra and len(b) and (own or outpath is None):
        # the library prints the rows (include/wgbsseg.h: wgbsseg_blocks_write_table): same bytes as the loop below
        from . import _lib
        if own:
            if first:
                with open(outpath, 'w') as f:
                    f.write('\t'.join(b.columns + table.names) + '\n')
        else:
            if first:
                sys.stdout.write('\t'.join(b.columns + table.names) + '\n')
            sys.stdout.flush()
        _lib.blocks_write_table(outpath if own else None, b.parsed, table.values, digits, append=True)
        return
    f = open(outpath, 'w' if first else 'a') if own else (sys.stdout if outpath is None else outpath)
    try:
        b = table.blocks
        if first:
            f.write('\t'.join(b.columns + table.names) + '\n')
        s_txt, e_txt = b.cpg_text()
        fmt = '%%.%df' % digits
        extra = [b.extra[k] for k in b.extra]
        vals = table.values
        lines = []
        for i in range(len(b)):
            row = [b.chr[i], b.start[i], b.end[i], s_txt[i], e_txt[i]] + [x[i] for x in extra]
            row += ['NA' if v != v else fmt % v for v in vals[i].tolist()]
            lines.append('\t'.join(row))
        if lines:
            f.write('\n'.join(lines) + '\n')
    finally:
        if own:
            f.close()


def beta2table_generator(betas, blocks, groups_file, min_cov, threads, chunk_size=None, verbose=False, device=0):
    if not op.isfile(blocks):
        raise IllegalArgumentError(f'Invalid file: {blocks}')
    gf = groups_load_wrap(groups_file, betas)
    table = load_blocks_file(blocks)
    step = chunk_size or max(1, len(table))
    eng = BlockSumEngine(drop_dup_keep_order(gf.full_path), device=device)        # the files go to the device once
    try:
        for a in range(0, len(table), step):
            yield get_table(table.rows(a, a + step), gf, min_cov, threads, verbose, engine=eng)
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
    first = True
    for chunk in beta2table_generator(args.betas, args.blocks, args.groups_file, args.min_cov, args.threads,
                                      args.chunk_size, args.verbose, args.device):
        dump(args.output, chunk, first, args.digits)
        first = False


if __name__ == '__main__':
    main()
