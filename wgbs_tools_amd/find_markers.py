"""`wgbstools find_markers` on MI355X: differentially methylated blocks between groups of samples,
over the block tables `segment` writes.

Drop-in for the reference's src/python/find_markers.py + fm_load_params.py (same flags, config-file format, defaults, output
files, messages).  The rules, per target group T against the background samples B, for every block (find_markers.py:181-372):

    coverage      a sample counts for a block when it has >= min_cov observations there; a block is kept when at least
                  (1 - na_rate) of T's and of B's samples count
    U markers     mean(T) <= unmeth_mean_thresh, mean(B) >= meth_mean_thresh, mean(B) - mean(T) >= delta_means; then with
                  qT = quantile(T, 1 - tg_quant), qB = quantile(B, bg_quant) (linear interpolation, missing values ignored):
                  qT <= unmeth_quant_thresh, qB >= meth_quant_thresh, qB - qT >= delta_quants
    M markers     the same with T and B (and their quantile parameters) exchanged; the reported means are swapped back
    tests         t-test, Mann-Whitney U and a Welch t-test on M-values (log2(b / (1 - b)), b clipped to [1e-4, 1 - 1e-4]) per
                  marker; the one named by --test_type filters at --pval
    output        Markers.<target>.bed: the block's columns, target, region, lenCpG, bp, tg_mean, bg_mean, delta_means,
                  delta_quants, delta_maxmin (= min(B) - max(T)), the three p-values, direction (U / M), %.3g floats

The data side runs on the GPU: the blocks x samples table of meth/cov ratios is ONE block-reduction launch per chunk of
blocks (wgbsseg_block_sums mode 3) and stays on the device; per target, the per-block group statistics the filters need —
sample counts, sums, minima, maxima of T and B — come from a second kernel over that table (wgbsseg_marker_stats).  Quantiles
and the statistical tests are evaluated on the host for the few blocks that pass the mean filters, with the same numpy /
scipy routines the reference calls.  No CPU fallback for the reduction.
"""
import argparse
import os
import os.path as op
import re
import sys
import warnings
from difflib import get_close_matches
from math import ceil

import numpy as np

from .beta_to_blocks import BlockSumEngine, BlocksTable, load_blocks_file
from .beta_to_table import drop_dup_keep_order, load_gfile_helper, match_prefix_to_bin
from .genome import IllegalArgumentError, eprint
from .cliutil import add_threads_option, require_file, require_files

# supplemental/find_markers_defaults.txt of the reference (key:value lines), as data
DEFAULTS = dict(blocks_path=None, groups_file=None, targets=None, background=None, beta_list_file=None, betas=None, min_bp=0,
                max_bp=10000000000, min_cpg=0, max_cpg=10000000000, min_cov=5, na_rate_tg=.334, na_rate_bg=.334, only_hyper=False,
                only_hypo=False, delta_means=0.3, delta_quants=0.0, tg_quant=0.25, bg_quant=0.025, unmeth_quant_thresh=1.0,
                meth_quant_thresh=0.0, unmeth_mean_thresh=1.0, meth_mean_thresh=0.0, out_dir='.', top=None, header=False,
                verbose=False, chunk_size=150000, pval=0.05, test_type='t', sort_by=None, delta_maxmin=-1)


def typed(val):
    """fm_load_params.py:63-71: a config value as int / float / None where it looks like one"""
    if isinstance(val, float) and val != val:
        return None
    if isinstance(val, str) and val.isdigit():
        return int(val)
    if isinstance(val, str) and re.match(r"[-+]?\d*\.\d+|\d+", val):
        return float(val)
    return val


def load_param_file(path):
    """key:value lines, '#' comments; NA / empty -> None, True / False, numbers, `targets` as a list"""
    if not path:
        return {}
    require_file(path)
    d = {}
    with open(path) as f:
        for line in f:
            line = line.split('#', 1)[0].strip()
            if not line or ':' not in line:
                continue
            key, val = line.split(':', 1)
            key, val = key.strip(), val.strip()
            val = None if val in ('', 'NA', 'NaN', 'nan', 'None') else typed(val)
            if val == 'True':
                val = True
            elif val == 'False':
                val = False
            elif key == 'targets' and val is not None:
                val = str(val).split()
            d[key] = val
    return d


# ---- parameter rules, as data --------------------------------------------------------------------------------------
# (the reference checks them one `if` after another, fm_load_params.py:79-136; the messages, and which of them go to stderr
# ahead of an exception without text, are its own)
_AT_LEAST = (('min_cpg', 0, 'min_cpg must be non negative'), ('max_cpg', 1, 'max_cpg must larger than 0'),
             ('min_bp', 0, 'min_bp must be non negative'), ('max_bp', 2, 'max_bp must be larger than 1'),
             ('chunk_size', 1, 'chunk_size must be larger than 1'))
_WITHIN = ((0, 1, ('na_rate_tg', 'na_rate_bg', 'tg_quant', 'bg_quant', 'unmeth_quant_thresh', 'meth_quant_thresh', 'unmeth_mean_thresh',
                   'meth_mean_thresh', 'pval')),
           (-1, 1, ('delta_means', 'delta_quants', 'delta_maxmin')))
_ONE_OF = (('sort_by', ('delta_means', 'delta_quants', 'delta_maxmin', 'startCpG', 'tg_quant', 'tg_mean'), True),    # True: None is allowed
           ('test_type', ('t', 'mw', 'm_t'), False))
_INPUT_FILES = ('blocks_path', 'groups_file')


def _refuse(text):
    """the reference's pattern for most parameter errors: the text on stderr, then an exception without one"""
    eprint(text)
    raise IllegalArgumentError()


class MFParams:
    """defaults <- config file <- command line (fm_load_params.py:14-40), then the rules above."""

    def __init__(self, args):
        layers = (DEFAULTS, load_param_file(args.config_file),
                  {k: v for k, v in vars(args).items() if v is not None and v is not False})      # flags that were not given leave the file's values alone
        for layer in layers:
            self.__dict__.update(layer)
        self.check()

    def check(self):
        for key, low, text in _AT_LEAST:
            if getattr(self, key) < low:
                raise IllegalArgumentError(text)
        for low, high, keys in _WITHIN:
            for key in keys:
                val = float(getattr(self, key))
                if not (high >= val >= low):
                    _refuse(f'Invalid value for {key} ({val}): must be in [{low}, {high}]')
        if self.only_hyper and self.only_hypo:
            _refuse('at most one of (only_hyper, only_hypo) can be specified')
        for key, allowed, optional in _ONE_OF:
            val = getattr(self, key)
            if not (val in allowed or (optional and val is None)):
                _refuse(f'{key} argument must be in: {", ".join(allowed)}')
        for key in _INPUT_FILES:
            if getattr(self, key) is None:
                _refuse(f'[wt fm] missing required parameter: {key}')
            setattr(self, key, op.abspath(require_file(getattr(self, key))))
        if (self.betas is None) == (self.beta_list_file is None):
            _refuse('[wt fm] Exactly one of the following must be specified: betas, beta_list_file')
        if self.beta_list_file:
            require_file(self.beta_list_file)
            with open(self.beta_list_file) as f:
                self.betas = [ln.strip() for ln in f if ln.strip()]
        elif isinstance(self.betas, str):
            self.betas = self.betas.split()
        require_files(self.betas)

    validate_args = check                                         # the reference's name for it


# ---- command line, as data: (flags, type or action, help); every option defaults to None / False = "not given" ------------------
_OPTIONS = (
    (('--config_file', '-p'), str, 'find_markers config file (key:value lines)'),
    (('--blocks_path', '-b'), str, 'Blocks bed path.'),
    (('--groups_file', '-g'), str, 'csv file of groups'),
    (('--targets',), '+', 'find markers only for these groups (OR relation)'),
    (('--background',), '+', 'find markers only against these groups (AND relation)'),
    (('-o', '--out_dir'), str, 'Output directory'),
    (('--min_bp',), int, None), (('--max_bp',), int, None), (('--min_cpg',), int, None), (('--max_cpg',), int, None),
    (('--delta_means',), float, 'Filter markers by beta values delta_means. range: [0.0, 1.0]. Default [0.3].'),
    (('--delta_quants',), float, 'Filter markers by beta values delta_quants. range: [0.0, 1.0]. Default [0.0]'),
    (('-c', '--min_cov'), int, 'Minimal number of binary observations in block coverage to be considered. [5]'),
    (('--only_hyper',), True, 'Only consider hyper-methylated markers'),
    (('--only_hypo',), True, 'Only consider hypo-methylated markers'),
    (('--top',), int, 'Output only the top TOP markers, under the constraints. [All]'),
    (('--header',), True, 'add header to output files'),
    (('--tg_quant',), float, 'quantile of target samples to ignore. [0.25]'),
    (('--bg_quant',), float, 'quantile of background samples to ignore. [0.025]'),
    (('--unmeth_mean_thresh',), float, 'average beta value for the unmethylated group'),
    (('--meth_mean_thresh',), float, 'average beta value for the methylated group'),
    (('--unmeth_quant_thresh',), float, 'quantlie beta value for the unmethylated group'),
    (('--meth_quant_thresh',), float, 'quantlie beta value for the methylated group'),
    (('--na_rate_tg',), float, 'rate of samples with insufficient coverage allowed in target samples. [.334]'),
    (('--na_rate_bg',), float, 'rate of samples with insufficient coverage allowed in background samples. [.334]'),
    (('--pval',), float, 'p-value threshold. DMRs with larger p-value are dropped. [0.05]'),
    (('--test_type',), str, 'The statistical test used for p-value filtering: t (two-sample t-test), mw (Mann-Whitney U), m_t (t-test on M-values). [t]'),
    (('--sort_by',), str, 'sort output markers by this column.'),
    (('--chunk_size',), int, 'Number of blocks to load on each step'),
    (('--verbose', '-v'), True, None),
)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Find differentially methylated blocks')
    betas = parser.add_mutually_exclusive_group()
    betas.add_argument('--betas', nargs='+', help='beta file paths. files not in the group files are ignored')
    betas.add_argument('--beta_list_file', help='file with a list of beta file paths.')
    for flags, kind, text in _OPTIONS:
        kw = dict(action='store_true') if kind is True else dict(nargs='+') if kind == '+' else dict(type=kind)
        parser.add_argument(*flags, help=text, **kw)
    add_threads_option(parser)
    parser.add_argument('--device', type=int, default=0, help='HIP device index [0]')
    return parser.parse_args(argv)


def get_validate_targets(subset, groups):
    """the whole list of groups when no subset is named; an unknown group is an error (with a suggestion)"""
    if not subset or subset[0] in ('NA', 'None'):
        return groups
    for group in [g for item in subset for g in item.split('+')]:
        if group not in groups:
            eprint(f'Invalid group: {group}')
            close = get_close_matches(group, groups)
            if close:
                eprint(f'Did you mean {close[0]}?')
            eprint('All possible groups:', groups)
            raise IllegalArgumentError()
    return subset


# column sets of a marker row that are not the block's own columns
STAT_COLS = ['tg_mean', 'bg_mean', 'delta_means', 'delta_quants', 'delta_maxmin', 'ttest', 'mw_test', 'mvalue_ttest']


class Markers:
    """rows found for one target: indexes into the (filtered) blocks table + their statistics, column-wise"""

    def __init__(self):
        self.row = np.zeros(0, dtype=np.int64)
        self.direction = []
        self.cols = {k: np.zeros(0) for k in STAT_COLS + ['tg_quant', 'bg_quant']}

    def extend(self, other):
        self.row = np.concatenate([self.row, other.row])
        self.direction += other.direction
        for k in self.cols:
            self.cols[k] = np.concatenate([self.cols[k], other.cols[k]])

    def take(self, idx):
        m = Markers()
        m.row = self.row[idx]
        m.direction = [self.direction[i] for i in np.asarray(idx).tolist()] if len(self.direction) else []
        m.cols = {k: v[idx] for k, v in self.cols.items()}
        return m

    def __len__(self):
        return int(self.row.size)


def descending_order(vals):
    """the order pandas' sort_values(ascending=False) gives (its default quicksort on the reversed column, NaNs last)"""
    vals = np.asarray(vals, dtype=np.float64)
    idx = np.arange(vals.size)
    nan = np.isnan(vals)
    good, gv = idx[~nan][::-1], vals[~nan][::-1]
    return np.concatenate([good[gv.argsort(kind='quicksort')][::-1], idx[nan]])


class MarkerFinder:
    def __init__(self, args, engine=None):
        self.args = args
        self.verbose = args.verbose
        self.chunk_count = 0
        self.nr_chunks = 1
        self.engine = engine
        if args.out_dir:
            os.makedirs(args.out_dir, exist_ok=True)
        require_file(args.groups_file)
        require_files(args.betas)
        gf = load_gfile_helper(args.groups_file)
        gf.full_path = match_prefix_to_bin(gf.fname, args.betas, '.beta')
        groups = sorted(set(gf.group))
        self.targets = get_validate_targets(args.targets, groups)
        self.background = get_validate_targets(args.background, groups)
        self.inds = {}
        for group in self.targets:
            tg = [f for f, g in zip(gf.fname, gf.group) if g == group]
            bg = [f for f in drop_dup_keep_order([f for f, g in zip(gf.fname, gf.group) if g in self.background]) if f not in tg]
            assert len(bg) + len(tg) <= len(set(gf.fname))
            assert len(bg)
            assert len(tg)
            self.inds[group] = (tg, bg)
        keep = [i for i, g in enumerate(gf.group) if g in list(self.background) + list(self.targets)]
        self.fname = [gf.fname[i] for i in keep]
        self.paths = drop_dup_keep_order([gf.full_path[i] for i in keep])          # the beta files that go to the device
        self.col = {}                                                              # sample name -> row of the device table
        for i in keep:
            self.col.setdefault(gf.fname[i], self.paths.index(gf.full_path[i]))
        self.res = {t: Markers() for t in self.targets}

    # ---- data -------------------------------------------------------------------------------------------------------
    def load_blocks(self):
        t = load_blocks_file(self.args.blocks_path, anno=True)
        if not len(t):
            return t
        n0 = len(t)
        if t.parsed is not None and t.parsed.bp_start is not None:      # (the library's parser has the two columns as integers)
            start, end = t.parsed.bp_start, t.parsed.bp_end
        else:
            start = np.array([int(x) for x in t.start], dtype=np.int64)
            end = np.array([int(x) for x in t.end], dtype=np.int64)
        ln_cpg = t.endCpG - t.startCpG
        ln = end - start
        a = self.args
        keep = (~t.na) & (ln_cpg >= a.min_cpg) & (ln_cpg <= a.max_cpg) & (ln >= a.min_bp) & (ln <= a.max_bp)
        idx = np.flatnonzero(keep)
        t = t.take(idx)
        if self.verbose:
            eprint(f'loaded {n0:,} blocks')
            if len(t) != n0:
                eprint(f'droppd to {len(t):,} ')
        return t

    def run(self):
        self.dump_params()
        blocks = self.load_blocks()
        if not len(blocks):
            eprint('Empty block set. Abort')
            return
        self.blocks = blocks
        step = self.args.chunk_size
        self.nr_chunks = ceil(len(blocks) / step)
        if self.verbose:
            eprint(f'processing data in {self.nr_chunks} chunks...')
        own = self.engine is None
        eng = BlockSumEngine(self.paths, device=getattr(self.args, 'device', 0)) if own else self.engine
        try:
            for a in range(0, len(blocks), step):
                self.proc_chunk(eng, a, min(a + step, len(blocks)))
        finally:
            if own:
                eng.close()
        for target in self.targets:
            self.dump_results(target, self.res[target])

    def proc_chunk(self, eng, a, b):
        if self.verbose:
            self.chunk_count += 1
            eprint(f'{self.chunk_count}/{self.nr_chunks} ) loading data for {b - a:,} blocks over {len(set(self.fname))} samples...')
        table = eng.reduce(self.blocks.rows(a, b), mode=3, min_cov=self.args.min_cov)          # [files][blocks]; also left on the device
        for group in self.targets:
            tg, bg = self.inds[group]
            stats = eng.marker_stats([self.col[f] for f in tg], [self.col[f] for f in bg], b - a)
            found = self.find_group_markers(table, stats, tg, bg)
            found.row = found.row + a
            self.res[group].extend(found)

    # ---- the rules --------------------------------------------------------------------------------------------------
    def find_group_markers(self, table, stats, tg, bg):
        a = self.args
        n_tg, s_tg, mn_tg, mx_tg, n_bg, s_bg, mn_bg, mx_bg = stats.T
        covered = (n_tg / len(tg) >= 1 - a.na_rate_tg) & (n_bg / len(bg) >= 1 - a.na_rate_bg)
        with np.errstate(divide='ignore', invalid='ignore'):
            mean_tg, mean_bg = s_tg / n_tg, s_bg / n_bg
        tcols = np.array([self.col[f] for f in tg])
        bcols = np.array([self.col[f] for f in bg])
        out = Markers()
        # U: the target low, the background high.  M: the roles (and the quantile parameters) exchanged.
        for direction, skip in (('U', a.only_hyper), ('M', a.only_hypo)):
            if skip:
                continue
            if direction == 'U':
                lo_mean, hi_mean, lo_cols, hi_cols, q_lo, q_hi = mean_tg, mean_bg, tcols, bcols, a.tg_quant, a.bg_quant
                maxmin = mn_bg - mx_tg
            else:
                lo_mean, hi_mean, lo_cols, hi_cols, q_lo, q_hi = mean_bg, mean_tg, bcols, tcols, a.bg_quant, a.tg_quant
                maxmin = mn_tg - mx_bg
            with np.errstate(invalid='ignore'):
                dm = hi_mean - lo_mean
                keep = covered & (lo_mean <= a.unmeth_mean_thresh) & (hi_mean >= a.meth_mean_thresh) & (dm >= a.delta_means)
            rows = np.flatnonzero(keep)
            if not rows.size:
                continue
            with warnings.catch_warnings():
                warnings.simplefilter('ignore', category=RuntimeWarning)
                lo_q = np.nanquantile(table[lo_cols][:, rows].T, 1 - q_lo, axis=1)
                hi_q = np.nanquantile(table[hi_cols][:, rows].T, q_hi, axis=1)
            dq = hi_q - lo_q
            ok = (lo_q <= a.unmeth_quant_thresh) & (hi_q >= a.meth_quant_thresh) & (dq >= a.delta_quants)
            rows, dq, lo_q, hi_q = rows[ok], dq[ok], lo_q[ok], hi_q[ok]
            if not rows.size:
                continue
            m = Markers()
            m.row = rows
            m.direction = [direction] * rows.size
            # an M marker is reported with the target's / background's own means (the exchange is undone), the deltas as computed
            m.cols['tg_mean'] = mean_tg[rows]
            m.cols['bg_mean'] = mean_bg[rows]
            m.cols['delta_means'] = dm[rows]
            m.cols['delta_quants'] = dq
            m.cols['delta_maxmin'] = maxmin[rows]
            m.cols['tg_quant'] = lo_q
            m.cols['bg_quant'] = hi_q
            for k in ('ttest', 'mw_test', 'mvalue_ttest'):
                m.cols[k] = np.full(rows.size, np.nan)
            out.extend(m)
        if not len(out):
            return out
        tv = table[tcols][:, out.row].T
        bv = table[bcols][:, out.row].T
        for name, fn in (('ttest', self.ttest), ('mw_test', self.mw_test), ('mvalue_ttest', self.m_value_ttest)):
            if not len(out):
                break
            keep = fn(out, tv, bv, name)
            if keep is not None:
                out = out.take(keep)
                tv, bv = tv[keep], bv[keep]
        return out

    def _filter(self, out, name, wanted):
        if self.args.test_type != wanted:
            return None
        with np.errstate(invalid='ignore'):
            return np.flatnonzero(out.cols[name] <= self.args.pval)

    def ttest(self, out, tv, bv, name):
        try:
            if tv.shape[1] == bv.shape[1] == 1:
                return None
            from scipy.stats import ttest_1samp, ttest_ind
            if tv.shape[1] == 1:
                r = ttest_1samp(bv, tv, axis=1, nan_policy='omit')
            elif bv.shape[1] == 1:
                r = ttest_1samp(tv, bv, axis=1, nan_policy='omit')
            else:
                r = ttest_ind(tv, bv, axis=1, nan_policy='omit')
            out.cols[name] = np.asarray(r.pvalue, dtype=np.float64)
            return self._filter(out, name, 't')
        except ModuleNotFoundError:
            eprint('[wt fm] WARNING: scipy is not installed. T-test is not performed.')
        except Exception:
            eprint('[wt fm] WARNING: Exception occured while computing T-test. T-test is not performed.')
        return None

    def mw_test(self, out, tv, bv, name):
        try:
            if tv.shape[1] == bv.shape[1] == 1:
                return None
            from scipy.stats import mannwhitneyu
            r = mannwhitneyu(tv, bv, axis=1, nan_policy='omit', alternative='two-sided')
            out.cols[name] = np.asarray(r.pvalue, dtype=np.float64)
            return self._filter(out, name, 'mw')
        except ModuleNotFoundError:
            eprint('[wt fm] WARNING: scipy is not installed. MW-test is not performed.')
        except Exception as e:
            eprint(f'[wt fm] WARNING: Exception occured while computing MW-test: {e}')
        return None

    def m_value_ttest(self, out, tv, bv, name):
        try:
            if tv.shape[1] == bv.shape[1] == 1:
                return None
            from scipy.stats import ttest_1samp, ttest_ind
            tc, bc = np.clip(tv, 0.0001, 0.9999), np.clip(bv, 0.0001, 0.9999)
            tm, bm = np.log2(tc / (1 - tc)), np.log2(bc / (1 - bc))
            if tv.shape[1] == 1:
                r = ttest_1samp(bm, tm, axis=1, nan_policy='omit')
            elif bv.shape[1] == 1:
                r = ttest_1samp(tm, bm, axis=1, nan_policy='omit')
            else:
                r = ttest_ind(tm, bm, axis=1, equal_var=False, nan_policy='omit')
            out.cols[name] = np.asarray(r.pvalue, dtype=np.float64)
            return self._filter(out, name, 'm_t')
        except ModuleNotFoundError:
            eprint('[wt fm] WARNING: scipy is not installed. T-test is not performed.')
        except Exception as e:
            eprint(f'[wt fm] WARNING: Exception occured while computing T-test: {e}')
        return None

    # ---- output -----------------------------------------------------------------------------------------------------
    def dump_results(self, target, m):
        eprint(f'Number of markers found: {len(m):,}')
        if not len(m):
            return
        a = self.args
        if a.sort_by:
            key = self.blocks.startCpG[m.row] if a.sort_by == 'startCpG' else m.cols[a.sort_by]
            m = m.take(descending_order(key))
        if a.top:
            m = m.take(np.arange(min(a.top, len(m))))
        b = self.blocks
        tg, bg = self.inds[target]
        outpath = op.join(a.out_dir, f'Markers.{target}.bed')
        eprint(f'dumping to {outpath}')
        cols = ['#chr', 'start', 'end', 'startCpG', 'endCpG', 'target', 'region', 'lenCpG', 'bp'] + STAT_COLS + ['direction']
        with_anno = 'anno' in b.extra and 'gene' in b.extra
        if with_anno:
            cols += ['anno', 'gene']

        def g3(v):
            return 'NA' if v != v else '%.3g' % v
        with open(outpath, 'w') as f:
            if a.header:
                for s in sorted(tg):
                    f.write(f'#> {s}\n')
                for s in sorted(bg):
                    f.write(f'#< {s}\n')
            f.write('\t'.join(cols) + '\n')
            stat = [m.cols[k].tolist() for k in STAT_COLS]
            coords = b.coords_of(m.row)                                        # the text of the markers' rows only
            extras = b.extras_of(m.row) if with_anno else {}
            for j, r in enumerate(m.row.tolist()):
                chrom, s, e = coords[j][0], int(coords[j][1]), int(coords[j][2])
                row = [chrom, str(s), str(e), str(int(b.startCpG[r])), str(int(b.endCpG[r])), target, f'{chrom}:{s}-{e}',
                       f'{int(b.endCpG[r] - b.startCpG[r])}CpGs', f'{e - s}bp'] + [g3(x[j]) for x in stat] + [m.direction[j]]
                if with_anno:
                    row += [extras['anno'][j] or 'NA', extras['gene'][j] or 'NA']
                f.write('\t'.join(row) + '\n')

    def dump_params(self):
        outpath = op.join(self.args.out_dir, 'params.txt')
        with open(outpath, 'w') as f:
            for key in vars(self.args):
                val = getattr(self.args, key)
                if key == 'beta_list_file':
                    val = None
                if key == 'betas':
                    val = ' '.join(val)
                if key == 'targets' and val is not None:
                    val = ' '.join(val)
                f.write(f'{key}:{val}\n')
        eprint(f'dumped parameter file to {outpath}')


def main(argv=None):
    """
    Find differentially methylated blocks
    """
    MarkerFinder(MFParams(parse_args(argv))).run()


if __name__ == '__main__':
    main()
