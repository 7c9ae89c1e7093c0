#!/usr/bin/env python3
"""Golden vectors of `wgbstools find_markers` from the REFERENCE ITSELF: runs only in the build
container, imports /root/reference/src/python/find_markers.py (with its fm_load_params, beta_to_table, beta_to_blocks, dmb)
and runs MarkerFinder on seeded synthetic beta files and a blocks table (tests/cases.py: marker_world).  Nothing of the
reference is replaced.  Writes tests/golden/marker_cases.json: per case the command-line arguments and the text of every
Markers.<target>.bed it wrote (+ stderr)."""
import contextlib
import io
import json
import os
import os.path as op
import sys
import tempfile

HERE = op.dirname(op.abspath(__file__))
ROOT = op.dirname(op.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, op.join(ROOT, 'tests'))
sys.path.insert(0, '/root/reference/src/python')

import cases                                   # noqa: E402

CASES = {
    'default':        [],
    'hypo_top':       ['--only_hypo', '--top', '7', '--sort_by', 'delta_means', '--header'],
    'hyper_quants':   ['--only_hyper', '--delta_quants', '0.2', '--tg_quant', '0.3', '--bg_quant', '0.1', '--unmeth_quant_thresh', '0.4',
                       '--meth_quant_thresh', '0.5'],
    'two_targets_bg': ['--targets', 'Liver', 'Blood', '--background', 'Colon', 'Lung', '--delta_means', '0.25', '--min_cpg', '8', '--max_bp', '4000'],
    'mw_test':        ['--test_type', 'mw', '--pval', '0.2', '--na_rate_tg', '0.5', '--na_rate_bg', '0.5', '-c', '25'],
    'mvalue_test':    ['--test_type', 'm_t', '--pval', '0.01', '--chunk_size', '700', '--sort_by', 'delta_maxmin'],
    'single_sample_target': ['--targets', 'Solo', '--delta_means', '0.2', '--pval', '1'],
}


def main():
    import find_markers as rf
    import fm_load_params as rp
    td = tempfile.mkdtemp()
    w = cases.marker_world(td)
    out = {'world': w['spec'], 'cases': {}}
    for name, extra in CASES.items():
        od = op.join(td, 'out_' + name)
        argv = ['find_markers', '-b', w['blocks'], '-g', w['groups'], '--betas'] + w['betas'] + ['-o', od] + extra
        err = io.StringIO()
        old = sys.argv
        sys.argv = argv
        try:
            with contextlib.redirect_stderr(err):
                rf.MarkerFinder(rp.MFParams(rp.parse_args())).run()
        finally:
            sys.argv = old
        files = {f: open(op.join(od, f)).read() for f in sorted(os.listdir(od)) if f.startswith('Markers.')}
        stderr = err.getvalue().replace(td, '<TMP>')
        out['cases'][name] = {'args': extra, 'files': files, 'stderr': stderr}
        print(name, {f: t.count('\n') for f, t in files.items()})
    with open(op.join(HERE, 'marker_cases.json'), 'w') as f:
        json.dump(out, f, separators=(',', ':'))
    print('wrote marker_cases.json (%.0f KB)' % (op.getsize(op.join(HERE, 'marker_cases.json')) / 1e3))


if __name__ == '__main__':
    main()
