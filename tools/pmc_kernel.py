#!/usr/bin/env python3
"""One rocprofv3 PMC pass (counters named on the command line) over a short bench.py run, summed over the dispatches of one kernel:
    cd /tmp && TMPDIR=/tmp python $REPO/tools/pmc_kernel.py --kernel k_block_sums --out pmc_bs.json [--counters A B ..] -- [bench args]
Writes gpurun_out/<out>.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wavefront (MI355X_MICROARCH.md)."""
import argparse
import csv
import glob
import json
import os
import os.path as op
import subprocess
import sys

ROOT = op.dirname(op.dirname(op.abspath(__file__)))
OUT = op.join(ROOT, 'gpurun_out')
DEFAULT = ['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_WAIT_ANY',
           'SQ_ACTIVE_INST_VMEM', 'SQ_INSTS_VALU']


def main():
    argv = sys.argv[1:]
    extra = []
    if '--' in argv:
        extra = argv[argv.index('--') + 1:]
        argv = argv[:argv.index('--')]
    ap = argparse.ArgumentParser()
    ap.add_argument('--kernel', required=True)
    ap.add_argument('--out', default='pmc_kernel.json')
    ap.add_argument('--counters', nargs='+', default=DEFAULT)
    a = ap.parse_args(argv)
    d = op.join(OUT, 'pmc_' + a.kernel)
    cmd = ['rocprofv3', '--kernel-trace', '--pmc'] + a.counters + ['--output-format', 'csv', '-d', d, '-o', 'pmc', '--',
           sys.executable, op.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0', '--cpu-seconds', '0', '--e2e', '0', '--extras', '0', '--matrix', '0', '--block-sums', '0', '--scan-carries', '0'] + extra
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    files = glob.glob(op.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        print(r.stdout[-3000:])
        raise SystemExit('no counter_collection.csv')
    tot, n = {}, 0
    for f in files:
        for row in csv.DictReader(open(f)):
            if a.kernel not in row.get('Kernel_Name', ''):
                continue
            tot[row['Counter_Name']] = tot.get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
            n += 1
    sys.path.insert(0, ROOT)
    from wgbs_tools_amd import build
    res = {'csrc_sha': build.source_hash(), 'kernel': a.kernel, 'counters': tot, 'rows': n, 'command': ' '.join(cmd[cmd.index('--') + 1:])}
    w = tot.get('SQ_WAVE_CYCLES', 0.0)
    if w:
        res['frac_of_wave_cycles'] = {k: v / w for k, v in tot.items() if k.startswith(('SQ_ACTIVE', 'SQ_WAIT'))}
    os.makedirs(OUT, exist_ok=True)
    json.dump(res, open(op.join(OUT, a.out), 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
