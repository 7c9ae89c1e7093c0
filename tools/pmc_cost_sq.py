#!/usr/bin/env python3
"""Where do k_cost's wavefront cycles go?  One rocprofv3 PMC pass (8 SQ counters) over a short bench run, summed
over the k_cost dispatches:
    cd /tmp && TMPDIR=/tmp python $REPO/tools/pmc_cost_sq.py [bench args]
Writes gpurun_out/pmc_cost_sq.json.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wavefront
(MI355X_MICROARCH.md); SQ_LDS_BANK_CONFLICT = extra LDS cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles."""
import csv
import glob
import json
import os
import os.path as op
import subprocess
import sys

ROOT = op.dirname(op.dirname(op.abspath(__file__)))
OUT = op.join(ROOT, 'gpurun_out')
COUNTERS = ['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_WAIT_ANY',
            'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE']


def main():
    d = op.join(OUT, 'pmc_cost_sq')
    cmd = ['rocprofv3', '--kernel-trace', '--pmc'] + COUNTERS + ['--output-format', 'csv', '-d', d, '-o', 'pmc', '--',
           sys.executable, op.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0', '--cpu-seconds', '0', '--e2e', '0', '--extras', '0', '--matrix', '0', '--block-sums', '0', '--scan-carries', '0'] + sys.argv[1:]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    files = glob.glob(op.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        print(r.stdout[-3000:])
        raise SystemExit('no counter_collection.csv')
    tot, n = {}, 0
    for f in files:
        for row in csv.DictReader(open(f)):
            if 'k_cost' not in row.get('Kernel_Name', ''):
                continue
            tot[row['Counter_Name']] = tot.get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
            n += 1
    sys.path.insert(0, ROOT)
    from wgbs_tools_amd import build
    res = {'csrc_sha': build.source_hash(), 'counters': tot, 'rows': n, 'command': ' '.join(cmd[cmd.index('--') + 1:])}
    w = tot.get('SQ_WAVE_CYCLES', 0.0)
    if w:
        res['frac_of_wave_cycles'] = {k: v / w for k, v in tot.items() if k.startswith(('SQ_ACTIVE', 'SQ_WAIT'))}
    if tot.get('SQ_LDS_IDX_ACTIVE'):
        res['lds_conflict_frac'] = tot.get('SQ_LDS_BANK_CONFLICT', 0.0) / tot['SQ_LDS_IDX_ACTIVE']
    os.makedirs(OUT, exist_ok=True)
    json.dump(res, open(op.join(OUT, 'pmc_cost_sq.json'), 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
