#!/usr/bin/env python3
"""HBM-side traffic of the scoring kernel (k_cost: the step) from the PMC counters, the way MI355X_MICROARCH.md prescribes: FETCH_SIZE and
WRITE_SIZE in SEPARATE rocprofv3 passes (with --kernel-trace only), units KB = 1024 B.  The guide calibrates FETCH_SIZE only for 16 B per lane
streaming reads (x2 on gfx950); k_cost stages its rows with 8 B per lane loads and writes 8 B per lane, so the same passes are first run over
tools/micro/fetch_calib (1 GiB read / written once with 4, 8, 16 B per lane) and the measured bytes-per-counter-KB factors are applied.
Runs on the GPU box; writes gpurun_out/cost_traffic.json (copy it to profiles/ as rNN_cost_traffic.json: bench.py reports it when csrc_sha matches).

    cd /tmp && TMPDIR=/tmp python $REPO/tools/pmc_cost_traffic.py [bench args]
"""
import csv
import glob
import json
import os
import os.path as op
import subprocess
import sys

ROOT = op.dirname(op.dirname(op.abspath(__file__)))
OUT = op.join(ROOT, 'gpurun_out')
EXTRA = sys.argv[1:]
BENCH = [sys.executable, op.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0', '--cpu-seconds', '0', '--e2e', '0', '--extras', '0', '--matrix', '0', '--block-sums', '0', '--scan-carries', '0'] + EXTRA
CALIB = [op.join(ROOT, 'tools', 'micro', '_build', 'fetch_calib')]


def rows_of(counter, command, tag):
    d = op.join(OUT, 'pmc_ct_%s_%s' % (tag, counter))
    cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', 'pmc', '--'] + command
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), timeout=300)
    out = []
    for f in glob.glob(op.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') == counter:
                    out.append((row.get('Kernel_Name', ''), int(row['Grid_Size']), float(row['Counter_Value'])))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    GIB = float(1 << 30)
    calib = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        for name, _, val in rows_of(counter, CALIB, 'calib'):
            if 'calib_' in name:
                calib.setdefault(counter, {})[name.split('(')[0].replace('void ', '')] = val
    # bytes per reported KB for each access width (1.0 would mean the counter is exact)
    fkb = {k: GIB / (v * 1024) for k, v in calib.get('FETCH_SIZE', {}).items() if 'read' in k and v > 0}
    wkb = {k: GIB / (v * 1024) for k, v in calib.get('WRITE_SIZE', {}).items() if 'write' in k and v > 0}
    f8 = next((v for k, v in fkb.items() if 'uint2' in k or 'HIP_vector_type<unsigned int, 2' in k), None)
    f16 = next((v for k, v in fkb.items() if 'uint4' in k or 'HIP_vector_type<unsigned int, 4' in k), None)
    w8 = next((v for k, v in wkb.items() if 'uint2' in k or 'HIP_vector_type<unsigned int, 2' in k), None)
    fetch = rows_of('FETCH_SIZE', BENCH, 'bench')
    write = rows_of('WRITE_SIZE', BENCH, 'bench')

    def per_kernel(rows):
        acc = {}
        for name, grid, val in rows:
            k = name.split('(')[0].replace('void ', '')
            a = acc.setdefault(k, {'launches': 0, 'sum_KB': 0.0, 'max_KB': 0.0})
            a['launches'] += 1
            a['sum_KB'] += val
            a['max_KB'] = max(a['max_KB'], val)
        return acc
    fk, wk = per_kernel(fetch), per_kernel(write)
    cost_f = {k: v for k, v in fk.items() if k.startswith('k_cost')}
    cost_w = {k: v for k, v in wk.items() if k.startswith('k_cost')}
    main_k = max(cost_f, key=lambda k: cost_f[k]['max_KB'])
    r = subprocess.run(BENCH, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
    bl = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    pairs = bl['roofline']['pairs_per_step']
    samples = int(bl['config']['workload'].split(' x ')[1].split(' ')[0])
    sites = int(bl['config']['workload'].split(' ')[1])
    ff = f8 or 2.0
    wf = w8 or 1.0
    read_b = cost_f[main_k]['max_KB'] * 1024 * ff
    write_b = cost_w.get(main_k, {'max_KB': 0.0})['max_KB'] * 1024 * wf
    sys.path.insert(0, ROOT)
    from wgbs_tools_amd import build
    rec = {'kernel': main_k, 'csrc_sha': build.source_hash(), 'workload': bl['config']['workload'],
           'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), gfx950, ROCm 7.2, over `python bench.py --steps 1 --warmup 0 --cpu-seconds 0 '
                     '--e2e 0 --extras 0 --matrix 0 --block-sums 0` + the arguments after --; the k_cost dispatch with the most bytes (the main batch: every chunk of the genome)',
           'calibration': {'what': 'tools/micro/fetch_calib under the same two passes: 1 GiB read / written once per kernel; bytes per reported KB (1024 would be exact)',
                           'FETCH_SIZE_KB': calib.get('FETCH_SIZE'), 'WRITE_SIZE_KB': calib.get('WRITE_SIZE'),
                           'read_factor_4_8_16B_per_lane': fkb, 'write_factor': wkb, 'read_factor_used': ff, 'write_factor_used': wf,
                           'guide': 'MI355X_MICROARCH.md: FETCH_SIZE reports exactly 1/2 of the bytes of a 16 B per lane streaming read on gfx950; other widths uncalibrated -> calibrated here'},
           'FETCH_SIZE_KB': cost_f[main_k]['max_KB'], 'WRITE_SIZE_KB': cost_w.get(main_k, {'max_KB': 0.0})['max_KB'],
           'read_bytes': read_b, 'write_bytes': write_b, 'traffic_bytes': read_b + write_b,
           # what the kernel must move: every beta byte of the chunks once (2 B x samples x sites) in, one double per scored block out
           'algorithmic_bytes': {'beta_bytes_read_once': 2.0 * samples * sites, 'scored_blocks_written': 8.0 * pairs, 'sum': 2.0 * samples * sites + 8.0 * pairs},
           'all_k_cost_dispatches': {'FETCH_SIZE': cost_f, 'WRITE_SIZE': cost_w}}
    rec['traffic_over_algorithmic'] = rec['traffic_bytes'] / rec['algorithmic_bytes']['sum']
    rec['read_over_beta_bytes'] = read_b / rec['algorithmic_bytes']['beta_bytes_read_once']
    json.dump(rec, open(op.join(OUT, 'cost_traffic.json'), 'w'), indent=1)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
