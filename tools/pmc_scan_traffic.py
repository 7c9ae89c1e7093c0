#!/usr/bin/env python3
"""HBM traffic of the dominant scan-pass launch (k_validate; k_scan when the job has wide tiles) from the PMC counters, the way MI355X_MICROARCH.md prescribes: FETCH_SIZE
and WRITE_SIZE in SEPARATE rocprofv3 passes (with --kernel-trace only), units KB = 1024 B, FETCH_SIZE doubled on gfx950
for wide coalesced streaming reads.  Runs on the GPU box; writes gpurun_out/scan_traffic.json (copy it to profiles/).

    cd /tmp && TMPDIR=/tmp python $REPO/tools/pmc_scan_traffic.py
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
FORCED = '--forced-carries' in sys.argv[1:]     # the prefix-sum pass with carries forced on the workload's chunk grid (bench.py: roofline_scan_carries)
EXTRA = [a for a in sys.argv[1:] if a != '--forced-carries'] + ['--scan-carries', '1' if FORCED else '0']            # further bench.py arguments (e.g. --islands: then the step's own pass is k_scan with its carries)


def one_pass(counter):
    d = op.join(OUT, 'pmc_' + counter)
    cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', 'pmc', '--',
           sys.executable, op.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0', '--cpu-seconds', '0', '--e2e', '0', '--extras', '0', '--matrix', '0', '--block-sums', '0'] + EXTRA
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'),
                   timeout=240)
    best = None
    for f in glob.glob(op.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get('Kernel_Name', '')
                if row.get('Counter_Name') != counter or not (name.startswith('k_scan') or (name.startswith('k_validate') and not FORCED)):
                    continue
                grid = int(row['Grid_Size'])
                val = float(row['Counter_Value'])
                # the launch that did the work: k_scan leaves at once (0 bytes) when the job has no wide units and vice versa
                if best is None or val > best[1]:
                    best = (grid, val, name.split('(')[0])
    assert best, 'no scan-pass dispatch with ' + counter
    return best


def main():
    os.makedirs(OUT, exist_ok=True)
    gf, fetch, kname = one_pass('FETCH_SIZE')
    try:
        gw, write, _ = one_pass('WRITE_SIZE')
    except AssertionError:
        gw, write = gf, 0.0
    # the exact algorithmic bytes of that launch (2 bytes x samples x sites of the main batch: chunks + upfront patches): bench.py reports it
    r = subprocess.run([sys.executable, op.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0', '--cpu-seconds', '0', '--e2e', '0', '--extras', '0', '--matrix', '0', '--block-sums', '0'] + EXTRA,
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=240)
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    bl = json.loads(line)
    alg = int(bl['roofline_scan_carries' if FORCED else 'roofline_scan']['algorithmic_bytes_per_launch'])
    sys.path.insert(0, ROOT)
    from wgbs_tools_amd import build
    rec = {'kernel': kname, 'csrc_sha': build.source_hash(), 'workload': bl['config']['workload'] + ': main batch (chunks + upfront patches; the patches lie inside the chunks and are not read again)',
           'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), gfx950, ROCm 7.2; the scan-pass dispatch with the most bytes of `python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --e2e 0 --extras 0 --matrix 0 --block-sums 0` + the arguments after --',
           'grid_size': gf, 'FETCH_SIZE_KB': fetch, 'WRITE_SIZE_KB': write,
           'correction': 'MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) -> doubled; WRITE_SIZE taken as is; units KB = 1024 B',
           'read_bytes': 2 * fetch * 1024, 'write_bytes': write * 1024, 'traffic_bytes': 2 * fetch * 1024 + write * 1024,
           'algorithmic_bytes': alg, 'traffic_over_algorithmic': (2 * fetch * 1024 + write * 1024) / alg, 'forced_carries': FORCED}
    if FORCED:
        rec['workload'] = bl['config']['workload'] + ': the chunk grid, carries forced (wgbsseg_scan_only want_carry = 1; the dispatch with the most bytes of 22)'
        rec['carry_bytes_written'] = bl['roofline_scan_carries']['carry_bytes_written_per_launch']
    json.dump(rec, open(op.join(OUT, 'scan_traffic%s.json' % ('_carries' if FORCED else '_islands' if '--islands' in EXTRA else '')), 'w'), indent=1)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
