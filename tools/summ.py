#!/usr/bin/env python3
"""one compact line per bench.py JSON line found in the given log files"""
import json
import sys

for f in sys.argv[1:]:
    for line in open(f, errors='replace'):
        line = line.strip()
        if not line.startswith('{'):
            continue
        d = json.loads(line)
        dm = d['device_ms_per_step']
        bs = d.get('block_sums')
        print('%-34s %7.3f ms/step | scan %.3f (%.2f of peak) win %.3f cost %.3f (%.3g ev/s) dp %.3f trace %.3f line %.3f | stages %s%s' % (
            f.split('/')[-1], d['ms_per_step'], dm['scan_ms'], d['roofline']['frac'], dm['window_ms'], dm['cost_ms'],
            d['roofline_cost']['evals_per_s'], dm['dp_ms'], dm['trace_ms'], dm['total_ms'], d['roofline_cost']['stages'],
            '' if not bs else ' | block_sums %.3f ms %.2f of peak' % (bs['ms_bin_rows'], bs['frac_of_hbm_peak'])))
