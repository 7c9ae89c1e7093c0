#!/usr/bin/env python3
"""one compact line per bench.py JSON line found in the given log files (round-3 layout: roofline = k_cost, roofline_scan)"""
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
        rc = d.get('roofline_cost') or d['roofline']
        rs = d.get('roofline_scan') or d['roofline']
        print('%-34s %7.3f ms/step | scan %.3f (%.2f of peak) win %.3f cost %.3f (%.3g ev/s) dp %.3f trace %.3f line %.3f | stages %s%s' % (
            f.split('/')[-1], d['ms_per_step'], dm['scan_ms'], rs['frac'], dm['window_ms'], dm['cost_ms'],
            rc['evals_per_s'], dm['dp_ms'], dm['trace_ms'], dm['total_ms'], rc['stages'],
            '' if not bs else ' | block_sums %.3f ms %.2f of peak' % (bs['ms_bin_rows'], bs['frac_of_hbm_peak'])))
        sc = d.get('roofline_scan_carries')
        if sc:
            print('    k_scan with carries: %.3f ms, %.2f of peak, traffic %s' % (sc['avg_launch_ms'], sc['frac'], sc['traffic_over_algorithmic']))
        for r in (d.get('matrix') or {}).get('rows', []):
            if 'shares_on_this_gpu' in r and 'failed' not in r:
                print('    x%-4d as %d shares on this GPU: %8.3f ms/step = %.3f ms per share, borders equal: %s' % (r['samples'], r['shares_on_this_gpu'], r['ms_per_step'], r['ms_per_share'], r['borders_equal_one_context']))
                continue
            if 'failed' in r:
                print('    x%-4d failed: %s' % (r['samples'], r['failed']))
            else:
                print('    x%-4d%s %8.3f ms/step  %.3g sites/s | cost %.3f (%.3g ev/s) dp %.3f scan %.3f (%.2f of peak)' % (
                    r['samples'], ' islands' if r.get('islands') else '', r['ms_per_step'], r['value'], r['cost_ms'], r['evals_per_s'], r['dp_ms'], r['scan_ms'], r['scan_frac_of_hbm_peak']))
