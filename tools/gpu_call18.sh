#!/bin/bash
# SQ counters (one PMC pass each, 8 counters) of the scan pass with and without carries and of the recurrence kernel
set -u
REPO=$PWD; O=$REPO/gpurun_out/c18; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 280 python $REPO/tools/pmc_kernel.py --kernel "k_scan(" --out c18/pmc_scan_sq_islands.json -- --islands > $O/a.log 2>&1; echo "k_scan: rc $?"
timeout 280 python $REPO/tools/pmc_kernel.py --kernel "k_validate(" --out c18/pmc_validate_sq.json > $O/b.log 2>&1; echo "k_validate: rc $?"
timeout 280 python $REPO/tools/pmc_kernel.py --kernel "k_dp<7, 64, true>" --out c18/pmc_dp_sq.json > $O/c.log 2>&1; echo "k_dp: rc $?"
rm -rf $REPO/gpurun_out/pmc_k_*
cd $REPO; for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], 'rows', d['rows'], {k: round(v, 4) for k, v in d.get('frac_of_wave_cycles', {}).items()}, 'busy', d['counters'].get('SQ_BUSY_CYCLES'), 'valu insts', d['counters'].get('SQ_INSTS_VALU'))
PY
done
