#!/usr/bin/env python3
"""VGPRs / scratch / occupancy of the kernels of csrc/wgbsseg.hip whose (demangled) name matches a pattern, for a set of -D flags (no GPU needed):
    python tools/micro/kernel_regs.py 'k_cost' [-DWG_COST_ILP=2 ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pat = sys.argv[1] if len(sys.argv) > 1 else 'k_cost'
flags = sys.argv[2:]
out = os.path.join(ROOT, 'tools', 'micro', '_build', 'regs.s')
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-S', '--cuda-device-only'] + flags +
                      [os.path.join(ROOT, 'wgbs_tools_amd', 'csrc', 'wgbsseg.hip'), '-o', out], stderr=subprocess.DEVNULL)
cur, rows = None, {}
for l in open(out):
    m = re.match(r'^(_Z\w+):', l)
    if m:
        cur = m.group(1)
    m = re.search(r'; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte): (\d+)', l)
    if m and cur:
        rows.setdefault(cur, {})[m.group(1)] = int(m.group(2))
names = list(rows)
dem = subprocess.run(['c++filt'] + names, stdout=subprocess.PIPE, text=True).stdout.split('\n')
for n, d in zip(names, dem):
    if re.search(pat, d):
        r = rows[n]
        print('%-60s vgpr %3d  scratch %3d  occupancy %d  code %6d B' % (d.split('(')[0].replace('void ', ''), r.get('NumVgprs', -1), r.get('ScratchSize', -1), r.get('Occupancy', -1), r.get('codeLenInByte', -1)))
