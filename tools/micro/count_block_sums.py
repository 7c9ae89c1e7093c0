#!/usr/bin/env python3
"""Instruction mix of k_block_sums_run<MODE> in the gfx950 ISA (no GPU needed): the kernel is straight-line per tile (8 tiles
unrolled), so totals / 8 = per tile and wavefront.
    python tools/micro/count_block_sums.py [MODE]
Counts by class (VALU, of which v_perm / adds / DPP; LDS reads and writes; vector memory; scalar) for the whole kernel and for the
stretch between two wave barriers that holds a `stage` (the per-site work) and the one that holds a `resolve` (the per-block work)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else '1'
out = os.path.join(ROOT, 'tools', 'micro', '_build', 'w.s')
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-S', '--cuda-device-only',
                       os.path.join(ROOT, 'wgbs_tools_amd', 'csrc', 'wgbsseg.hip'), '-o', out], stderr=subprocess.DEVNULL)
lines = open(out).read().split('\n')
name = '_Z16k_block_sums_runILi%sEE' % mode
s = [i for i, l in enumerate(lines) if l.startswith(name) and ':' in l][0]
e = [i for i in range(s, len(lines)) if lines[i].strip().startswith('.Lfunc_end')][0]
body = [x.strip() for x in lines[s:e] if x.strip() and not x.strip().startswith(('.', ';', '_Z'))]
for l in lines[e:e + 80]:
    m = re.search(r'; (NumVgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize): (\d+)', l)
    if m: print(m.group(1), m.group(2))


def mix(seg):
    c = dict(total=len(seg), valu=0, perm=0, dpp=0, lds_r=0, lds_w=0, vmem=0, salu=0, readlane=0)
    for x in seg:
        op = x.split()[0]
        if op.startswith('v_'):
            c['valu'] += 1
            if op.startswith('v_perm'): c['perm'] += 1
            if 'dpp' in x or 'row_' in x: c['dpp'] += 1
            if op.startswith('v_readlane') or op.startswith('v_readfirstlane'): c['readlane'] += 1
        elif op.startswith('ds_read') or op.startswith('ds_load'): c['lds_r'] += 1
        elif op.startswith('ds_write') or op.startswith('ds_store'): c['lds_w'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_')): c['vmem'] += 1
        elif op.startswith('s_'): c['salu'] += 1
    return c


print('whole kernel:', mix(body))
# straight-line stretches between wave barriers (s_barrier is not used; the wave barrier leaves no instruction: split at the
# ds_write_b128 groups instead: a stage = the stretch that ends with the tile's 4 ds_write_b128 + ds_write_b64)
idx = [i for i, x in enumerate(body) if x.startswith('ds_write_b128') or x.startswith('ds_store_b128')]
print('ds_write_b128 count:', len(idx), '(4 per tile)')
perm = sum(1 for x in body if x.startswith('v_perm'))
print('v_perm_b32 per tile: %.1f' % (perm / 8.0))
print('VALU per tile: %.1f, LDS reads per tile: %.1f, vector memory per tile: %.1f, scalar per tile: %.1f' % (
    mix(body)['valu'] / 8.0, mix(body)['lds_r'] / 8.0, mix(body)['vmem'] / 8.0, mix(body)['salu'] / 8.0))
