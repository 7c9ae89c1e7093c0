#!/usr/bin/env python3
"""VALU / SALU / LDS instruction counts of the loops of k_cost<TI, FAST, SPLIT> in the gfx950 ISA (no GPU needed):
    python tools/micro/count_cost_loop.py [TI] [FAST] [SPLIT] [--show]
compiles csrc/wgbsseg.hip to assembly (device only) and lists every loop of the kernel; the sample loop is the one
with ~25 fp64 instructions.  Used to keep an eye on the instruction count per (block, sample) evaluation."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ti = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].isdigit() else '64'
fast = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2].isdigit() else '2'
split = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3].isdigit() else '0'
out = os.path.join(ROOT, 'tools', 'micro', '_build', 'w.s')
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-S',
                       '--cuda-device-only', os.path.join(ROOT, 'wgbs_tools_amd', 'csrc', 'wgbsseg.hip'), '-o', out],
                      stderr=subprocess.DEVNULL)
lines = open(out).read().split('\n')
name = '_Z6k_costILi%sELi%sELi%sEE' % (ti, fast, split)
s = [i for i, l in enumerate(lines) if l.startswith(name) and l.rstrip().endswith(':') or (l.startswith(name) and ':' in l)][0]
e = [i for i in range(s, len(lines)) if lines[i].strip().startswith('.Lfunc_end')][0]
body = lines[s:e]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
for l in body:
    m = re.search(r'; (NumVgprs|ScratchSize|Occupancy): (\d+)', l)
for l in lines[e:e + 60]:
    m = re.search(r'; (NumVgprs|ScratchSize|Occupancy|NumSgprs): (\d+)', l)
    if m: print(m.group(1), m.group(2))
loops = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
for a, b in loops:
    seg = [x.strip() for x in body[a:b + 1] if x.strip() and not x.strip().startswith(('.', ';'))]
    v = sum(1 for x in seg if x.startswith('v_'))
    f64 = sum(1 for x in seg if re.match(r'v_\w+_f64', x))
    if 20 <= f64 <= 40 or '--all' in sys.argv:
        print('loop lines %d-%d: %d instructions, VALU %d (fp64 %d), SALU %d, LDS %d' % (
            a, b, len(seg), v, f64, sum(1 for x in seg if x.startswith('s_')), sum(1 for x in seg if x.startswith('ds_'))))
        if '--show' in sys.argv:
            print('\n'.join(x for x in body[a:b + 1] if x.strip() and not x.strip().startswith(';')))
