#!/usr/bin/env python3
"""VALU / SALU / LDS instruction counts of the loops of k_cost<TI, FAST, SPLIT> in the gfx950 ISA (no GPU needed):
    python tools/micro/count_cost_loop.py [TI] [FAST] [SPLIT] [ONEG] [--show] [--all] [--json PATH]
compiles csrc/wgbsseg.hip to assembly (device only) and lists every loop of the kernel; the sample loop is the one with the
most fp64 instructions (four evaluations per trip, the rare exact path inside it behind s_cbranch_execz).

--json PATH: the instruction MIX of one (block, sample) evaluation on the common path of that loop — the walk follows every
`s_cbranch_execz` (the rare paths are skipped, as a wavefront whose lanes all stay on the common path does) — by issue class,
with the issue cycles per wavefront instruction of each class, written with the hash of the kernel sources
(wgbs_tools_amd/build.py source_hash) so that bench.py can tell whether the file describes the library it runs:
    fp32    plain fp32 and 32-bit integer VOP2 adds/subs (v_add_f32 v_mul_f32 v_fma_f32 v_sub_u32 ...)          2 cycles
            (MI355X_MICROARCH.md: "v_fma_f32 (wave64) 2 cyc"; profiles/r02_valu_rates.log measured the same rate for v_sub_u32)
    fp64    v_*_f64                                                                                               4 cycles
    cvt     conversions incl. SDWA forms                                                                          4 cycles
    int3    bit-field / shift-add / compare / other 32-bit integer and move instructions                          4 cycles
            (profiles/r02_valu_rates.log: v_lshl_add_u32, v_bfe_u32 at the fp64 rate; unmeasured ones are counted at 4)
    trans   v_rcp_f32 and the other transcendentals                                                               8 cycles
"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
args = [a for a in sys.argv[1:] if not a.startswith('-')]
jpath = sys.argv[sys.argv.index('--json') + 1] if '--json' in sys.argv else None
if jpath in args: args.remove(jpath)
ti = args[0] if len(args) > 0 else '64'
fast = args[1] if len(args) > 1 else '2'
split = args[2] if len(args) > 2 else '0'
out = os.path.join(ROOT, 'tools', 'micro', '_build', 'w.s')
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-S'] + [a for a in sys.argv[1:] if a.startswith('-D')] + [
                       '--cuda-device-only', os.path.join(ROOT, 'wgbs_tools_amd', 'csrc', 'wgbsseg.hip'), '-o', out],
                      stderr=subprocess.DEVNULL)
lines = open(out).read().split('\n')
oneg = args[3] if len(args) > 3 else '1'          # one sample group (k_cost<.., true>: the x32 bench's form)
name = '_Z6k_costILi%sELi%sELi%sELb%sEE' % (ti, fast, split, oneg)
s = [i for i, l in enumerate(lines) if l.startswith(name) and ':' in l][0]
e = [i for i in range(s, len(lines)) if lines[i].strip().startswith('.Lfunc_end')][0]
body = lines[s:e]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
meta = {}
for l in lines[e:e + 60]:
    m = re.search(r'; (NumVgprs|ScratchSize|Occupancy|NumSgprs): (\d+)', l)
    if m:
        meta[m.group(1)] = int(m.group(2))
        print(m.group(1), m.group(2))
loops = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))


def instr(x):
    x = x.strip()
    return x if x and not x.startswith(('.', ';')) and not x.endswith(':') else None


best = None
for a, b in loops:
    seg = [instr(x) for x in body[a:b + 1]]
    seg = [x for x in seg if x]
    v = sum(1 for x in seg if x.startswith('v_'))
    f64 = sum(1 for x in seg if re.match(r'v_\w+_f64', x))
    if 20 <= f64 <= 40 or '--all' in sys.argv:
        print('loop lines %d-%d: %d instructions, VALU %d (fp64 %d), SALU %d, LDS %d' % (
            a, b, len(seg), v, f64, sum(1 for x in seg if x.startswith('s_')), sum(1 for x in seg if x.startswith('ds_'))))
        if '--show' in sys.argv:
            print('\n'.join(x for x in body[a:b + 1] if x.strip() and not x.strip().startswith(';')))
    # the innermost loop with the most fp64 work = the sample loop (its trips hold EV evaluations)
    if best is None or f64 > best[2] or (f64 == best[2] and b - a < best[1] - best[0]):
        if not any(a <= a2 and b2 <= b and (a2, b2) != (a, b) and sum(1 for x in body[a2:b2 + 1] if re.match(r'\s*v_\w+_f64', x)) >= 20 for a2, b2 in loops):
            best = (a, b, f64)


def klass(op):
    if re.match(r'v_(rcp|rsq|sqrt|log|exp|sin|cos)_', op): return 'trans'
    if re.match(r'v_cvt_', op): return 'cvt'
    if re.match(r'v_\w+_f64', op): return 'fp64'
    if re.match(r'v_(add|sub|subrev|mul|fma|fmac|mac|mad|max|min)_f32', op): return 'fp32'
    if re.match(r'v_(add|sub|subrev)_(u32|i32|co_u32)', op): return 'fp32'
    return 'int3'


CYC = {'fp32': 2, 'fp64': 4, 'cvt': 4, 'int3': 4, 'trans': 8}
if jpath and best:
    a, b, _ = best
    i, mix, salu, lds, nops, walked = a, {k: 0 for k in CYC}, 0, 0, 0, 0
    ops = {}
    while i <= b and walked < 5000:
        x = instr(body[i]); walked += 1
        if not x: i += 1; continue
        op = x.split()[0]
        m = re.match(r's_cbranch_execz\s+(\.LBB\d+_\d+)', x)
        if op.startswith('v_'):
            k = klass(op); mix[k] += 1; ops[op] = ops.get(op, 0) + 1
        elif op.startswith('ds_'): lds += 1
        elif op == 's_nop': nops += 1
        elif op.startswith('s_'): salu += 1
        if m and labels.get(m.group(1), -1) > i and labels[m.group(1)] <= b:
            i = labels[m.group(1)]                      # the common path: no lane needs the block behind the branch
        else:
            i += 1
    ev = sum(1 for x in body[a:b + 1] if re.match(r'\s*v_rcp_f32', x))       # one division per evaluation
    from wgbs_tools_amd import build
    rec = {'kernel': 'k_cost<%s,%s,%s>' % (ti, fast, split), 'csrc_sha': build.source_hash(), 'evaluations_per_trip': ev,
           'valu_per_eval': sum(mix.values()) / ev, 'mix_per_eval': {k: v / ev for k, v in mix.items()},
           'cycles_per_class': CYC, 'issue_cycles_per_eval': sum(CYC[k] * v for k, v in mix.items()) / ev,
           'salu_per_eval': salu / ev, 's_nop_per_eval': nops / ev, 'lds_per_eval': lds / ev, 'ops_per_trip': ops,
           'vgprs': meta.get('NumVgprs'), 'occupancy_waves_per_simd': meta.get('Occupancy'),
           'how': 'tools/micro/count_cost_loop.py %s %s %s --json: common path of the sample loop (s_cbranch_execz followed)' % (ti, fast, split)}
    json.dump(rec, open(jpath, 'w'), indent=1)
    print(json.dumps({k: rec[k] for k in ('kernel', 'evaluations_per_trip', 'valu_per_eval', 'mix_per_eval', 'issue_cycles_per_eval', 'salu_per_eval', 'lds_per_eval')}))
