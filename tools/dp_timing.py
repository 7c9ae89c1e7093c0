#!/usr/bin/env python3
"""Where does a step of the recurrence go, and WHERE does it run?  A -DWGBSSEG_DP_TIMING build of the library (made here on first use) lets
k_dp leave, per chunk, the cycles its recurrence wavefront and its first worker spent in the batch loop and how many of them at the barrier
(s_memtime ticks), and (round 5) the HW_ID of every wavefront by role + the XCC: which CU the workgroup ran on, which SIMD its recurrence
wavefront sat on, and whether another workgroup's recurrence shared that SIMD.  This script runs the chunk DPs of a stretch of the bench
genome (one batch: chunks only, no stitching) and prints the averages.

    python tools/dp_timing.py [n_chunks] [samples] [--flags "-D..."] [--tag name]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = sys.argv[1:]
flags, tag = [], 'dptiming'
if '--flags' in argv:
    i = argv.index('--flags'); flags = argv[i + 1].split(); del argv[i:i + 2]
if '--tag' in argv:
    i = argv.index('--tag'); tag = 'dptiming_' + argv[i + 1]; del argv[i:i + 2]
LIB = os.path.join(ROOT, 'tools', 'micro', '_build', 'libwgbsseg_%s.so' % tag)      # the library with -DWGBSSEG_DP_TIMING (the product build carries no counters)
SRC = os.path.join(ROOT, 'wgbs_tools_amd', 'csrc', 'wgbsseg.hip')
if not os.path.isfile(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(os.path.join(ROOT, 'wgbs_tools_amd', 'csrc', f)) for f in os.listdir(os.path.join(ROOT, 'wgbs_tools_amd', 'csrc')) if f.endswith(('.h', '.hip'))):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-pthread',
                           '-DWGBSSEG_DP_TIMING'] + flags + [SRC, '-o', LIB])
if '--build-only' in argv:
    sys.exit(0)
os.environ['WGBSSEG_LIB'] = LIB
os.environ['WGBSSEG_ALLOW_LIB_OVERRIDE'] = '1'
sys.path.insert(0, ROOT)
import numpy as np
from wgbs_tools_amd import _lib, synth
nch = int(argv[0]) if len(argv) > 0 else 480
ns = int(argv[1]) if len(argv) > 1 else 8
chunk = 60000
n = nch * chunk
loci = synth.synth_loci(20260926, [n])
import torch, ctypes as C
pitch = ((2 * n + 255) // 256) * 256 + 256
buf = torch.empty((ns, pitch), dtype=torch.uint8, device='cuda:0')
assert _lib.load_synth().wgbssynth_fill_betas(C.c_void_p(buf.data_ptr()), pitch, n, 0, ns, 20260926, None) == 0
print('== %s: %d chunks x %d samples, build flags %s' % (tag, nch, ns, ' '.join(['-DWGBSSEG_DP_TIMING'] + flags)))
with _lib.Segmenter(0) as sg:
    sg.set_betas_device(buf.data_ptr(), ns, pitch, n, keepalive=buf)
    sg.set_loci(loci)
    st = np.arange(nch, dtype=np.int64) * chunk
    for rep in range(3):
        sg.segment_chunks(st, [chunk] * nch, 15.0, 1000, 2000)
        tm = sg.timings()
        d = sg.debug_fetch('dpstate', np.float64, nch * 258)
        stride = d.size // nch
        full = d.reshape(nch, stride)
        d = full[:, :7]
        tot0, wait0, tot1, wait1, vm, commit, issue = d.mean(0)
        print('   worker 0 per batch of 64 steps: %.0f ticks, of which waiting for the rows loaded a batch ago %.0f, arranging them into LDS (incl. that wait) %.0f, issuing the next loads %.0f, barrier %.0f'
              % (tot1 / chunk * 64, vm / chunk * 64, commit / chunk * 64, issue / chunk * 64, wait1 / chunk * 64))
        print('chunks %d x %d samples: k_dp %.3f ms = %.1f ns/step | recurrence wavefront: %.0f ticks per step, %.1f %% at the barrier | worker 0: %.0f ticks per step, %.1f %% at the barrier | max/min over chunks of the recurrence loop: %.3f'
              % (nch, ns, tm['dp_ms'], tm['dp_ms'] * 1e6 / chunk, tot0 / chunk, 100 * wait0 / tot0, tot1 / chunk, 100 * wait1 / tot1, d[:, 0].max() / d[:, 0].min()), flush=True)
        if rep == 2:
            # placement: HW_ID by role (slot 8 = recurrence, 9.. = workers), XCC_ID in slot 16, loop begin / end ticks in 17 / 18
            hw = full[:, 8:16].astype(np.int64)
            xcc = full[:, 16].astype(np.int64) & 15
            simd = (hw >> 4) & 3
            cu = (xcc << 8) | ((hw[:, 0] >> 8) & 0xff)                 # (xcc, se, sh, cu) of the workgroup
            t_begin, t_end = full[:, 17], full[:, 18]
            rec_ticks = full[:, 0] / chunk                                 # ticks per step of the recurrence loop (barrier waits included)
            rec_busy = (full[:, 0] - full[:, 1]) / chunk                   # ... without them
            groups = {}
            for c in range(nch):
                groups.setdefault(int(cu[c]), []).append(c)
            per_cu = np.bincount([len(v) for v in groups.values()])
            print('   placement: %d CUs in use; workgroups per CU: %s' % (len(groups), {k: int(v) for k, v in enumerate(per_cu) if v}))
            print('   SIMD of the recurrence wavefront: %s; SIMDs of the roles of chunk 0: %s' % (np.bincount(simd[:, 0], minlength=4).tolist(), simd[0].tolist()))
            cls = {'alone on its CU': [], 'CU shared, recurrences on DIFFERENT SIMDs': [], 'CU shared, recurrences on the SAME SIMD': []}
            for key, v in groups.items():
                for c in v:
                    # co-resident = overlapping in time with another workgroup of the same CU for most of its life
                    others = [o for o in v if o != c and min(t_end[c], t_end[o]) - max(t_begin[c], t_begin[o]) > 0.5 * (t_end[c] - t_begin[c])]
                    if not others:
                        cls['alone on its CU'].append(c)
                    elif any(simd[o, 0] == simd[c, 0] for o in others):
                        cls['CU shared, recurrences on the SAME SIMD'].append(c)
                    else:
                        cls['CU shared, recurrences on DIFFERENT SIMDs'].append(c)
            for k, v in cls.items():
                if v:
                    print('   %-48s %4d chunks: %.1f ticks per step in the loop, %.1f without the barrier waits' % (k, len(v), rec_ticks[v].mean(), rec_busy[v].mean()))
            nw = int((hw[0] != 0).sum())
            waits = full[:, 20:20 + nw] / full[:, 0:1]                     # fraction of the loop each role spends at the barrier
            issue = full[:, 40:40 + nw] / chunk * 64                       # ticks per batch each worker spends issuing its row loads
            same = np.array([[simd[c, r] == simd[c, 0] for r in range(nw)] for c in range(nch)])
            print('   barrier wait by role (0 = recurrence, then the workers), mean over chunks: %s' % ' '.join('%.2f' % x for x in waits.mean(0)))
            print('   load-issue ticks per batch by role: %s' % ' '.join('%.0f' % x for x in issue.mean(0)))
            for r in range(1, nw):
                a, b = waits[same[:, r], r], waits[~same[:, r], r]
                if a.size and b.size:
                    print('   worker %d: on the recurrence wavefront\'s SIMD in %d chunks: wait %.2f, issue %.0f | elsewhere in %d: wait %.2f, issue %.0f'
                          % (r - 1, a.size, a.mean(), issue[same[:, r], r].mean(), b.size, b.mean(), issue[~same[:, r], r].mean()))
            life = (t_end - t_begin)
            print('   loop lifetime (ticks): min %.3g mean %.3g max %.3g; first begin .. last end %.3g' % (life.min(), life.mean(), life.max(), t_end.max() - t_begin.min()))
