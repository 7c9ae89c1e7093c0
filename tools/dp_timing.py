#!/usr/bin/env python3
"""Where does a step of the recurrence go?  A -DWGBSSEG_DP_TIMING build of the library (made here on first use) lets k_dp leave, per chunk, the cycles its recurrence wavefront and its
first worker spent in the batch loop and how many of them at the barrier (s_memtime ticks); this script runs the chunk DPs of a stretch
of the bench genome (one batch: chunks only, no stitching) and prints the averages.      python tools/dp_timing.py [n_chunks] [samples]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'tools', 'micro', '_build', 'libwgbsseg_dptiming.so')      # the library with -DWGBSSEG_DP_TIMING (the product build carries no counters)
SRC = os.path.join(ROOT, 'wgbs_tools_amd', 'csrc', 'wgbsseg.hip')
if not os.path.isfile(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(os.path.join(ROOT, 'wgbs_tools_amd', 'csrc', f)) for f in os.listdir(os.path.join(ROOT, 'wgbs_tools_amd', 'csrc')) if f.endswith(('.h', '.hip'))):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-pthread',
                           '-DWGBSSEG_DP_TIMING', SRC, '-o', LIB])
os.environ['WGBSSEG_LIB'] = LIB
os.environ['WGBSSEG_ALLOW_LIB_OVERRIDE'] = '1'
sys.path.insert(0, ROOT)
import numpy as np
from wgbs_tools_amd import _lib, synth
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 480
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 8
chunk = 60000
n = nch * chunk
loci = synth.synth_loci(20260926, [n])
import torch, ctypes as C
pitch = ((2 * n + 255) // 256) * 256 + 256
buf = torch.empty((ns, pitch), dtype=torch.uint8, device='cuda:0')
assert _lib.load_synth().wgbssynth_fill_betas(C.c_void_p(buf.data_ptr()), pitch, n, 0, ns, 20260926, None) == 0
with _lib.Segmenter(0) as sg:
    sg.set_betas_device(buf.data_ptr(), ns, pitch, n, keepalive=buf)
    sg.set_loci(loci)
    st = np.arange(nch, dtype=np.int64) * chunk
    for rep in range(3):
        sg.segment_chunks(st, [chunk] * nch, 15.0, 1000, 2000)
        tm = sg.timings()
        d = sg.debug_fetch('dpstate', np.float64, nch * 258)
        stride = d.size // nch
        d = d.reshape(nch, stride)[:, :7]
        tot0, wait0, tot1, wait1, vm, commit, issue = d.mean(0)
        print('   worker 0 per batch of 64 steps: %.0f ticks, of which waiting for the rows loaded a batch ago %.0f, arranging them into LDS (incl. that wait) %.0f, issuing the next loads %.0f, barrier %.0f'
              % (tot1 / chunk * 64, vm / chunk * 64, commit / chunk * 64, issue / chunk * 64, wait1 / chunk * 64))
        print('chunks %d x %d samples: k_dp %.3f ms = %.1f ns/step | recurrence wavefront: %.0f ticks per step, %.1f %% at the barrier | worker 0: %.0f ticks per step, %.1f %% at the barrier | max/min over chunks of the recurrence loop: %.3f'
              % (nch, ns, tm['dp_ms'], tm['dp_ms'] * 1e6 / chunk, tot0 / chunk, 100 * wait0 / tot0, tot1 / chunk, 100 * wait1 / tot1, d[:, 0].max() / d[:, 0].min()), flush=True)
