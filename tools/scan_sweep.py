#!/usr/bin/env python3
"""The scan pass alone (k_validate: a job without wide tiles), back-to-back launches, for a sweep of piece sizes:
    python tools/scan_sweep.py [--samples 32] [--sites 28217448] [--repeat 50] [--pieces 2048,4096,...]
One line per setting: ms per launch, GB/s, fraction of the 8 TB/s HBM peak.  Chunks = the reference's grid."""
import argparse
import ctypes as C
import os
import os.path as op
import sys

import numpy as np

ROOT = op.dirname(op.dirname(op.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from wgbs_tools_amd import _lib, parallel, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=32)
    ap.add_argument('--sites', type=int, default=28217448)
    ap.add_argument('--chunk', type=int, default=60000)
    ap.add_argument('--repeat', type=int, default=50)
    ap.add_argument('--pieces', default='2048,4096,8192,16384,32768,65536')
    args = ap.parse_args()
    names, sizes = synth.genome_shape(args.sites, 25 if args.sites >= 2500000 else 1)
    regions = parallel.regions_of_sizes([int(s) for s in sizes])
    grid = parallel.chunk_grid(regions, args.chunk)
    st = np.array([a - 1 for _, a, b in grid], dtype=np.int64)
    ln = np.array([b - a for _, a, b in grid], dtype=np.int32)
    S = _lib.load_synth()
    n = args.sites
    pitch = ((2 * n + 255) // 256) * 256 + 256
    b = torch.empty((args.samples, pitch), dtype=torch.uint8, device='cuda:0')
    assert S.wgbssynth_fill_betas_range(C.c_void_p(b.data_ptr()), pitch, 0, n, 0, args.samples, 20240601, 0) == 0
    for piece in [int(x) for x in args.pieces.split(',')]:
        os.environ['WGBSSEG_SCAN_PIECE_SITES'] = str(piece)
        seg = _lib.Segmenter(0)
        seg.set_betas_device(b.data_ptr(), args.samples, pitch, n, keepalive=b)
        best = None
        for _ in range(3):
            ms, nbytes = seg.scan_only(st, ln, repeat=args.repeat)
            best = ms if best is None else min(best, ms)
        print('piece %6d sites: %.4f ms per launch, %.0f GB/s, %.3f of peak (%d bytes)' % (piece, best, nbytes / best / 1e6, nbytes / best / 1e6 / 8000, nbytes), flush=True)
        seg.close()


if __name__ == '__main__':
    main()
