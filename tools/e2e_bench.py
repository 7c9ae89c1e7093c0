#!/usr/bin/env python3
"""End-to-end timing of `wgbstools segment` (page-cached .beta files -> BED), SURVEY.md §8(d)(ii).

Writes a synthetic hg19-shaped genome directory and N .beta files under --dir (default /tmp/wgbs_e2e), then runs the
CLI on them twice (the second run has every input in the page cache) and prints the phases.
    python tools/e2e_bench.py --samples 32
"""
import argparse
import gzip
import io
import contextlib
import os
import os.path as op
import sys
import time

import numpy as np

ROOT = op.dirname(op.dirname(op.abspath(__file__)))
sys.path.insert(0, ROOT)
from wgbs_tools_amd import synth, wgbs_tools          # noqa: E402

SEED = 20260926


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=32)
    ap.add_argument('--sites', type=int, default=28217448)
    ap.add_argument('--dir', default='/tmp/wgbs_e2e')
    ap.add_argument('--keep', action='store_true')
    args = ap.parse_args()
    ref = op.join(args.dir, 'references', 'synth')
    os.makedirs(ref, exist_ok=True)
    names, sizes = synth.genome_shape(args.sites, 25)
    sizes = [int(s) for s in sizes]
    t0 = time.perf_counter()
    loci = synth.synth_loci(SEED, sizes)
    with open(op.join(ref, 'CpG.chrome.size'), 'w') as f:
        for c, s in zip(names, sizes):
            f.write('%s\t%d\n' % (c, s))
    with open(op.join(ref, 'chrome.size'), 'w') as f:
        pos = 0
        for c, s in zip(names, sizes):
            f.write('%s\t%d\n' % (c, int(loci[pos + s - 1]) + 10000)); pos += s
    # the CpG dictionary itself (28 M text lines) is only ever read to build loci.u32: a stub stands in for it here
    with gzip.open(op.join(ref, 'CpG.bed.gz'), 'wb') as f:
        f.write(b'')
    if not op.lexists(op.join(ref, 'rev.CpG.bed.gz')):
        os.symlink('CpG.bed.gz', op.join(ref, 'rev.CpG.bed.gz'))
    loci.tofile(op.join(ref, 'loci.u32'))
    os.utime(op.join(ref, 'loci.u32'))
    paths = [op.join(args.dir, 's%03d.beta' % s) for s in range(args.samples)]
    if not all(op.isfile(p) and op.getsize(p) == 2 * args.sites for p in paths):
        import ctypes as C
        import torch
        from wgbs_tools_amd import _lib
        pitch = ((2 * args.sites + 255) // 256) * 256 + 256
        S = _lib.load_synth()
        buf = torch.empty((1, pitch), dtype=torch.uint8, device='cuda:0')
        for s, p in enumerate(paths):                      # the device twin of synth_betas (bit-identical, tests/test_gpu_parity.py)
            rc = S.wgbssynth_fill_betas(C.c_void_p(buf.data_ptr() - s * pitch), pitch, args.sites, s, 1, SEED, None)   # row s of a virtual [*][pitch] buffer
            assert rc == 0
            buf[0, :2 * args.sites].cpu().numpy().tofile(p)
        del buf
    print('inputs ready in %.1f s (%d files, %.2f GB)' % (time.perf_counter() - t0, len(paths), 2e-9 * args.sites * args.samples), flush=True)
    os.environ['WGBSSEG_PROFILE'] = '1'
    out = op.join(args.dir, 'blocks.bed')
    for rep in range(3):
        if op.exists(out):
            os.remove(out)            # (a user's run writes a new file; truncating the previous 115 MB costs more than writing it)
        err = io.StringIO()
        t0 = time.perf_counter()
        with contextlib.redirect_stderr(err):
            rc = wgbs_tools.main(['wgbstools', 'segment', '--betas'] + paths + ['--genome', ref, '-o', out])
        dt = time.perf_counter() - t0
        lines = [l for l in err.getvalue().splitlines() if 'phases' in l or 'found' in l or 'betas to the device' in l]
        print('run %d: rc %s, %.3f s wall, %.3g CpG-sites/s end to end; %s' % (rep, rc, dt, args.sites / dt, ' | '.join(lines)), flush=True)
    import hashlib
    print('BED: %d rows, %.1f MB, md5 %s' % (sum(1 for _ in open(out)), op.getsize(out) / 1e6, hashlib.md5(open(out, 'rb').read()).hexdigest()))
    if not args.keep:
        for p in paths:
            os.remove(p)


if __name__ == '__main__':
    main()
