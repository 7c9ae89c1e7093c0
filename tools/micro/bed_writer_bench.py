#!/usr/bin/env python3
"""The BED writer alone (wgbsseg_add_loci: CpG-index blocks -> `chr start end startCpG endCpG` rows), on a block table of
whole-genome size, host only:   python tools/micro/bed_writer_bench.py [--blocks 2800000] [--threads 0,4,8,16,32,64]"""
import argparse, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wgbs_tools_amd import _lib, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--blocks', type=int, default=2800000)
    ap.add_argument('--sites', type=int, default=28217448)
    ap.add_argument('--threads', default='0')
    ap.add_argument('--dir', default=None)
    a = ap.parse_args()
    names, sizes = synth.genome_shape(a.sites, 25)
    loci = np.concatenate([synth.synth_loci(20240601 + i, [int(s)]) for i, s in enumerate(sizes)]).astype(np.uint32)
    cum = np.cumsum(sizes)
    rng = np.random.default_rng(1)
    cuts = np.unique(np.concatenate([rng.integers(1, a.sites + 1, a.blocks), cum[:-1] + 1, [1, a.sites + 1]]))
    s, e = cuts[:-1].astype(np.int64), cuts[1:].astype(np.int64)
    ok = np.ones(s.size, bool)                       # no block across a chromosome border
    for c in cum[:-1]:
        ok &= ~((s <= c) & (e > c + 1))
    s, e = s[ok], e[ok]
    td = tempfile.mkdtemp(dir=a.dir)
    out = os.path.join(td, 'blocks.bed')
    for th in [int(x) for x in a.threads.split(',')]:
        best = None
        for _ in range(4):
            t0 = time.perf_counter()
            _lib.add_loci(loci, list(names), cum, s, e, out, threads=th)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        sz = os.path.getsize(out)
        print('threads %s: %d rows, %.1f MB, best of 4: %.1f ms (%.2f GB/s)' % (th or 'default', s.size, sz / 1e6, best * 1e3, sz / best / 1e9), flush=True)
    os.unlink(out); os.rmdir(td)


if __name__ == '__main__':
    main()
