#!/usr/bin/env python3
"""The whole synthetic hg19 genome x N betas against the REFERENCE BINARY (oracle/_ref/segmentor), on the library that is in the tree.

north_star: "bit-exact block boundaries at 28M CpGs x 200 betas".  Round 6: the suite's own x200 test runs this check in full
(tests/fullref.py::whole_genome_vs_reference, shared with this tool); the tool is for other cohorts, chromosomes and the adversarial world:

  1. wgbsseg_segment_regions over the 25 chromosomes (the product call);
  2. per chromosome: every chunk of the reference's grid through wgbsseg_segment_chunks, then the reference's pairwise stitching tree
     (tests/reftree.py, pinned by vectors of the reference's own driver) over those chunk DPs, with every patch it asks for computed by the
     HIP path — the stitched list must equal (1);
  3. EVERY range the 25 trees touched (all chunks + all junction patches) through the reference binary, one single-threaded process per range
     on all host cores, compared border by border with what the HIP path returned for that range.

Prints one summary line per stage and a final JSON line; exit code 1 on any difference.  Test infrastructure: the oracle is the checker here.

    python tools/full_vs_reference.py [--samples 200] [--procs N] [--chromosomes 0-24]
"""
import argparse
import json
import os
import os.path as op
import sys
import time

import numpy as np

ROOT = op.dirname(op.dirname(op.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, op.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=200)
    ap.add_argument('--sites', type=int, default=0, help='0: hg19 (28,217,448)')
    ap.add_argument('--chunk', type=int, default=60000)
    ap.add_argument('--pcount', type=float, default=15.0)
    ap.add_argument('--max-cpg', type=int, default=1000)
    ap.add_argument('--max-bp', type=int, default=2000)
    ap.add_argument('--procs', type=int, default=0, help='reference processes at a time (0: all logical CPUs)')
    ap.add_argument('--chromosomes', default='', help='e.g. 0-24 or 20,21 (default: all)')
    ap.add_argument('--adversarial', action='store_true',
                    help='another world at full size: loci with CpG islands (windows of hundreds of sites: medium and wide tiles, the 32-step recurrence, carries), a few '
                         'runs of EQUAL positions and backward steps (the plain path on whole chunks), and in every sample thousands of stretches of zero coverage, '
                         'saturated counts (255, 255), (0, 255), (127, 255) and single reads (1, 1) — the shapes the small fuzz worlds hold, at the size of the genome')
    args = ap.parse_args()

    import fullref                                     # the check itself: the same function the suite's x200 test runs
    import test_gpu_fullsize as F                      # the suite's helpers: device genome, whole-genome call, properties
    from oracle import oracle
    from wgbs_tools_amd import _lib, synth, build as nbuild
    import torch

    sites = args.sites or synth.HG19_NR_SITES
    N, chunk, pc, mb = args.samples, args.chunk, args.pcount, args.max_bp
    mc = min(args.max_cpg, mb // 2)                    # segment.py:65
    names, sizes = synth.genome_shape(sites, 25 if sites >= 2500000 else max(1, min(25, sites // 100000)))
    sizes = [int(s) for s in sizes]
    loci = synth.synth_loci(F.SEED, sizes)
    regions, pos = [], 1
    for s in sizes:
        regions.append((pos, pos + s))
        pos += s
    which = list(range(len(regions)))
    if args.chromosomes:
        which = []
        for part in args.chromosomes.split(','):
            a, _, b = part.partition('-')
            which += list(range(int(a), int(b or a) + 1))
    if args.adversarial:
        loci = synth.synth_loci(F.SEED, sizes, islands=True)
        rng = np.random.default_rng(20260927)
        L = loci.astype(np.int64)
        edges = np.concatenate([[0], np.cumsum(sizes)])
        n_disorder = 0
        for c in rng.choice(len(sizes) - 1, 8, replace=False):          # eight places, each inside one chromosome
            p0 = int(rng.integers(edges[c] + 1000, edges[c + 1] - 1000))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                k = int(rng.integers(1, 6)); L[p0 + 1:p0 + 1 + k] = L[p0]            # a run of equal positions
            elif kind == 1:
                L[p0 + 1] = L[p0] - int(rng.integers(1, 50))                          # one step backwards
            else:
                k = int(rng.integers(3, 30)); L[p0:p0 + k] = L[p0:p0 + k][::-1].copy()   # a descending run
            n_disorder += 1
        loci = L.astype(np.uint32)
    sha = nbuild.source_hash()
    print('library csrc_sha %s, ABI %d; %d CpGs x %d betas, chunk %d, max_cpg %d, max_bp %d, pcount %g; reference binary: %s'
          % (sha, _lib.load().wgbsseg_version(), sites, N, chunk, mc, mb, pc, oracle.REF_BIN), flush=True)
    assert oracle.have_ref(), 'oracle/_ref/segmentor is missing'
    t0 = time.time()
    buf, pitch = F._device_genome(sites, N)
    if args.adversarial:
        kinds = torch.tensor([[0, 0], [255, 255], [0, 255], [127, 255], [1, 1], [255, 255]], dtype=torch.uint8, device=buf.device)
        n_st = 0
        for smp in range(N):
            r = np.random.default_rng(1000 + smp)
            K = 1500
            a = r.integers(0, sites - 1, K)
            ln = np.minimum(np.where(r.random(K) < 0.8, r.integers(1, 60, K), r.integers(60, 4000, K)), sites - a)
            kd = r.integers(0, len(kinds), K)
            row = buf[smp, :2 * sites].view(sites, 2)
            for x, l, k in zip(a.tolist(), ln.tolist(), kd.tolist()):
                row[x:x + l] = kinds[k]
            n_st += K
        torch.cuda.synchronize()
        print('adversarial world: loci with CpG islands, %d places with equal / backward positions, %d stretches of zero / saturated / single-read counts over the %d samples'
              % (n_disorder, n_st, N), flush=True)
    bad = 0
    with _lib.Segmenter(0) as seg:
        seg.set_betas_device(buf.data_ptr(), N, pitch, sites, keepalive=buf)
        seg.set_loci(loci)
        res, stats = F._run_whole(seg, regions, chunk, pc, mc, mb)
        F._check_properties(res, regions, loci, mc, mb)
        n_blocks = int(sum(len(r) - 1 for r in res))
        print('[1] whole-genome call: %d chunks, %d blocks, stitch stats %s (%.1f s since start)'
              % (stats['chunks'], n_blocks, {k: int(v) for k, v in stats.items()}, time.time() - t0), flush=True)
        # 2. + 3.: the reference's tree over the HIP path's chunk / patch DPs, every range through the reference binary (tests/fullref.py)
        chk = fullref.whole_genome_vs_reference(seg, buf, loci, regions, chunk, pc, mc, mb, res, which=which, names=names,
                                                procs=args.procs, log=lambda *a: print(*a, flush=True))
        for line in chk['different']:
            print('    DIFFERENT: ' + line, flush=True)
        bad = chk['differences']
    out = {'csrc_sha': sha, 'sites': sites, 'samples': N, 'chunk': chunk, 'max_cpg': mc, 'max_bp': mb, 'pcount': pc, 'blocks': n_blocks,
           'wall_s': time.time() - t0, 'device': torch.cuda.get_device_name(0), 'adversarial': bool(args.adversarial)}
    out.update(chk)
    print(json.dumps(out), flush=True)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
