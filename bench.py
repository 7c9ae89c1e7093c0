#!/usr/bin/env python3
"""bench.py — CpG-sites/sec segmented on synthetic hg19-shaped betas (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--samples 32] [--sites 28217448]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one whole `segment` job over the resident beta bytes: scan/validate pass, window extents, block
scoring, changepoint DP, traceback, junction patches and stitching (wgbsseg_segment_regions), borders back on the
host.  Inputs are synthetic (seeded; wgbs_tools_amd/synth.py) and already in HBM when the timed region starts.
N > 1 under torch.distributed.run: the chunk grid is cut into N contiguous runs of chunks (one per rank, balanced by the
scored blocks they hold); every rank generates and holds only its own window; no collective on the data path, ranks
only meet at the timing barrier.  `python bench.py --gpus N` WITHOUT the launcher drives N GPUs from this one process
through a share group (the product's `wgbstools segment --gpus N`: one host thread per GPU, one host-side stitching
tree).  Total work is fixed as N grows => "scaling": "strong".

The JSON line carries `roofline` for the HBM-bound scan pass (k_validate for a job without wide tiles — the default
genome —, k_scan with carries otherwise; algorithmic bytes = 2 * samples * sites per launch) measured with HIP events on the kernel's own stream inside the timed steps, the fp64-VALU
bound scoring kernel's rate as `scoring`, and `cpu_baseline`: the reference's own `segmentor` (oracle/_ref, built
from the reference sources) timed on this host on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import os.path as op
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = op.dirname(op.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20260926
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
FP64_VALU_PEAK = 78.6e12         # flop/s, vector fp64 (FMA counted as 2)
VALU_PER_EVAL = 44.5             # VALU instructions per evaluation, common path (tools/micro/count_cost_loop.py): 8-instruction division core
VALU_PER_EVAL_SHORT = 40.5       # the same with the verified 4-instruction division core (narrow tiles, pseudo count >= 1)
FP64_FLOP_PER_EVAL = 29          # fp64 flops of one evaluation, FMA = 2: log2f 5 FMA + 1 mul, fast log2 7 FMA, fused sum 1 FMA, 1-p, accumulate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--samples', type=int, default=32, help='number of beta files (BASELINE.json configs[2]: 32)')
    ap.add_argument('--sites', type=int, default=28217448, help='CpG sites of the synthetic genome (hg19: 28,217,448)')
    ap.add_argument('--chunk', type=int, default=60000)
    ap.add_argument('--max-cpg', type=int, default=1000)
    ap.add_argument('--max-bp', type=int, default=2000)
    ap.add_argument('--pcount', type=float, default=15.0)
    ap.add_argument('--islands', action='store_true', help='add CpG islands to the synthetic loci (windows of several hundred sites; not the BASELINE workload)')
    ap.add_argument('--block-sums', action='store_true', help='also time the block reduction (beta_to_blocks / beta_to_table kernel) over the blocks just found')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='target wall time of the CPU baseline sample (0: skip)')
    ap.add_argument('--e2e', type=int, default=1, help='also time the CLI end to end on page-cached files (1 GPU only; 0: skip)')
    return ap.parse_args()


def cpu_baseline(args, buf, sizes, loci, seg, params):
    """The reference `segmentor` (oracle/_ref, built from the reference's own sources) on this host's cores over a
    bounded sample of the same workload: rounds of `cores` default-size chunks taken evenly from the genome's chunk
    grid, one single-threaded process per core — the reference's own parallel shape (segment.py:144-146, a Pool of
    chunk processes).  Falls back to the oracle's C port (threads) when the reference binary is absent.  Also
    checks the GPU's borders for exactly those chunks against what the CPU produced."""
    import threading
    from oracle import oracle
    cores = os.cpu_count() or 1
    n = args.chunk
    kind = 'reference' if oracle.have_ref() else 'port'
    grid, pos = [], 0
    for sz in sizes:                                       # full-size chunks only (0-based starts)
        grid += [s for s in range(pos, pos + sz - n + 1, n)]
        pos += sz
    if not grid:
        grid, n = [0], min(n, sizes[0])
    pick = [grid[i] for i in np.linspace(0, len(grid) - 1, min(len(grid), cores * 8)).astype(int)]
    pick = sorted(set(pick))
    done, wall, borders = 0, 0.0, {}
    td = tempfile.mkdtemp(dir='/dev/shm' if op.isdir('/dev/shm') else None)
    try:
        t_end = time.time() + args.cpu_seconds
        while done < len(pick) and (done == 0 or time.time() < t_end):
            todo = pick[done:done + cores]
            host = {st: buf[:, 2 * st:2 * (st + n)].cpu().numpy() for st in todo}
            if kind == 'reference':
                jobs = []
                for st in todo:
                    d = op.join(td, 'c%d' % st)
                    os.mkdir(d)
                    paths = []
                    for s in range(host[st].shape[0]):
                        pth = op.join(d, 's%04d.beta' % s)
                        host[st][s].tofile(pth)
                        paths.append(pth)
                    cmd = [oracle.REF_BIN] + paths + ['-s', '0', '-n', str(n), '-max_cpg', str(params['max_cpg']),
                                                     '-ps', repr(float(args.pcount)), '-max_bp', str(args.max_bp)]
                    stdin = ('\n'.join(map(str, loci[st:st + n].tolist())) + '\n').encode()
                    jobs.append((st, cmd, stdin))
                outs = {}

                def run(st, cmd, stdin):
                    outs[st] = subprocess.run(cmd, input=stdin, stdout=subprocess.PIPE, check=True).stdout
                th = [threading.Thread(target=run, args=j) for j in jobs]
                t0 = time.time()
                [t.start() for t in th]
                [t.join() for t in th]
                wall += time.time() - t0
                for st in todo:
                    borders[st] = np.array(outs[st].split(), dtype=np.int64)
            else:
                t0 = time.time()

                def runp(st):                              # ctypes drops the GIL: one chunk per thread
                    sl = [host[st][s].reshape(-1, 2) for s in range(host[st].shape[0])]
                    borders[st] = oracle.segment_chunk(sl, loci[st:st + n], args.pcount, params['max_cpg'], args.max_bp).astype(np.int64)
                thr = [threading.Thread(target=runp, args=(st,)) for st in todo]
                [t.start() for t in thr]
                [t.join() for t in thr]
                wall += time.time() - t0
            done += len(todo)
    finally:
        import shutil
        shutil.rmtree(td, ignore_errors=True)
    sts = sorted(borders)
    got = seg.segment_chunks(sts, [n] * len(sts), args.pcount, params['max_cpg'], args.max_bp)
    same = all(np.array_equal(g.astype(np.int64), borders[st]) for g, st in zip(got, sts))
    used = min(cores, len(sts))
    return {'value': len(sts) * n / wall, 'unit': 'CpG-sites/s', 'cores': used, 'kind': kind,
            'sample': '%d chunks of %d CpGs x %d betas spread over the genome, %d concurrent single-threaded %s, %.1f s wall '
                      '(host has %d logical CPUs)' % (len(sts), n, args.samples, used,
                                                      'reference segmentor processes' if kind == 'reference' else 'oracle-port threads',
                                                      wall, cores),
            'gpu_borders_identical_on_sample': bool(same)}


def end_to_end(args, buf, sizes, names, loci):
    """`wgbstools segment` as a user runs it: .beta files (just written: in the page cache) -> BED, through the CLI entry point in
    this process.  SURVEY.md 8(d)(ii); PCIe- and file-I/O-inclusive, never `value`."""
    import contextlib
    import io
    import shutil
    from wgbs_tools_amd import wgbs_tools
    d = tempfile.mkdtemp(prefix='wgbs_e2e_')      # just written = in the page cache (tmpfs, measured, is the slower place: its page faults and writes cost 2x)
    try:
        ref = op.join(d, 'references', 'synth')
        os.makedirs(ref)
        with open(op.join(ref, 'CpG.chrome.size'), 'w') as f:
            for c, sz in zip(names, sizes):
                f.write('%s\t%d\n' % (c, sz))
        with open(op.join(ref, 'chrome.size'), 'w') as f:
            pos = 0
            for c, sz in zip(names, sizes):
                f.write('%s\t%d\n' % (c, int(loci[pos + sz - 1]) + 10000))
                pos += sz
        import gzip
        with gzip.open(op.join(ref, 'CpG.bed.gz'), 'wb') as f:       # only ever read to build loci.u32, which is written below
            f.write(b'')
        os.symlink('CpG.bed.gz', op.join(ref, 'rev.CpG.bed.gz'))
        loci.tofile(op.join(ref, 'loci.u32'))
        paths = []
        for s in range(args.samples):
            pth = op.join(d, 's%03d.beta' % s)
            buf[s, :2 * args.sites].cpu().numpy().tofile(pth)
            paths.append(pth)
        out = op.join(d, 'blocks.bed')
        best, rows, phases = None, 0, []
        for _ in range(3):
            err = io.StringIO()
            t0 = time.perf_counter()
            with contextlib.redirect_stderr(err):
                rc = wgbs_tools.main(['wgbstools', 'segment', '--betas'] + paths + ['--genome', ref, '-o', out, '--gpus', '1',
                                     '--stats', op.join(d, 'run.json')])
            dt = time.perf_counter() - t0
            assert rc == 0, err.getvalue()[-500:]
            if best is None or dt < best:
                best = dt
                try:
                    with open(op.join(d, 'run.json')) as f:
                        phases = [json.load(f).get('phases_s')]
                except Exception:
                    phases = []
        rows = sum(1 for _ in open(out))
        return {'wall_s': best, 'value': args.sites / best, 'unit': 'CpG-sites/s', 'bed_rows': rows, 'bed_MB': op.getsize(out) / 1e6,
                'phases_of_best_run': phases[0] if phases else None,
                'what': '`wgbstools segment --betas <%d page-cached files> -o blocks.bed`, best of 3 in-process runs: files -> HBM -> borders -> BED '
                        '(PCIe and file I/O included; not `value`)' % args.samples}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from wgbs_tools_amd import _lib, synth, parallel

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    ndev = torch.cuda.device_count()
    group_mode = world == 1 and args.gpus > 1    # ONE process drives --gpus GPUs (the product's `wgbstools segment --gpus N`)
    oversub = world > 1 and ndev < world            # test mode: more ranks than GPUs (e.g. 2 ranks on a 1-GPU box)
    local = local % max(1, ndev)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        if oversub:
            dist.init_process_group('gloo')          # RCCL refuses two ranks on one device; the barrier is all we need
        else:
            dist.init_process_group('nccl', device_id=dev)

    names, sizes = synth.genome_shape(args.sites, 25 if args.sites >= 2500000 else max(1, min(25, args.sites // 100000)))
    sizes = [int(s) for s in sizes]
    loci = synth.synth_loci(SEED, sizes, islands=args.islands)
    max_cpg = min(args.max_cpg, args.max_bp // 2)            # segment.py:65
    params = dict(max_cpg=max_cpg, pcount=args.pcount, max_bp=args.max_bp)
    regions = parallel.regions_of_sizes(sizes)
    S = _lib.load_synth()

    def device_rows(lo, hi, device):
        """synthetic sites [lo, hi) of every sample, straight into the HBM of `device`"""
        n = hi - lo
        pitch = ((2 * n + 255) // 256) * 256 + 256
        b = torch.empty((args.samples, pitch), dtype=torch.uint8, device=torch.device('cuda', device))
        rc = S.wgbssynth_fill_betas_range(C.c_void_p(b.data_ptr()), pitch, lo, hi, 0, args.samples, SEED, device)
        assert rc == 0, 'synthetic fill failed (hip error %d)' % rc
        return b, pitch

    seg = grp = buf = None
    shares = None
    if group_mode:
        devices = [d % max(1, ndev) for d in range(args.gpus)]
        grp = _lib.SegmenterGroup(devices)
        shares = grp.plan(loci, regions, args.chunk, args.pcount, max_cpg, args.max_bp)
        for d in range(args.gpus):
            lo, hi = int(shares['win_lo'][d]), int(shares['win_hi'][d])
            if hi > lo:
                b, pitch = device_rows(lo, hi, devices[d])
                grp.share_set_device(d, b.data_ptr(), args.samples, pitch, keepalive=b)
        torch.cuda.set_device(local)
        my_sites = args.sites
        n_chunks_total = int(shares['chunks'].sum())

        def step():
            return grp.segment_regions(copy=False)

        def timings():
            return grp.timings(0)
    else:
        if world > 1:
            # one process per GPU: this rank holds (and generates) only its own window of the genome
            shares = parallel.plan(regions, args.chunk, world, loci, params)
            lo, hi = int(shares['win_lo'][rank]), int(shares['win_hi'][rank])
            mine = parallel.pieces_of_rank(regions, args.chunk, world, rank, loci, params, shares=shares)
            n_chunks_total = int(shares['chunks'].sum())
        else:
            lo, hi = 0, args.sites
            mine = [(i, a, b) for i, (a, b) in enumerate(regions)]
            n_chunks_total = len(parallel.chunk_grid(regions, args.chunk))
        buf, pitch = device_rows(lo, max(hi, lo + 1), local)
        seg = _lib.Segmenter(local)
        seg.set_betas_device(buf.data_ptr(), args.samples, pitch, max(hi - lo, 1), keepalive=buf)
        seg.set_loci(loci[lo:max(hi, lo + 1)])
        seg.set_site_base(lo)
        st = np.array([p[1] for p in mine], dtype=np.int64) - lo
        en = np.array([p[2] for p in mine], dtype=np.int64) - lo
        my_sites = int((en - st).sum())

        def step():
            if not len(st):
                return [], {}
            return seg.segment_regions(st, en, args.chunk, args.pcount, max_cpg, args.max_bp, copy=False)

        def timings():
            return seg.timings()
    torch.cuda.synchronize()

    def barrier():
        for d in range(ndev if group_mode else 0):
            torch.cuda.synchronize(d)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier() if oversub else dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    acc = None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, stats = step()
        t = timings()
        if acc is None:
            acc = dict(t)
        else:
            for k in t:
                acc[k] = max(acc[k], t[k]) if k in ('max_window', 'n_stages', 'scan_main_bytes', 'div_short') else acc[k] + t[k]
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device='cpu' if oversub else dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    n_blocks = int(sum(len(r) - 1 for r in res))
    block_sums = None
    if args.block_sums and rank == 0 and seg is not None:
        # the immediate consumer of the borders: (#meth, #cov) of every block in every sample.
        # Algorithmic bytes = 2 * N * (sites covered); the D2H copy of the table is outside the kernel time.
        bs = np.concatenate([np.asarray(r[:-1], dtype=np.int64) for r in res]) - 1
        be = np.concatenate([np.asarray(r[1:], dtype=np.int64) for r in res]) - 1
        times = []
        for mode in (1, 1, 1, 3):
            seg.block_sums(bs, be, mode=mode, min_cov=4)
            times.append(seg.last_block_sums_ms())
        covered = int((be - bs).sum())
        block_sums = {'kernel': 'k_block_sums (per block, per sample sums of meth/cov; .bin rows or means)', 'blocks': int(bs.size),
                      'ms_bin_rows': min(times[:3]), 'ms_means': times[3], 'algorithmic_bytes': 2 * covered * args.samples,
                      'GB_per_s': 2 * covered * args.samples / (min(times[:3]) * 1e-3) / 1e9, 'frac_of_hbm_peak': 2 * covered * args.samples / (min(times[:3]) * 1e-3) / 1e9 / HBM_PEAK_GBS}

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = args.sites / (dt / args.steps)
        # dominant kernel launch = the scan of the batch that holds the chunks (one per step); the follow-up batches
        # (a few hundred ~100-site patches) launch it on kilobytes and are reported separately
        main_ms = acc['scan_main_ms'] / args.steps
        scan_gbs = acc['scan_main_bytes'] / (main_ms * 1e-3) / 1e9
        scan_all_gbs = acc['scan_bytes'] / (acc['scan_ms'] * 1e-3) / 1e9
        evals_s = acc['evals'] / (acc['cost_ms'] * 1e-3)
        valu_per_eval = VALU_PER_EVAL_SHORT if acc.get('div_short') else VALU_PER_EVAL
        stats_wide = acc['max_window'] > 60          # WG_NARROW_WMAX: wide scoring tiles exist, so the scan keeps its carries
        # HBM traffic of that launch from the PMC counters (collected separately with rocprofv3, profiles/): only
        # reported when the committed measurement is for exactly this workload
        traffic, traffic_note = None, 'traffic: PMC pass not available for this workload'
        for tj in (op.join(ROOT, 'profiles', 'r02_scan_traffic.json'), op.join(ROOT, 'profiles', 'r01_scan_traffic.json')):
            if op.isfile(tj) and world == 1 and not group_mode:
                tr = json.load(open(tj))
                if abs(tr['algorithmic_bytes'] - acc['scan_main_bytes']) <= 0.001 * acc['scan_main_bytes']:
                    traffic = tr['traffic_bytes']
                    traffic_note = ('traffic (bytes per launch) = 2 x FETCH_SIZE + WRITE_SIZE from profiles/%s '
                                    '(rocprofv3 PMC passes of this same command; gfx950 FETCH_SIZE x2 correction)' % op.basename(tj))
                    break
        mode = ('one process, %d GPUs: a share group (work-balanced contiguous chunk runs, one host thread per GPU, one host-side tree)' % args.gpus
                if group_mode else 'one process per GPU: work-balanced contiguous chunk runs per rank, no collective' if world > 1 else 'one GPU')
        out = {
            'metric': 'CpG-sites/sec segmented',
            'value': value, 'unit': 'CpG-sites/s', 'n_gpus': args.gpus if group_mode else world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_step, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'u8 counts -> u32 prefix sums -> f32/f64 log-likelihood (bit-exact with the reference)',
            'data': 'synthetic (seeded hg19-shaped genome and betas, generated on the device)',
            'config': {'workload': 'hg19-shaped %d CpGs x %d betas, whole-genome segment, chunk_size %d, max_cpg %d, max_bp %d, pcount %g'
                                   % (args.sites, args.samples, args.chunk, args.max_cpg, args.max_bp, args.pcount) + (' + CpG islands in the loci' if args.islands else ''),
                       'baseline_config': 'BASELINE.json configs[2]' if (args.sites, args.samples, args.islands) == (28217448, 32, False) else 'custom',
                       'chunks': n_chunks_total, 'chromosomes': len(sizes), 'sharding': mode,
                       'share_chunks': None if shares is None else [int(x) for x in shares['chunks']],
                       'share_work': None if shares is None else [int(x) for x in shares['work']],
                       'rank0_sites': my_sites, 'rank0_stats': stats, 'rank0_blocks': n_blocks},
            'roofline': {'kernel': ('k_scan (per-sample prefix scan -> 128-site carries + meth<=cov validation: the job has wide tiles)' if stats_wide else
                                    'k_validate (the scan pass of a job without wide tiles: every beta byte read once, meth<=cov checked; no carries needed)'), 'bound': 'hbm',
                         'achieved': scan_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': scan_gbs / HBM_PEAK_GBS,
                         'traffic': traffic,
                         'algorithmic_bytes_per_launch': acc['scan_main_bytes'], 'avg_launch_ms': main_ms,
                         'launches_timed': args.steps,
                         'all_launches': {'count': acc['scan_launches'], 'bytes': acc['scan_bytes'], 'ms': acc['scan_ms'],
                                          'GB/s': scan_all_gbs},
                         'note': 'rank 0 / share 0, HIP events on the kernel stream inside the timed steps; ' + traffic_note},
            # the kernel that IS the step (k_cost: ~85 %): neither HBM- nor MFMA-bound, the bound is VALU issue.
            # VALU_PER_EVAL = VALU instructions per (block, sample) evaluation on the common path of the guard-free form in the
            # gfx950 ISA (tools/micro/count_cost_loop.py); issue peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz.
            # fp64: FP64_FLOP_PER_EVAL counts the fp64 VALU work of one evaluation with an FMA as 2 against the
            # 78.6 TFLOP/s vector-fp64 peak - informative only, the kernel also spends issue slots on fp32/int/conversions.
            'roofline_cost': {'kernel': 'k_cost (block log-likelihoods)', 'bound': 'valu-issue',
                              'achieved': evals_s * valu_per_eval, 'peak': 256 * 4 * 16 * 2.4e9, 'unit': 'lane-ops/s',
                              'frac': evals_s * valu_per_eval / (256 * 4 * 16 * 2.4e9),
                              'evals_per_s': evals_s, 'valu_instr_per_eval': valu_per_eval,
                              'division_core': '4 instructions, verified on the device for this pseudo count' if acc.get('div_short') else '8 instructions',
                              'fp64_flop_per_eval': FP64_FLOP_PER_EVAL, 'fp64_tflops': evals_s * FP64_FLOP_PER_EVAL / 1e12,
                              'fp64_frac_of_peak': evals_s * FP64_FLOP_PER_EVAL / FP64_VALU_PEAK,
                              'evals_per_step': acc['evals'] / args.steps, 'pairs_per_step': acc['pairs'] / args.steps,
                              'max_window': acc['max_window'], 'stages': acc['n_stages'],
                              'avg_ms_per_step': acc['cost_ms'] / args.steps},
            'block_sums': block_sums,
            'device_ms_per_step': {k: acc[k] / args.steps for k in ('scan_ms', 'window_ms', 'cost_ms', 'dp_ms', 'trace_ms', 'total_ms')},
        }
        if world == 1 and not group_mode and args.e2e:
            try:
                out['end_to_end'] = end_to_end(args, buf, sizes, names, loci)
            except Exception as e:
                out['end_to_end'] = {'value': None, 'what': 'failed: %r' % (e,)}
        if world == 1 and not group_mode and args.cpu_seconds > 0:
            try:
                out['cpu_baseline'] = cpu_baseline(args, buf, sizes, loci, seg, params)
                out['cpu_baseline']['gpu_over_cpu'] = value / out['cpu_baseline']['value']
            except Exception as e:                       # the baseline must never break the bench line
                out['cpu_baseline'] = {'value': None, 'unit': 'CpG-sites/s', 'cores': os.cpu_count(), 'kind': 'reference',
                                       'sample': 'failed: %r' % (e,)}
        print(json.dumps(out), flush=True)
    if seg is not None:
        seg.close()
    if grp is not None:
        grp.close()
    if world > 1:
        dist.barrier() if oversub else dist.barrier(device_ids=[local])
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
