#!/usr/bin/env python3
"""bench.py — CpG-sites/sec segmented on synthetic hg19-shaped betas (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--samples 32] [--sites 28217448]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one whole `segment` job over the resident beta bytes: scan/validate pass, window extents, block scoring,
changepoint DP, traceback, junction patches and the reference's stitching tree, borders back on the host.  Inputs are
synthetic (seeded; wgbs_tools_amd/synth.py) and already in HBM when the timed region starts.

N = 1: wgbsseg_segment_regions on one context.
N > 1 under torch.distributed.run (one process per GPU): the product's multi-process path, wgbs_tools_amd/parallel.py
    ShardedRun — every rank computes the chunks and up-front junction patches it owns (work-balanced contiguous runs of
    chunks; it generates and holds only its own window of the genome), hands its border lists to rank 0 through a slot in
    /dev/shm, and rank 0 runs the ONE stitching tree over all chunks inside the timed step (follow-up patches on its own
    GPU).  No collective on the data path; the ranks meet at the hand-over barrier of each step.
N > 1 WITHOUT the launcher: one process drives N GPUs through a share group (`wgbstools segment --gpus N`).
Total work is fixed as N grows => "scaling": "strong".

The JSON line carries
  roofline        the dominant kernel, k_cost (block log-likelihoods: ~85 % of a step).  Neither HBM- nor MFMA-bound: integer
                  scan + scalar fp32/fp64 cost on the vector ALUs.  achieved = algorithmic flops (~55
                  fp64-equivalent flops per (block, sample) evaluation) x evaluations per launch / the launch's duration (HIP
                  events on the kernel's stream inside the timed steps), against the 78.6 TFLOP/s vector-fp64 peak; `issue` gives
                  the instruction-issue bound of the kernel's actual instruction mix (profiles/*cost_isa_mix.json x the guide's
                  cycle table).
  roofline_scan   the HBM-bound scan pass (k_validate / k_scan): algorithmic bytes = 2 x samples x sites per launch.
  block_sums      the block reduction (beta_to_blocks / beta_to_table kernel) over the blocks just found.
  matrix          the same step at the other sample counts of the metric (x8, x200, x512), a few steps each.
  cpu_baseline    the reference's own `segmentor` (oracle/_ref) on this host: 1 core, one process per physical core, one per
                  logical CPU; bounded sample of the same workload.
  end_to_end      the CLI on page-cached files (PCIe and file I/O included; never `value`); the x200 row of `matrix` carries one too.
  extras          (round 5) the SURVEY 8(f) rows that had no figure: `convert` (k_convert), `pat2beta` (kernel alone, from host memory, through the CLI on
                  a BGZF file, the host's inflate alone, and the reference's stdin2beta on this host), find_markers' statistics (k_marker_stats).
N > 1 lines say what ran: `n_gpus` = DISTINCT physical devices among the shares / ranks (device_report), with `gpus_requested`, `shares`,
`distinct_devices`, `devices_visible`, `oversubscribed` in `config` — eight shares wrapped onto one GPU are `n_gpus: 1, shares: 8`.
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
FP64_VALU_PEAK = 78.6e12         # flop/s, vector fp64 (FMA counted as 2): 256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz
FLOP_PER_EVAL = 55               # one (block, sample) evaluation ~ 50-60 fp64-equivalent flops (2 int sub, 2 cvt, 2 fadd,
                                 # fdiv, log2f, fmul, dsub, log2, dmul, 2 dadd, cvt)
FP64_FLOP_EXECUTED = 29          # fp64 flops the kernel actually executes per evaluation (FMA = 2), informative
N_SIMD, CLOCK_HZ = 256 * 4, 2.4e9
MATRIX_SAMPLES = (8, 200, 512)   # the other sample counts of BASELINE.json's metric


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
    ap.add_argument('--block-sums', type=int, default=1, help='also time the block reduction over the blocks just found (0: skip)')
    ap.add_argument('--scan-carries', type=int, default=1, help='also time the prefix-sum pass with carries (k_scan) on the chunk grid (0: skip)')
    ap.add_argument('--matrix', type=int, default=1, help='also run a few steps at x8, x200 and x512 (1 GPU only; 0: skip)')
    ap.add_argument('--cpu-seconds', type=float, default=25.0, help='wall-time budget of the CPU baseline runs (0: skip)')
    ap.add_argument('--extras', type=int, default=1, help='also time convert, pat2beta and the find_markers statistics (1 GPU only; 0: skip)')
    ap.add_argument('--e2e', type=int, default=1, help='also time the CLI end to end on page-cached files (1 GPU only; 0: skip)')
    ap.add_argument('--oversubscribe', action='store_true',
                    help='dry runs of the N > 1 forms on fewer GPUs than shares / ranks (several shares per device): without this flag such a launch is refused')
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference binary on this host
# ------------------------------------------------------------------------------------------------------------
def physical_cores():
    """One logical CPU of every physical core this process may run on (thread_siblings_list), and all allowed logical CPUs."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, firsts = set(), []
    for c in allowed:
        try:
            sib = open('/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list' % c).read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            firsts.append(c)
    return firsts, allowed


def cpu_baseline(args, buf, sizes, loci, seg, params):
    """The reference `segmentor` (oracle/_ref, built from the reference's own sources; the oracle's C port when it is absent) on
    default-size chunks spread evenly over the genome's chunk grid, one single-threaded process per chunk — the reference's own
    parallel shape (segment.py:144-146) — three ways: ONE process on one core, one process pinned to each PHYSICAL core, one per
    LOGICAL CPU (every round of a run starts all its processes together).  `value` is the best of them.  The GPU's borders of the
    sampled chunks are compared with the CPU's."""
    from oracle import oracle
    kind = 'reference' if oracle.have_ref() else 'port'
    phys, logical = physical_cores()
    n = args.chunk
    grid, pos = [], 0
    for sz in sizes:                                       # full-size chunks only (0-based starts)
        grid += list(range(pos, pos + sz - n + 1, n))
        pos += sz
    if not grid:
        grid, n = [0], min(n, sizes[0])
    want = min(len(grid), len(logical))
    pick = sorted(set(grid[i] for i in np.linspace(0, len(grid) - 1, want).astype(int)))
    td = tempfile.mkdtemp(dir='/dev/shm' if op.isdir('/dev/shm') else None)
    borders, runs = {}, []
    t_begin = time.time()

    def prepare(st):
        d = op.join(td, 'c%d' % st)
        if op.isdir(d):
            return
        os.mkdir(d)
        host = buf[:, 2 * st:2 * (st + n)].cpu().numpy()
        for s in range(host.shape[0]):
            host[s].tofile(op.join(d, 's%04d.beta' % s))
        with open(op.join(d, 'loci.txt'), 'w') as f:
            f.write('\n'.join(map(str, loci[st:st + n].tolist())) + '\n')

    def run_round(chunks, cpus):
        """all `chunks` at once, chunk i pinned to cpus[i] (None: wherever the scheduler puts it) -> wall seconds"""
        for st in chunks:
            prepare(st)
        procs = []
        t0 = time.time()
        for i, st in enumerate(chunks):
            d = op.join(td, 'c%d' % st)
            if kind == 'reference':
                cmd = [oracle.REF_BIN] + [op.join(d, 's%04d.beta' % s) for s in range(args.samples)] + \
                      ['-s', '0', '-n', str(n), '-max_cpg', str(params['max_cpg']), '-ps', repr(float(args.pcount)), '-max_bp', str(args.max_bp)]
            else:                                          # the oracle's C port through a tiny driver process
                cmd = [sys.executable, '-c', 'import sys,numpy as np;sys.path.insert(0,%r);from oracle import oracle;'
                       'd=%r;n=%d;sl=[np.fromfile(d+"/s%%04d.beta"%%s,dtype=np.uint8).reshape(-1,2) for s in range(%d)];'
                       'print(*oracle.segment_chunk(sl,np.loadtxt(d+"/loci.txt",dtype=np.uint32),%r,%d,%d).tolist())'
                       % (ROOT, d, n, args.samples, float(args.pcount), params['max_cpg'], args.max_bp)]
            cpu = cpus[i] if cpus else None
            procs.append(subprocess.Popen(cmd, stdin=open(op.join(d, 'loci.txt')), stdout=open(op.join(d, 'out.txt'), 'w'),
                                          preexec_fn=(lambda c=cpu: os.sched_setaffinity(0, {c})) if cpu is not None else None))
        for p in procs:
            if p.wait() != 0:
                raise RuntimeError('CPU baseline process failed')
        wall = time.time() - t0
        for st in chunks:
            borders[st] = np.array(open(op.join(td, 'c%d' % st, 'out.txt')).read().split(), dtype=np.int64)
        return wall

    try:
        plans = [('1 core', [pick[len(pick) // 2]], [phys[0]])]
        if len(phys) > 1:
            sub = [pick[i] for i in np.linspace(0, len(pick) - 1, min(len(pick), len(phys))).astype(int)]
            plans.append(('%d physical cores, one pinned process each' % len(sub), sub, phys[:len(sub)]))
        if len(logical) > len(phys):
            plans.append(('%d logical CPUs, one process each (unpinned)' % len(pick), pick, None))
        for what, chunks, cpus in plans:
            if runs and time.time() - t_begin > args.cpu_seconds:
                runs.append({'what': what, 'skipped': 'time budget (--cpu-seconds %g) spent' % args.cpu_seconds})
                continue
            wall = run_round(chunks, cpus)
            runs.append({'what': what, 'cores': len(chunks), 'chunks': len(chunks), 'wall_s': wall, 'value': len(chunks) * n / wall,
                         'per_core': n / wall})
    finally:
        import shutil
        shutil.rmtree(td, ignore_errors=True)
    sts = sorted(borders)
    got = seg.segment_chunks(sts, [n] * len(sts), args.pcount, params['max_cpg'], args.max_bp)
    same = all(np.array_equal(g.astype(np.int64), borders[st]) for g, st in zip(got, sts))
    done = [r for r in runs if 'value' in r]
    best = max(done, key=lambda r: r['value'])
    return {'value': best['value'], 'unit': 'CpG-sites/s', 'cores': best['cores'], 'kind': kind,
            'sample': 'default-size chunks (%d CpGs x %d betas) spread over the genome, single-threaded %s; best of the runs below: %s, %.1f s wall '
                      '(host: %d physical cores, %d logical CPUs)' % (n, args.samples, 'reference segmentor processes' if kind == 'reference' else 'oracle-port processes',
                                                                      best['what'], best['wall_s'], len(phys), len(logical)),
            'single_core_value': done[0]['value'], 'runs': runs, 'chunks_compared_with_gpu': len(sts),
            'gpu_borders_identical_on_sample': bool(same)}


# ------------------------------------------------------------------------------------------------------------
# end to end through the CLI
# ------------------------------------------------------------------------------------------------------------
def end_to_end(args, buf, sizes, names, loci, samples=None, reps=3):
    """`wgbstools segment` as a user runs it: .beta files (just written: in the page cache) -> BED, through the CLI entry point in
    this process.  SURVEY.md 8(d)(ii); PCIe- and file-I/O-inclusive, never `value`."""
    import contextlib
    import io
    import shutil
    from wgbs_tools_amd import wgbs_tools
    d = tempfile.mkdtemp(prefix='wgbs_e2e_')      # just written = in the page cache (tmpfs, measured, is the slower place: its page faults and writes cost 2x)
    try:
        ref = write_reference_dir(d, names, sizes, loci)
        paths = []
        samples = samples or args.samples
        for s in range(samples):
            pth = op.join(d, 's%03d.beta' % s)
            buf[s, :2 * args.sites].cpu().numpy().tofile(pth)
            paths.append(pth)
        best, rows, phases = None, 0, []
        for rep in range(reps):
            # a NEW output file per run, as a user's run writes one: overwriting the previous 115 MB of BED makes the kernel drop its page-cache
            # pages first (O_TRUNC: ~25 ms on this host, more than the writing itself) — that is not part of the pipeline
            out = op.join(d, 'blocks_%d.bed' % rep)
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
                'what': '`wgbstools segment --betas <%d page-cached files> -o blocks.bed`, best of %d in-process runs: files -> HBM -> borders -> BED '
                        '(PCIe and file I/O included; not `value`)' % (samples, reps), 'samples': samples,
                'beta_GB': 2e-9 * args.sites * samples}
    finally:
        shutil.rmtree(d, ignore_errors=True)


# ------------------------------------------------------------------------------------------------------------
# the rows of SURVEY.md 8(f) beside the path: convert, pat2beta, find_markers' statistics
# ------------------------------------------------------------------------------------------------------------
def synth_pat_text(seed, n_sites, n_reads):
    """A pat file's text (bytes) over CpGs 1..n_sites, built with numpy: `chr1 \t first CpG \t pattern over {C,T,H,.} \t count \n`, reads of
    1-12 CpGs, counts 1-40, sorted by start as real pat files are (the same distribution as synth.synth_pat_lines, millions of lines per second)."""
    from wgbs_tools_amd.synth import hash_at
    idx = np.arange(n_reads, dtype=np.int64)
    start = np.sort((hash_at(seed, 91, idx) % np.uint64(n_sites)).astype(np.int64) + 1)
    h1 = hash_at(seed, 92, idx)
    ln = 1 + ((h1 & np.uint64(15)).astype(np.int64) % 12)
    cnt = 1 + ((h1 >> np.uint64(8)) % np.uint64(40)).astype(np.int64)
    hp = hash_at(seed, 93, idx)
    ds = np.floor(np.log10(start)).astype(np.int64) + 1                   # digits of the start
    dc = 1 + (cnt >= 10)
    size = 5 + ds + 1 + ln + 1 + dc + 1
    off = np.concatenate([[0], np.cumsum(size)])
    buf = np.empty(int(off[-1]), dtype=np.uint8)
    o = off[:-1]
    for k, ch in enumerate(b'chr1\t'):
        buf[o + k] = ch
    for k in range(int(ds.max())):
        m = ds > k
        buf[o[m] + 5 + ds[m] - 1 - k] = 48 + (start[m] // 10 ** k) % 10
    buf[o + 5 + ds] = 9
    alphabet = np.frombuffer(b'CCCTTTH.', dtype=np.uint8)
    po = o + 6 + ds
    for k in range(12):
        m = ln > k
        buf[po[m] + k] = alphabet[((hp[m] >> np.uint64(3 * k)) & np.uint64(7)).astype(np.int64)]
    buf[po + ln] = 9
    co = po + ln + 1
    two = dc == 2
    buf[co[two]] = 48 + cnt[two] // 10
    buf[co + dc - 1] = 48 + cnt % 10
    buf[co + dc] = 10
    return buf.tobytes(), int(ln.sum())


def write_bgzf(path, data, level=6):
    """`data` as a BGZF file (what bgzip writes and wgbstools' .pat.gz are: gzip members of <= 64 KB with their size in a 'BC' extra field)."""
    import struct
    import zlib
    with open(path, 'wb') as f:
        for p in list(range(0, len(data), 65280)) + [None]:
            blk = b'' if p is None else data[p:p + 65280]
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            body = c.compress(blk) + c.flush()
            f.write(b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00' + struct.pack('<H', len(body) + 25) + body +
                    struct.pack('<II', zlib.crc32(blk), len(blk)))


def write_reference_dir(d, names, sizes, loci):
    """a genome directory the CLI entry points accept (the loci cached as loci.u32: CpG.bed.gz is never parsed)"""
    import gzip
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
    with gzip.open(op.join(ref, 'CpG.bed.gz'), 'wb') as f:       # only ever read to build loci.u32, which is written below
        f.write(b'')
    os.symlink('CpG.bed.gz', op.join(ref, 'rev.CpG.bed.gz'))
    loci.tofile(op.join(ref, 'loci.u32'))
    return ref


def extras(args, seg, loci, sizes, names, res):
    """Throughput of the SURVEY.md 8(f) rows that had none in the line: `convert` (k_convert), `pat2beta` (k_pat_count + k_pat_trim; kernel only,
    from host memory, through the CLI on a BGZF file, and the reference's stdin2beta on this host beside it), find_markers' per-block
    statistics (k_marker_stats).  Each with the bound it runs against.  A few seconds in total."""
    import shutil
    from oracle import pat2beta_oracle as OP
    from wgbs_tools_amd import _lib, wgbs_tools
    out = {}
    cum = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    L64 = loci.astype(np.int64)
    n_sites = int(cum[-1])
    # ---- convert: 2 M BED regions -> CpG ranges (two lower bounds in the chromosome's slice of the loci per region)
    try:
        rng = np.random.default_rng(7)
        n = 2000000
        ci = rng.integers(0, len(sizes), n)
        clo, chi = cum[ci], cum[ci + 1]
        anchor = L64[rng.integers(clo, chi)]
        start = anchor + rng.integers(-300, 300, n)
        end = start + rng.integers(1, 4000, n)
        cbp = L64[chi - 1] + 10000
        slow = (rng.random(n) < 0.1).astype(np.uint8)
        seg.convert_regions(clo, chi, cbp, start, end, slow)
        t0 = time.perf_counter()
        s_, e_ = seg.convert_regions(clo, chi, cbp, start, end, slow)
        wall = time.perf_counter() - t0
        ms = seg.last_block_sums_ms()
        steps = 2 * int(np.ceil(np.log2(max(sizes))))
        out['convert'] = {'kernel': 'k_convert (one thread per BED region: two lower bounds in its chromosome\'s slice of the resident loci, both rule sets of convert.py:133-185)',
                          'regions': n, 'kernel_ms': ms, 'regions_per_s': n / (ms * 1e-3), 'call_wall_ms': wall * 1e3, 'regions_per_s_call': n / wall,
                          'mapped': int((s_ != 0).sum()),
                          'bound': 'memory latency: ~%d dependent 4-byte loads per region into a %.0f MB array (L2 / MALL resident); the call itself is the '
                                   'PCIe copy of 5 x 16 MB in and 2 x 16 MB out' % (steps, loci.nbytes / 1e6),
                          'dependent_loads_per_s': n * steps / (ms * 1e-3)}
    except Exception as e:
        out['convert'] = {'failed': repr(e)}
    # ---- find_markers: per-block statistics of a target and a background set over the ratio table of the blocks just found
    try:
        bs = np.concatenate([np.asarray(r[:-1], dtype=np.int64) for r in res]) - 1
        be = np.concatenate([np.asarray(r[1:], dtype=np.int64) for r in res]) - 1
        seg.block_sums(bs, be, mode=3, min_cov=4)
        t_table = seg.last_block_sums_ms()
        tg = list(range(0, args.samples, 4))
        bg = [s for s in range(args.samples) if s % 4]
        seg.marker_stats(tg, bg, bs.size)
        seg.marker_stats(tg, bg, bs.size)
        ms = seg.last_block_sums_ms()
        rd = 8.0 * bs.size * (len(tg) + len(bg))
        out['find_markers'] = {'kernel': 'k_marker_stats (per block: count, sequential sum, min, max of the target and of the background samples\' ratios; find_markers.py:318-335) '
                                         'over the device-resident ratio table of block-reduction mode 3',
                               'blocks': int(bs.size), 'samples': len(tg) + len(bg), 'kernel_ms': ms, 'block_samples_per_s': bs.size * (len(tg) + len(bg)) / (ms * 1e-3),
                               'ratio_table_ms': t_table, 'algorithmic_bytes': rd + 64.0 * bs.size, 'GB_per_s': (rd + 64.0 * bs.size) / (ms * 1e-3) / 1e9,
                               'frac_of_hbm_peak': (rd + 64.0 * bs.size) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'bound': 'hbm (every ratio of the table read once, 64 B written per block)'}
    except Exception as e:
        out['find_markers'] = {'failed': repr(e)}
    # ---- pat2beta
    d = tempfile.mkdtemp(prefix='wgbs_pat_')
    try:
        n_reads = 4000000
        text, n_chars = synth_pat_text(SEED, n_sites, n_reads)
        mb = len(text) / 1e6
        cuts = [0]
        while cuts[-1] < len(text):
            cuts.append(len(text) if len(text) - cuts[-1] <= (64 << 20) else text.rfind(b'\n', cuts[-1], cuts[-1] + (64 << 20)) + 1)
        rows0 = None
        best_wall, best_k = None, None
        for rep in range(3):
            with _lib.PatBeta(1, n_sites + 1) as pb:
                t0 = time.perf_counter()
                for a, b in zip(cuts[:-1], cuts[1:]):
                    pb.feed(text[a:b])
                k_ms = pb.kernel_ms()
                rows = pb.finish()
                wall = time.perf_counter() - t0
            rows0 = rows
            if rep and (best_wall is None or wall < best_wall):
                best_wall = wall
            if rep and (best_k is None or k_ms < best_k):
                best_k = k_ms
        # the CLI on a BGZF file (inflate on the host's threads + the feed above + the .beta file)
        ref = write_reference_dir(d, names, sizes, loci)
        gz = op.join(d, 'smp.pat.gz')
        write_bgzf(gz, text, level=1)
        cli = None
        for rep in range(2):
            t0 = time.perf_counter()
            rc = wgbs_tools.main(['wgbstools', 'pat2beta', gz, '-o', d, '--genome', ref, '-f'])
            w = time.perf_counter() - t0
            assert not rc
            cli = w if cli is None else min(cli, w)
        same_file = bool(np.array_equal(np.fromfile(op.join(d, 'smp.beta'), dtype=np.uint8).reshape(-1, 2), rows0))
        # the host side of that run alone: the file's text through pat_chunks (BGZF blocks inflated on the host's threads), nothing fed to the device
        from wgbs_tools_amd import pat2beta as P2B
        inflate = None
        for rep in range(2):
            t0 = time.perf_counter()
            got = sum(len(c) for c in P2B.pat_chunks(gz))
            w = time.perf_counter() - t0
            assert got == len(text)
            inflate = w if inflate is None else min(inflate, w)
        # the reference's binary on this host (1 core): the text from memory, and its own pipeline `gunzip -c | stdin2beta` (pat2beta.py:30)
        refrec = None
        if OP.have_ref():
            sample = text[:cuts[1]]
            t0 = time.perf_counter()
            r = subprocess.run([OP.REF_BIN, '1', str(n_sites + 1)], input=sample, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            t_ref = time.perf_counter() - t0
            # (its stdout is the whole genome's counts as text: 2 x 28 M numbers — part of what the reference's pipeline pays per file)
            t0 = time.perf_counter()
            r2 = subprocess.run('gunzip -c %s | %s 1 %d > %s' % (gz, OP.REF_BIN, n_sites + 1, op.join(d, 'ref_counts.txt')), shell=True)
            t_pipe = time.perf_counter() - t0
            refrec = {'binary': 'oracle/_ref/stdin2beta (the reference\'s own source, g++ -O2)', 'cores': 1, 'sample_MB': len(sample) / 1e6, 'wall_s': t_ref,
                      'text_MB_per_s': len(sample) / 1e6 / t_ref, 'pipeline': '`gunzip -c smp.pat.gz | stdin2beta 1 N+1 > counts.txt` over the whole file (pat2beta.py:30; the reference '
                      'then parses that text again with numpy and trims it)', 'pipeline_wall_s': t_pipe, 'pipeline_text_MB_per_s': mb / t_pipe, 'pipeline_rc': r2.returncode, 'rc': r.returncode}
        out['pat2beta'] = {'kernel': 'k_pat_count (4 KB tiles of text through LDS, one line per thread, int32 atomics per covered site) + k_pat_trim; stdin2beta.cpp:59-123',
                           'text_MB': mb, 'reads': n_reads, 'covered_sites_chars': n_chars, 'chunks': len(cuts) - 1,
                           'kernel_ms': best_k, 'kernel_text_GB_per_s': mb / 1e3 / (best_k * 1e-3), 'kernel_reads_per_s': n_reads / (best_k * 1e-3),
                           'from_host_memory_wall_s': best_wall, 'from_host_memory_text_MB_per_s': mb / best_wall, 'from_host_memory_reads_per_s': n_reads / best_wall,
                           'cli_bgzf_wall_s': cli, 'cli_bgzf_text_MB_per_s': mb / cli, 'cli_bgzf_reads_per_s': n_reads / cli, 'cli_file_equals_device_rows': same_file,
                           'bgzf_MB': op.getsize(gz) / 1e6, 'host_inflate_only_wall_s': inflate, 'host_inflate_only_text_MB_per_s': mb / inflate, 'reference_cpu': refrec,
                           'bound': 'the kernel is not the bound: from host memory the call is the copy of the text into page-locked memory and over PCIe (1 byte of text moved per byte '
                                    'counted) + 2 x 4 B x %d sites of counts trimmed and copied back; through the CLI it is the host (host_inflate_only_wall_s: the BGZF blocks inflated on the host\'s threads and cut at line ends, in Python; the rest is the 56 MB .beta file, '
                                    'the genome tables and the accumulator\'s 2 x 4 B x %d counters)' % (n_sites, n_sites)}
    except Exception as e:
        out['pat2beta'] = {'failed': repr(e)}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


# ------------------------------------------------------------------------------------------------------------
# profiles keyed on the library's sources
# ------------------------------------------------------------------------------------------------------------
def keyed_profile(pattern, sha, **match):
    """Newest profiles/<pattern> whose `csrc_sha` is this library's and whose other fields match; None otherwise (a profile of
    another source state is not evidence for this one)."""
    import glob
    for f in sorted(glob.glob(op.join(ROOT, 'profiles', pattern)), reverse=True):
        try:
            rec = json.load(open(f))
        except Exception:
            continue
        if rec.get('csrc_sha') == sha and all(rec.get(k, False if v is False else None) == v for k, v in match.items()):
            rec['_file'] = op.basename(f)
            return rec
    return None


def device_report(form, asked, placements):
    """What actually ran, for the N > 1 lines: `placements` = one (host, device index) per share (share group) or per rank (one process per
    GPU).  `n_gpus` is the number of DISTINCT physical devices among them — never the number asked for: eight shares wrapped onto one GPU
    (a 1-GPU box running `--gpus 8`) or two ranks sharing a device are reported as what they are.  Pure function (tests/test_bench_report_cpu.py)."""
    placements = [(str(h), int(d)) for h, d in placements]
    distinct = len(set(placements))
    units = len(placements)
    what = {'one': 'one GPU',
            'group': 'one process, %d share%s on %d distinct GPU%s: a share group (work-balanced contiguous chunk runs, one host thread per share, one host-side tree)'
                     % (units, '' if units == 1 else 's', distinct, '' if distinct == 1 else 's'),
            'ranks': 'one process per share (parallel.ShardedRun), %d rank%s on %d distinct GPU%s: work-balanced contiguous chunk runs per rank, border lists handed to '
                     'rank 0 through /dev/shm, ONE stitching tree on rank 0 inside the timed step; no collective'
                     % (units, '' if units == 1 else 's', distinct, '' if distinct == 1 else 's')}[form]
    if form != 'one' and distinct < units:
        what += ' — OVERSUBSCRIBED: %d shares share %d device%s, so this line is NOT a %d-GPU measurement' % (units, distinct, '' if distinct == 1 else 's', units)
    return {'n_gpus': distinct, 'gpus_requested': int(asked), 'shares': units, 'distinct_devices': distinct,
            'oversubscribed': distinct < units, 'sharding': what}


def refuse_oversubscription(form, asked, units, devices_visible, allowed):
    """None when every share (share group) / rank (one process per GPU) of an N > 1 launch gets a GPU of its own, or when the launch opted in with
    --oversubscribe; otherwise the message bench.py dies with: a `--gpus 8` run on a box with one GPU must not end as a quiet line that says
    n_gpus 1.  Pure function (tests/test_bench_report_cpu.py)."""
    if form == 'one' or allowed or devices_visible >= units:
        return None
    return ('bench.py: --gpus %d asks for %d %s but this host shows %d GPU%s: the shares would double up and the line would not be a %d-GPU measurement. '
            'Run it on a node with %d GPUs, or pass --oversubscribe for a dry run of the N > 1 plumbing (the line then says n_gpus = the distinct devices and OVERSUBSCRIBED).'
            % (asked, units, 'shares of one process' if form == 'group' else 'ranks', devices_visible, '' if devices_visible == 1 else 's', units, units))


def accumulate(acc, t):
    if acc is None:
        return dict(t)
    for k in t:
        acc[k] = max(acc[k], t[k]) if k in ('max_window', 'n_stages', 'scan_main_bytes', 'div_short') else acc[k] + t[k]
    return acc


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from wgbs_tools_amd import _lib, synth, parallel, build as nbuild

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    ndev = torch.cuda.device_count()
    # the multi-process path: under a launcher with more than one rank — or, to exercise its plumbing (RCCL group, host group,
    # /dev/shm slots) on a 1-GPU box, with ONE rank when WGBSSEG_BENCH_DIST=1 (torch.distributed.run --nproc-per-node 1)
    multi = world > 1 or (os.environ.get('WGBSSEG_BENCH_DIST') == '1' and 'RANK' in os.environ)
    group_mode = not multi and args.gpus > 1    # ONE process drives --gpus GPUs (the product's `wgbstools segment --gpus N`)
    oversub = world > 1 and ndev < world            # dry-run mode: more ranks than GPUs (e.g. 2 ranks on a 1-GPU box)
    refusal = refuse_oversubscription('ranks' if multi else 'group' if group_mode else 'one', args.gpus, world if multi else args.gpus, ndev, args.oversubscribe)
    if refusal:
        if rank == 0:
            print(refusal, file=sys.stderr, flush=True)
        sys.exit(2)
    local = local % max(1, ndev)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    host_group = None
    if multi:
        if oversub:
            dist.init_process_group('gloo')          # RCCL refuses two ranks on one device; the barrier is all we need
            host_group = dist
        else:
            dist.init_process_group('nccl', device_id=dev)

    names, sizes = synth.genome_shape(args.sites, 25 if args.sites >= 2500000 else max(1, min(25, args.sites // 100000)))
    sizes = [int(s) for s in sizes]
    loci = synth.synth_loci(SEED, sizes, islands=args.islands)
    max_cpg = min(args.max_cpg, args.max_bp // 2)            # segment.py:65
    params = dict(max_cpg=max_cpg, pcount=args.pcount, max_bp=args.max_bp)
    regions = parallel.regions_of_sizes(sizes)
    S = _lib.load_synth()
    sha = nbuild.source_hash()

    def device_rows(lo, hi, device, samples=None):
        """synthetic sites [lo, hi) of every sample, straight into the HBM of `device`"""
        samples = samples or args.samples
        n = hi - lo
        pitch = ((2 * n + 255) // 256) * 256 + 256
        b = torch.empty((samples, pitch), dtype=torch.uint8, device=torch.device('cuda', device))
        rc = S.wgbssynth_fill_betas_range(C.c_void_p(b.data_ptr()), pitch, lo, hi, 0, samples, SEED, device)
        assert rc == 0, 'synthetic fill failed (hip error %d)' % rc
        return b, pitch

    seg = grp = buf = run = None
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
    elif multi:
        # one process per GPU: the product's multi-process path (wgbs_tools_amd/parallel.py ShardedRun).  A rank generates and
        # holds only its own window; rank 0, which also serves the follow-up patches of the tree, holds the genome.
        if host_group is None:
            host_group = dist.new_group(backend='gloo')      # the hand-over barrier of a step is a host-side one

        class _G:                                            # the slice of torch.distributed ShardedRun uses, on the gloo group
            @staticmethod
            def barrier():
                dist.barrier(group=host_group) if host_group is not dist else dist.barrier()

            @staticmethod
            def all_gather_object(out, obj):
                dist.all_gather_object(out, obj, group=host_group) if host_group is not dist else dist.all_gather_object(out, obj)

            @staticmethod
            def broadcast_object_list(lst, src=0):
                dist.broadcast_object_list(lst, src=src, group=host_group) if host_group is not dist else dist.broadcast_object_list(lst, src=src)

            @staticmethod
            def gather_object(obj, out, dst=0):
                dist.gather_object(obj, out, dst=dst, group=host_group) if host_group is not dist else dist.gather_object(obj, out, dst=dst)
        run = parallel.ShardedRun(_G, regions, args.chunk, loci, params, rank, world)
        shares = run.shares
        lo, hi = (0, args.sites) if rank == 0 else run.window()
        n_chunks_total = int(shares['chunks'].sum())
        buf, pitch = device_rows(lo, max(hi, lo + 1), local)
        seg = _lib.Segmenter(local)
        seg.set_betas_device(buf.data_ptr(), args.samples, pitch, max(hi - lo, 1), keepalive=buf)
        seg.set_loci(loci[lo:max(hi, lo + 1)])
        seg.set_site_base(lo)
        my_sites = int((run.ends[run.idx[rank]] - run.starts[run.idx[rank]])[run.idx[rank] < run.n_chunks].sum())
        own_acc = [None]

        def compute(starts, ends, off=None, out=None):
            flat, off = seg.segment_chunks_csr(starts - 1 - lo, (ends - starts).astype(np.int32), args.pcount, max_cpg, args.max_bp, out=out, off=off)
            own_acc[0] = seg.timings()                       # (of the rank's own items: rank 0's follow-up patches are not in it)
            return off, flat

        def patches(starts, ends):                            # rank 0: what the tree's rehearsal still misses (a hundred ~100-site patches)
            flat, off = seg.segment_chunks_csr(starts - 1 - lo, (ends - starts).astype(np.int32), args.pcount, max_cpg, args.max_bp)
            return off, flat

        def step():
            merged = run.step(compute, patches, copy=False)
            return merged, (run.last_stats or {})

        def timings():
            return own_acc[0]
    else:
        lo, hi = 0, args.sites
        n_chunks_total = len(parallel.chunk_grid(regions, args.chunk))
        buf, pitch = device_rows(lo, hi, local)
        seg = _lib.Segmenter(local)
        seg.set_betas_device(buf.data_ptr(), args.samples, pitch, hi - lo, keepalive=buf)
        seg.set_loci(loci)
        seg.set_site_base(0)
        st = np.array([a for a, _ in regions], dtype=np.int64)
        en = np.array([b for _, b in regions], dtype=np.int64)
        my_sites = args.sites

        def step():
            return seg.segment_regions(st, en, args.chunk, args.pcount, max_cpg, args.max_bp, copy=False)

        def timings():
            return seg.timings()
    torch.cuda.synchronize()

    def barrier():
        for d in range(ndev if group_mode else 0):
            torch.cuda.synchronize(d)
        torch.cuda.synchronize()
        if multi:
            dist.barrier() if oversub else dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    acc = None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, stats = step()
        acc = accumulate(acc, timings())
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device='cpu' if oversub else dev)
    if multi:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    # what ran where: one (host, device) per share / rank, so that the line can only state the devices it really used
    import socket
    host = socket.gethostname()
    if group_mode:
        placements, form = [(host, d) for d in devices], 'group'
    elif multi:
        gathered = [None] * world
        _G.all_gather_object(gathered, (host, local))
        placements, form = gathered, 'ranks'
    else:
        placements, form = [(host, local)], 'one'
    devrep = device_report(form, args.gpus, placements)
    devrep['devices_visible'] = ndev

    # N > 1: the same sharded step at x200 (BASELINE.json configs[3], the 8-GPU configuration: scoring-bound, so the shares' fixed costs weigh
    # least there), a few steps, timed like the main run (barrier, max over ranks)
    multi_rows = None
    if args.matrix and (multi or group_mode):
        multi_rows = []
        for ns in ([200] if args.samples != 200 else []):
            try:
                k2 = 3
                if multi:
                    b2, pitch2 = device_rows(lo, max(hi, lo + 1), local, samples=ns)
                    s2 = _lib.Segmenter(local)
                    s2.set_betas_device(b2.data_ptr(), ns, pitch2, max(hi - lo, 1), keepalive=b2)
                    s2.set_loci(loci[lo:max(hi, lo + 1)])
                    s2.set_site_base(lo)

                    def compute2(starts, ends, off=None, out=None):
                        flat, off = s2.segment_chunks_csr(starts - 1 - lo, (ends - starts).astype(np.int32), args.pcount, max_cpg, args.max_bp, out=out, off=off)
                        return off, flat

                    def patches2(starts, ends):
                        flat, off = s2.segment_chunks_csr(starts - 1 - lo, (ends - starts).astype(np.int32), args.pcount, max_cpg, args.max_bp)
                        return off, flat
                    step2 = lambda: run.step(compute2, patches2, copy=False)
                    closer = s2.close
                else:
                    g2 = _lib.SegmenterGroup(devices)
                    sh2 = g2.plan(loci, regions, args.chunk, args.pcount, max_cpg, args.max_bp)
                    keep2 = []
                    for d in range(args.gpus):
                        l2, h2 = int(sh2['win_lo'][d]), int(sh2['win_hi'][d])
                        if h2 > l2:
                            b2, pitch2 = device_rows(l2, h2, devices[d], samples=ns)
                            g2.share_set_device(d, b2.data_ptr(), ns, pitch2, keepalive=b2)
                    torch.cuda.set_device(local)
                    step2 = lambda: g2.segment_regions(copy=False)
                    closer = g2.close
                step2()
                barrier()
                t1 = time.perf_counter()
                for _ in range(k2):
                    r2 = step2()
                barrier()
                d2 = (time.perf_counter() - t1) / k2
                t2 = torch.tensor([d2], dtype=torch.float64, device='cpu' if oversub else dev)
                if multi:
                    dist.all_reduce(t2, op=dist.ReduceOp.MAX)
                d2 = float(t2.item())
                multi_rows.append({'samples': ns, 'steps': k2, 'ms_per_step': d2 * 1e3, 'value': args.sites / d2, 'unit': 'CpG-sites/s', 'n_gpus': devrep['n_gpus'], 'shares': devrep['shares']})
                closer()
                torch.cuda.empty_cache()
            except Exception as e:
                multi_rows.append({'samples': ns, 'failed': repr(e)})

    if rank == 0:
        res = [np.array(r) for r in res]          # (the lists are views into a buffer later calls reuse)
        n_blocks = int(sum(len(r) - 1 for r in res))
        block_sums = None
        if args.block_sums and seg is not None:
            # the immediate consumer of the borders: (#meth, #cov) of every block in every sample.
            # Algorithmic bytes = 2 * N * (sites covered); the D2H copy of the table is outside the kernel time.
            bs = np.concatenate([np.asarray(r[:-1], dtype=np.int64) for r in res]) - 1
            be = np.concatenate([np.asarray(r[1:], dtype=np.int64) for r in res]) - 1
            times = []
            for mode in (1, 1, 1, 1, 1, 3, 3, 0, 0):
                seg.block_sums(bs, be, mode=mode, min_cov=4)
                times.append(seg.last_block_sums_ms())
            covered = int((be - bs).sum())
            alg = 2 * covered * args.samples
            t_bin = min(times[:5])
            block_sums = {'kernel': 'k_block_sums_prep + k_block_sums_run + k_block_sums_direct (per block, per sample sums of meth/cov -> .bin rows; HIP events around the launches)',
                          'blocks': int(bs.size), 'ms_bin_rows': t_bin, 'ms_bin_rows_all': times[:5], 'ms_means': min(times[5:7]), 'ms_raw_sums': min(times[7:9]),
                          'algorithmic_bytes': alg, 'GB_per_s': alg / (t_bin * 1e-3) / 1e9, 'frac_of_hbm_peak': alg / (t_bin * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          # not credited above, reported beside it: the table the kernel has to write (2 B per block and sample as .bin rows, 8 B as
                          # means or raw sums) — with it the three output forms move 4.7-5.0 TB/s alike, which is why the raw sums (a third fewer
                          # instructions, four times the output) are not the fastest form
                          'written_bytes': {'bin_rows': 2 * int(bs.size) * args.samples, 'means': 8 * int(bs.size) * args.samples, 'raw_sums': 8 * int(bs.size) * args.samples},
                          'GB_per_s_reads_and_writes': {'bin_rows': (alg + 2 * int(bs.size) * args.samples) / (t_bin * 1e-3) / 1e9,
                                                        'means': (alg + 8 * int(bs.size) * args.samples) / (min(times[5:7]) * 1e-3) / 1e9,
                                                        'raw_sums': (alg + 8 * int(bs.size) * args.samples) / (min(times[7:9]) * 1e-3) / 1e9},
                          'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s'}

        # the prefix-sum pass PROPER (k_scan: per-sample prefix sums of (meth, cov), a carry per 128 sites, the validation) on this
        # workload's chunk grid.  A default-parameter job never runs it (its windows stay <= 252 sites: no wide tile reads carries, the
        # read-only k_validate is its scan pass, `roofline_scan` below); jobs with CpG islands / deep windows do.  north_star names
        # this pass, so it is timed here on the BASELINE workload with carries forced (wgbsseg_scan_only, want_carry = 1).
        scan_carries = None
        if args.scan_carries and seg is not None and not multi:
            grid = parallel.chunk_grid(regions, args.chunk)
            g_st = np.array([s0 - 1 for _, s0, _ in grid], dtype=np.int64)
            g_ln = np.array([e0 - s0 for _, s0, e0 in grid], dtype=np.int32)
            sc_ms, sc_bytes, sc_carry = seg.scan_only(g_st, g_ln, repeat=20, want_carry=True)
            trc = keyed_profile('*scan_traffic*.json', sha, kernel='k_scan', forced_carries=True)
            sc_traffic, sc_note = None, 'traffic: no PMC pass of this source state (csrc_sha %s) for k_scan with forced carries under profiles/' % sha
            if trc and abs(trc['algorithmic_bytes'] - sc_bytes) <= 0.001 * sc_bytes:
                sc_traffic = trc['traffic_bytes']
                sc_note = 'traffic (bytes per launch) = 2 x FETCH_SIZE + WRITE_SIZE from profiles/%s (gfx950 FETCH_SIZE x2 correction)' % trc['_file']
            scan_carries = {'kernel': 'k_scan with carries (coalesced 32 B per lane loads, per-sample prefix sums of (meth, cov) by SWAR + DPP wave scans, one carry per 128 sites '
                                      'staged in LDS and written as full-wavefront stores, meth<=cov validation): segmentor.cpp:164-190 + the scan of SURVEY (a-spec)',
                            'bound': 'hbm', 'achieved': sc_bytes / (sc_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': sc_bytes / (sc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'traffic': sc_traffic,
                            'algorithmic_bytes_per_launch': int(sc_bytes), 'carry_bytes_written_per_launch': int(sc_carry),
                            'traffic_over_algorithmic': None if sc_traffic is None else sc_traffic / sc_bytes,
                            'avg_launch_ms': sc_ms, 'launches_timed': 20, 'chunks': len(grid),
                            'note': 'HIP events around 20 back-to-back launches on the kernel\'s stream, after one warm-up launch; carries forced on the whole chunk grid of this '
                                    'workload (the step above does not need them); ' + sc_note}

        ms_step = dt / args.steps * 1e3
        value = args.sites / (dt / args.steps)
        # the scan pass: the launch of the batch that holds the chunks (one per step); the follow-up batches (a few hundred
        # ~100-site patches) launch it on kilobytes and are reported separately
        main_ms = acc['scan_main_ms'] / args.steps
        scan_gbs = acc['scan_main_bytes'] / (main_ms * 1e-3) / 1e9
        scan_all_gbs = acc['scan_bytes'] / (acc['scan_ms'] * 1e-3) / 1e9
        evals_s = acc['evals'] / (acc['cost_ms'] * 1e-3)
        stats_wide = acc['max_window'] > 252         # WG_MEDIUM_WMAX: wide scoring tiles exist, so the scan keeps its carries
        scan_kernel = 'k_scan' if stats_wide else 'k_validate'
        # HBM traffic of the scan launch from the PMC counters (separate rocprofv3 passes, tools/pmc_scan_traffic.py): only a file
        # made from THIS source state, for this kernel and these algorithmic bytes, is reported
        tr = keyed_profile('*scan_traffic*.json', sha, kernel=scan_kernel, forced_carries=False) if (not multi and not group_mode) else None
        traffic, traffic_note = None, 'traffic: no PMC pass of this source state (csrc_sha %s) for %s on this workload under profiles/' % (sha, scan_kernel)
        if tr and abs(tr['algorithmic_bytes'] - acc['scan_main_bytes']) <= 0.001 * acc['scan_main_bytes']:
            traffic = tr['traffic_bytes']
            traffic_note = ('traffic (bytes per launch) = 2 x FETCH_SIZE + WRITE_SIZE from profiles/%s (rocprofv3 PMC passes of this command on this '
                            'source state; gfx950 FETCH_SIZE x2 correction)' % tr['_file'])
        # Round 6: inside the step the scan pass no longer has the chip to itself — it starts with the batch on the lowest-priority stream, beside the
        # windows pass, the tile plan and the first scoring tiles — so its duration THERE says how well it hides, not how fast the kernel reads.  The
        # kernel's own rate (the roofline figure) is timed like `roofline_scan_carries`: 20 back-to-back launches on this workload's chunk grid.
        scan_in_step = {'avg_launch_ms': main_ms, 'GB/s': scan_gbs, 'frac_of_hbm_peak': scan_gbs / HBM_PEAK_GBS,
                        'note': 'HIP events around the launch INSIDE the timed steps, where it runs on the lowest-priority stream beside the windows pass, the tile plan and the '
                                'first scoring tiles (round 6: scoring begins 0.3 ms into the batch instead of 0.7)'}
        scan_alone_note = 'rank 0 / share 0, HIP events on the kernel stream inside the timed steps; '
        if seg is not None and not multi:
            grid_s = parallel.chunk_grid(regions, args.chunk)
            a_ms, a_bytes = seg.scan_only(np.array([s0 - 1 for _, s0, _ in grid_s], dtype=np.int64), np.array([e0 - s0 for _, s0, e0 in grid_s], dtype=np.int32),
                                          repeat=20, want_carry=stats_wide)[:2]
            if abs(a_bytes - acc['scan_main_bytes']) <= 0.001 * acc['scan_main_bytes']:
                main_ms, scan_gbs = a_ms, a_bytes / (a_ms * 1e-3) / 1e9
                scan_alone_note = 'rank 0; HIP events around 20 back-to-back launches on the kernel\'s stream over this workload\'s chunk grid, after one warm-up launch (the launch inside the step: `in_step`); '
        # instruction mix of the scoring kernel's evaluation (tools/micro/count_cost_loop.py --json), same keying
        mixes = {k: keyed_profile('*cost_isa_mix_%s.json' % k, sha) for k in ('narrow', 'narrow128', 'wide')}
        main_mix = mixes['narrow128' if args.samples <= 16 else 'narrow']
        issue = None
        if main_mix:
            cyc = main_mix['issue_cycles_per_eval']
            issue = {'cycles_per_eval_per_wavefront': cyc, 'valu_instr_per_eval': main_mix['valu_per_eval'], 'mix_per_eval': main_mix['mix_per_eval'],
                     'cycles_per_class': main_mix['cycles_per_class'], 'peak_evals_per_s': N_SIMD * 64 * CLOCK_HZ / cyc,
                     'frac': evals_s / (N_SIMD * 64 * CLOCK_HZ / cyc), 'file': main_mix['_file'],
                     'note': 'issue bound of the narrow-tile kernel\'s common path: sum over instruction classes of count x issue cycles per wavefront '
                             'instruction (fp32 2, fp64 / conversions / 3-operand integer 4, v_rcp_f32 8) on %d SIMDs at %.1f GHz; wide tiles (when the job '
                             'has any) run a longer evaluation: %s VALU' % (N_SIMD, CLOCK_HZ * 1e-9, mixes['wide']['valu_per_eval'] if mixes['wide'] else '?')}
        # the same bound with the issue rates MEASURED on this chip for a mixed stream (profiles/r03_valu_rates.json): in a stream that
        # alternates cheap (fp32 / simple integer) and fp64-class instructions every instruction issues at the fp64 rate, so the
        # evaluation costs (VALU instructions + the extra slots of v_rcp_f32) x the measured fp64 time per wavefront instruction
        if issue:
            try:
                vr = json.load(open(op.join(ROOT, 'profiles', 'r03_valu_rates.json')))['ns_per_wave_instr']
                t64, trcp = vr['v_fma_f64'] * 1e-9, vr['v_rcp_f32'] * 1e-9
                slots = main_mix['valu_per_eval'] + main_mix['mix_per_eval'].get('trans', 0) * (trcp / t64 - 1.0)
                issue['measured_mixed_stream'] = {'slots_per_eval': slots, 'ns_per_slot': t64 * 1e9, 'peak_evals_per_s': N_SIMD * 64 / (slots * t64),
                                                  'frac': evals_s / (N_SIMD * 64 / (slots * t64)), 'file': 'r03_valu_rates.json',
                                                  'note': 'every instruction of a mixed fp32 / fp64 stream issues at the fp64 rate measured on this chip (a stream alternating '
                                                          'v_fma_f32 and v_fma_f64 runs at 0.95 of the pure fp64 time per instruction); v_rcp_f32 at its own measured time'}
            except Exception:
                pass
            # round 4: the same question asked with a stream shaped like the evaluation itself (10 cheap, 1 v_rcp_f32, 29 fp64-class instructions in the
            # kernel's order, dependent chain, 5 unsynchronised wavefronts per SIMD): tools/micro/gen_valu_cluster.py, profiles/r04_valu_cluster.json —
            # clustering the cheap instructions (runs of 1 .. 16) does not change the rate, so this IS the issue bound of the mix
            try:
                vc = json.load(open(op.join(ROOT, 'profiles', 'r04_valu_cluster.json')))['rows']
                row = [r for r in vc if r['pattern'].startswith('evaluation-shaped:') and r['waves_per_simd'] == 5 and r['desync'] == 1][0]
                n_instr = 40.0                                                        # instructions of that stream per evaluation
                t_eval = row['ns_per_wave_instr'] * 1e-9 * n_instr * (main_mix['valu_per_eval'] / n_instr)
                issue['measured_evaluation_shaped_stream'] = {'ns_per_wave_instr': row['ns_per_wave_instr'], 'peak_evals_per_s': N_SIMD * 64 / t_eval,
                                                              'frac': evals_s / (N_SIMD * 64 / t_eval), 'file': 'r04_valu_cluster.json',
                                                              'note': 'a bare instruction stream of the evaluation\'s mix (no LDS, no branches, no per-block work) on another box of the pool; '
                                                                      'runs of 1 / 2 / 4 / 8 / 16 cheap instructions in a 1:3 mix all issue at 1.77-1.80 ns per instruction: clustering buys nothing'}
            except Exception:
                pass
        mode = devrep['sharding']
        cost_ms = acc['cost_ms'] / args.steps
        # HBM-side traffic of the scoring kernel's main launch from the PMC counters (tools/pmc_cost_traffic.py: FETCH_SIZE / WRITE_SIZE in separate
        # rocprofv3 passes, calibrated in the same passes on known byte counts): only a file of THIS source state and this workload is reported
        ct = keyed_profile('*cost_traffic*.json', sha) if (not multi and not group_mode) else None
        cost_traffic, cost_traffic_note = None, 'traffic: no PMC pass of this source state (csrc_sha %s) for k_cost on this workload under profiles/' % sha
        alg_cost_bytes = 2.0 * args.samples * args.sites + 8.0 * acc['pairs'] / args.steps
        if ct and abs(ct['algorithmic_bytes']['sum'] - alg_cost_bytes) <= 0.02 * alg_cost_bytes:
            cost_traffic = ct['traffic_bytes']
            cost_traffic_note = ('traffic (HBM-side bytes of the main launch) = %.3f x the algorithmic bytes (every beta byte once + one double per scored block): %.2f GB read '
                                 '(%.2f x the beta bytes: the halo re-reads of neighbouring tiles stay in L2) + %.2f GB written; FETCH_SIZE / WRITE_SIZE of profiles/%s, '
                                 'calibrated there on 1 GiB read / written once at the kernel\'s access widths'
                                 % (ct['traffic_over_algorithmic'], ct['read_bytes'] / 1e9, ct['read_over_beta_bytes'], ct['write_bytes'] / 1e9, ct['_file']))
        out = {
            'metric': 'CpG-sites/sec segmented',
            'value': value, 'unit': 'CpG-sites/s', 'n_gpus': devrep['n_gpus'], 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_step, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'u8 counts -> u32 prefix sums -> f32/f64 log-likelihood (bit-exact with the reference)',
            'data': 'synthetic (seeded hg19-shaped genome and betas, generated on the device)',
            'config': {'workload': 'hg19-shaped %d CpGs x %d betas, whole-genome segment, chunk_size %d, max_cpg %d, max_bp %d, pcount %g'
                                   % (args.sites, args.samples, args.chunk, args.max_cpg, args.max_bp, args.pcount) + (' + CpG islands in the loci' if args.islands else ''),
                       'baseline_config': 'BASELINE.json configs[2]' if (args.sites, args.samples, args.islands) == (28217448, 32, False) else 'custom',
                       'chunks': n_chunks_total, 'chromosomes': len(sizes), 'sharding': mode,
                       'gpus_requested': devrep['gpus_requested'], 'shares': devrep['shares'], 'distinct_devices': devrep['distinct_devices'],
                       'devices_visible': devrep['devices_visible'], 'oversubscribed': devrep['oversubscribed'],
                       'share_chunks': None if shares is None else [int(x) for x in shares['chunks']],
                       'share_work': None if shares is None else [int(x) for x in shares['work']],
                       # how even the shares are: the largest share's work over the mean (1.0 = perfectly even; rank 0 is given less on purpose in the
                       # multi-process form: it also runs the one stitching tree)
                       'share_work_max_over_mean': None if shares is None else float(max(shares['work'])) / (float(sum(shares['work'])) / len(shares['work'])),
                       'rank0_sites': my_sites, 'rank0_stats': stats, 'rank0_blocks': n_blocks, 'csrc_sha': sha},
            # the kernel that IS the step.  Not HBM- and not MFMA-bound (integer scan + scalar cost on the vector ALUs): priced in
            # algorithmic flops against the vector-fp64 peak, with the instruction-issue bound of its real mix beside it.
            'roofline': {'kernel': 'k_cost (block log-likelihoods: %.0f %% of the step)' % (100 * cost_ms / ms_step), 'bound': 'valu (vector fp64 peak; no MFMA: not a contraction)',
                         'achieved': evals_s * FLOP_PER_EVAL / 1e12, 'peak': FP64_VALU_PEAK / 1e12, 'unit': 'TFLOP/s',
                         'frac': evals_s * FLOP_PER_EVAL / FP64_VALU_PEAK, 'traffic': cost_traffic,
                         'algorithmic_flop_per_eval': FLOP_PER_EVAL, 'evals_per_launch': acc['evals'] / args.steps, 'avg_launch_ms': cost_ms,
                         'launches_timed': args.steps, 'evals_per_s': evals_s,
                         'fp64_flop_executed_per_eval': FP64_FLOP_EXECUTED, 'fp64_executed_frac_of_peak': evals_s * FP64_FLOP_EXECUTED / FP64_VALU_PEAK,
                         'issue': issue, 'division_core': '4 instructions, verified on the device for this pseudo count' if acc.get('div_short') else '8 instructions',
                         'pairs_per_step': acc['pairs'] / args.steps, 'max_window': acc['max_window'], 'stages': acc['n_stages'],
                         'algorithmic_bytes_per_launch': alg_cost_bytes,
                         'note': 'rank 0 / share 0; HIP events on the scoring stream inside the timed steps; ' + cost_traffic_note + ' — the kernel moves ~0.3 TB/s: '
                                 'not a memory kernel, its bound is VALU issue'},
            'roofline_scan': {'kernel': ('k_scan (per-sample prefix scan -> 128-site carries + meth<=cov validation: the job has wide tiles)' if stats_wide else
                                         'k_validate (the scan pass of a job without wide tiles: every beta byte read once, meth<=cov checked; no carries needed)'), 'bound': 'hbm',
                              'achieved': scan_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': scan_gbs / HBM_PEAK_GBS,
                              'traffic': traffic,
                              'algorithmic_bytes_per_launch': acc['scan_main_bytes'], 'avg_launch_ms': main_ms,
                              'launches_timed': args.steps, 'in_step': scan_in_step,
                              'all_launches': {'count': acc['scan_launches'], 'bytes': acc['scan_bytes'], 'ms': acc['scan_ms'],
                                               'GB/s': scan_all_gbs},
                              'note': scan_alone_note + traffic_note},
            'roofline_scan_carries': scan_carries,
            'block_sums': block_sums,
            'device_ms_per_step': {k: acc[k] / args.steps for k in ('scan_ms', 'window_ms', 'cost_ms', 'dp_ms', 'trace_ms', 'total_ms')},
        }
        single = not multi and not group_mode
        if multi_rows is not None:
            out['matrix'] = {'what': 'the same sharded whole-genome step at x200 (BASELINE.json configs[3]), timed like the main run (barrier, max over ranks)', 'rows': multi_rows}
        if single and args.extras:
            try:
                out['extras'] = extras(args, seg, loci, sizes, names, res)
            except Exception as e:
                out['extras'] = {'failed': repr(e)}
        if single and args.e2e:
            try:
                out['end_to_end'] = end_to_end(args, buf, sizes, names, loci)
            except Exception as e:
                out['end_to_end'] = {'value': None, 'what': 'failed: %r' % (e,)}
        if single and args.cpu_seconds > 0:
            try:
                out['cpu_baseline'] = cpu_baseline(args, buf, sizes, loci, seg, params)
                out['cpu_baseline']['gpu_over_cpu'] = value / out['cpu_baseline']['value']
            except Exception as e:                       # the baseline must never break the bench line
                out['cpu_baseline'] = {'value': None, 'unit': 'CpG-sites/s', 'cores': os.cpu_count(), 'kind': 'reference',
                                       'sample': 'failed: %r' % (e,)}
        if single and args.matrix:
            # the other sample counts of the metric, same genome and parameters, a few steps each (timed like the main run)
            rows = []
            # the 8-GPU form of the product on this one GPU: the same genome as a share group of 8 with all shares on this device (each share's launches
            # are those of a GPU's share; the shares' host threads run them concurrently here).  Borders must equal the one-context result.
            # (the one-GPU context goes first: what `wgbstools segment --gpus 8` has in its process is the group alone — and a context's lowest-priority scan
            # stream, even idle, changes how the runtime spreads the group's 32 streams over its hardware queues: 34 -> 38-40 ms per step for this row)
            seg.close()
            seg = None
            try:
                g8 = _lib.SegmenterGroup([local] * 8)
                w8 = g8.plan(loci, regions, args.chunk, args.pcount, max_cpg, args.max_bp)
                for d in range(8):
                    g8.share_set_device(d, int(buf.data_ptr()) + 2 * int(w8['win_lo'][d]), args.samples, pitch, keepalive=buf)
                r8, _s8 = g8.segment_regions(copy=False)
                same = len(r8) == len(res) and all(np.array_equal(a, b) for a, b in zip(r8, res))
                torch.cuda.synchronize()
                k8, t1 = 5, time.perf_counter()
                for _ in range(k8):
                    g8.segment_regions(copy=False)
                torch.cuda.synchronize()
                d8 = (time.perf_counter() - t1) / k8
                rows.append({'samples': args.samples, 'shares_on_this_gpu': 8, 'steps': k8, 'ms_per_step': d8 * 1e3, 'ms_per_share': d8 * 1e3 / 8,
                             'value': args.sites / d8, 'unit': 'CpG-sites/s', 'borders_equal_one_context': bool(same),
                             'share_work_max_over_mean': float(max(w8['work'])) / (float(sum(w8['work'])) / 8),
                             'what': 'the product\'s 8-GPU form (a share group of 8: work-balanced chunk runs, one host thread per share, ONE host-side tree) with every '
                                     'share on this one GPU: checks that it gives the one-context borders, and times it.  The shares\' launches run concurrently here, '
                                     'so ms_per_share = step / 8 is a THROUGHPUT figure; a share alone on its own GPU also pays its front, stage drains and recurrence '
                                     'tail unhidden (bench.py --sites 3527181: ~4.4 ms, profiles/r04_bench_one_eighth.json)'})
                g8.close()
                del g8
            except Exception as e:
                rows.append({'samples': args.samples, 'shares_on_this_gpu': 8, 'failed': repr(e)})
            buf = None
            torch.cuda.empty_cache()
            for ns in MATRIX_SAMPLES:
                if ns == args.samples:
                    continue
                try:
                    b2, pitch2 = device_rows(0, args.sites, local, samples=ns)
                    s2 = _lib.Segmenter(local)
                    s2.set_betas_device(b2.data_ptr(), ns, pitch2, args.sites, keepalive=b2)
                    s2.set_loci(loci)
                    k = 5 if ns <= 32 else 3 if ns <= 200 else 2
                    s2.segment_regions(st, en, args.chunk, args.pcount, max_cpg, args.max_bp, copy=False)
                    torch.cuda.synchronize()
                    a2, t1 = None, time.perf_counter()
                    for _ in range(k):
                        r2, _st = s2.segment_regions(st, en, args.chunk, args.pcount, max_cpg, args.max_bp, copy=False)
                        a2 = accumulate(a2, s2.timings())
                    torch.cuda.synchronize()
                    d2 = (time.perf_counter() - t1) / k
                    ev2 = a2['evals'] / (a2['cost_ms'] * 1e-3)
                    rows.append({'samples': ns, 'steps': k, 'ms_per_step': d2 * 1e3, 'value': args.sites / d2, 'unit': 'CpG-sites/s',
                                 'blocks': int(sum(len(r) - 1 for r in r2)),
                                 'cost_ms': a2['cost_ms'] / k, 'dp_ms': a2['dp_ms'] / k, 'scan_ms': a2['scan_main_ms'] / k,
                                 'evals_per_s': ev2, 'roofline_frac': ev2 * FLOP_PER_EVAL / FP64_VALU_PEAK,
                                 'scan_GB_per_s': a2['scan_main_bytes'] / (a2['scan_main_ms'] / k * 1e-3) / 1e9,
                                 'scan_frac_of_hbm_peak': a2['scan_main_bytes'] / (a2['scan_main_ms'] / k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 'scan_timed': 'inside the step: on the lowest-priority stream beside the windows pass, the tile plan and the first scoring tiles (not the kernel alone)',
                                 'beta_GB_resident': ns * pitch2 / 1e9})
                    s2.close()
                    del s2
                    if ns == 200 and args.e2e:
                        # the atlas-scale cohort end to end: 11.3 GB of page-cached files -> HBM -> BED; the upload is longer than the compute
                        try:
                            rows[-1]['end_to_end'] = end_to_end(args, b2, sizes, names, loci, samples=ns, reps=2)
                        except Exception as e:
                            rows[-1]['end_to_end'] = {'value': None, 'what': 'failed: %r' % (e,)}
                    del b2
                    torch.cuda.empty_cache()
                except Exception as e:
                    rows.append({'samples': ns, 'failed': repr(e)})
            # the genome WITH CpG islands (windows up to ~250 sites and a few beyond: narrow + medium + wide tiles, k_scan with carries, the
            # 32-step recurrence), x32: the shape of a real genome, one row
            try:
                loci_i = synth.synth_loci(SEED, sizes, islands=True)
                b2, pitch2 = device_rows(0, args.sites, local, samples=32)
                s2 = _lib.Segmenter(local)
                s2.set_betas_device(b2.data_ptr(), 32, pitch2, args.sites, keepalive=b2)
                s2.set_loci(loci_i)
                k = 5
                s2.segment_regions(st, en, args.chunk, args.pcount, max_cpg, args.max_bp, copy=False)
                torch.cuda.synchronize()
                a2, t1 = None, time.perf_counter()
                for _ in range(k):
                    r2, _st = s2.segment_regions(st, en, args.chunk, args.pcount, max_cpg, args.max_bp, copy=False)
                    a2 = accumulate(a2, s2.timings())
                torch.cuda.synchronize()
                d2 = (time.perf_counter() - t1) / k
                ev2 = a2['evals'] / (a2['cost_ms'] * 1e-3)
                rows.append({'samples': 32, 'islands': True, 'steps': k, 'ms_per_step': d2 * 1e3, 'value': args.sites / d2, 'unit': 'CpG-sites/s',
                             'blocks': int(sum(len(r) - 1 for r in r2)), 'max_window': a2['max_window'],
                             'cost_ms': a2['cost_ms'] / k, 'dp_ms': a2['dp_ms'] / k, 'scan_ms': a2['scan_main_ms'] / k,
                             'scan_kernel': 'k_scan (carries)' if a2['max_window'] > 252 else 'k_validate',
                             'evals_per_s': ev2, 'roofline_frac': ev2 * FLOP_PER_EVAL / FP64_VALU_PEAK,
                             'scan_GB_per_s': a2['scan_main_bytes'] / (a2['scan_main_ms'] / k * 1e-3) / 1e9,
                             'scan_frac_of_hbm_peak': a2['scan_main_bytes'] / (a2['scan_main_ms'] / k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 'scan_timed': 'inside the step: on the lowest-priority stream beside the windows pass, the tile plan and the first scoring tiles (not the kernel alone)'})
                s2.close()
                del b2, s2
                torch.cuda.empty_cache()
            except Exception as e:
                rows.append({'samples': 32, 'islands': True, 'failed': repr(e)})
            out['matrix'] = {'what': 'the same whole-genome step at the other sample counts of the metric (inputs resident, 1 warm-up, timed with '
                                     'synchronize + perf_counter around the steps like the main run)', 'rows': rows}
        print(json.dumps(out), flush=True)
    if seg is not None:
        seg.close()
    if grp is not None:
        grp.close()
    if multi:
        dist.barrier() if oversub else dist.barrier(device_ids=[local])
        if run is not None:
            run.close()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
