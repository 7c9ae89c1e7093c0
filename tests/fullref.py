"""TEST INFRASTRUCTURE: a whole genome on the HIP path against the REFERENCE BINARY (oracle/_ref/segmentor).

`whole_genome_vs_reference()` is what both `tests/test_gpu_fullsize.py::test_hg19_x200_atlas_scale` (inside `pytest -m gpu`)
and `tools/full_vs_reference.py` (the same check from a shell, other cohorts / worlds) run:

  1. `wgbsseg_segment_regions` over the regions (the product call);
  2. per region: every chunk of the reference's grid (segment.py:124-135) through `wgbsseg_segment_chunks`, then the
     reference's pairwise stitching tree (segment.py:157-165,199-252; tests/reftree.py, pinned by vectors of the reference's own
     driver) over those chunk DPs, every patch it asks for computed by the HIP path — the stitched list must equal (1);
  3. EVERY range the trees touched (all chunks + all junction patches) through the reference binary (segmentor.cpp:60-159), one
     single-threaded process per range, a pool of them over the host's CPUs, compared border by border with what the HIP path
     returned for that range.

Not imported by the product.
"""
import os
import os.path as op
import queue
import shutil
import subprocess
import tempfile
import threading
import time

import numpy as np

from oracle import oracle


def ref_on_ranges(buf, loci, ranges, pcount, max_cpg, max_bp, procs=None):
    """The reference binary on 0-based site ranges [(start0, n), ...] of the device-resident betas `buf` ([N][pitch] uint8 torch
    tensor): every range's bytes are written as per-sample .beta files (what `segmentor` reads, segmentor.cpp:164-177) and one
    single-threaded process per range runs with the loci on stdin (segment.py:48-55 minus tabix).  A pool of `procs` workers: each
    takes the next range, copies its bytes off the device, writes the files, runs the process, parses its stdout and removes the
    files — at most `procs` ranges sit in /dev/shm at any time.  -> {(start0, n): int64 borders}"""
    assert oracle.have_ref(), 'oracle/_ref/segmentor did not travel with the snapshot'
    procs = max(1, min(procs or (os.cpu_count() or 8), len(ranges) or 1))
    out, errors = {}, []
    td = tempfile.mkdtemp(dir='/dev/shm' if op.isdir('/dev/shm') else None)
    todo = queue.SimpleQueue()
    for r in ranges:
        todo.put(r)
    grab = threading.Lock()                      # one device -> host copy at a time (they share one DMA queue anyway)

    def worker():
        while not errors:
            try:
                st, n = todo.get_nowait()
            except queue.Empty:
                return
            d = op.join(td, 'r%d_%d' % (st, n))
            try:
                with grab:
                    host = buf[:, 2 * st:2 * (st + n)].cpu().numpy()
                os.mkdir(d)
                paths = []
                for s in range(host.shape[0]):
                    p = op.join(d, 's%04d.beta' % s)
                    host[s].tofile(p)
                    paths.append(p)
                del host
                cmd = [oracle.REF_BIN] + paths + ['-s', '0', '-n', str(n), '-max_cpg', str(max_cpg), '-ps', repr(float(pcount)),
                                                 '-max_bp', str(max_bp)]
                stdin = ('\n'.join(map(str, loci[st:st + n].tolist())) + '\n').encode()
                r = subprocess.run(cmd, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                if r.returncode != 0:
                    raise RuntimeError('reference binary failed on [%d, +%d): %s' % (st, n, r.stderr.decode()[-500:]))
                out[(st, n)] = np.array(r.stdout.split(), dtype=np.int64)
            except BaseException as e:          # noqa: B902 — handed to the caller's thread below
                errors.append(e)
            finally:
                shutil.rmtree(d, ignore_errors=True)

    try:
        th = [threading.Thread(target=worker) for _ in range(procs)]
        [t.start() for t in th]
        [t.join() for t in th]
    finally:
        shutil.rmtree(td, ignore_errors=True)
    if errors:
        raise errors[0]
    return out


class Recorder:
    """chunk engine over the HIP path that remembers every range it was asked for (1-based half-open)."""

    def __init__(self, seg, pcount, max_cpg, max_bp):
        self.seg, self.p = seg, (pcount, max_cpg, max_bp)
        self.asked = {}

    def segment_many(self, sites_list, params):
        st0 = [a - 1 for a, _ in sites_list]
        ln = [b - a for a, b in sites_list]
        res = self.seg.segment_chunks(st0, ln, *self.p)
        outs = []
        for (a, b), r in zip(sites_list, res):
            r = r.astype(np.int64) + a
            self.asked[(a, b)] = r
            outs.append(r)
        return outs


def grid(a, e, chunk):
    bords = list(range(a, e, chunk)) + [e]
    return list(zip(bords[:-1], bords[1:]))


def tree(chunks, eng):
    """segment.py:157-165 over stitch_2_dfs (segment.py:199-232), junction by junction (tests/reftree.py)."""
    import reftree
    return reftree.tree(chunks, lambda sites: eng.segment_many(sites, {}))


def pool_size(n_samples, chunk, procs=0):
    """reference processes at a time: all logical CPUs, but a pool's ranges sit in /dev/shm as .beta files while they run — keep
    them under half of what is free there."""
    procs = procs or (os.cpu_count() or 8)
    shm = '/dev/shm' if op.isdir('/dev/shm') else tempfile.gettempdir()
    free = shutil.disk_usage(shm).free
    return int(max(1, min(procs, (free // 2) // max(1, 2 * n_samples * chunk)))), shm, free


def whole_genome_vs_reference(seg, buf, loci, regions, chunk, pcount, max_cpg, max_bp, res, which=None, names=None, procs=0,
                              log=lambda *a: None):
    """Steps 2 and 3 of the module text for the stitched result `res` of `seg.segment_regions` over `regions` (1-based half-open).
    -> dict of counts; 'differences' == 0 means bit-identical everywhere."""
    t0 = time.time()
    which = list(range(len(regions))) if which is None else list(which)
    names = names or ['region %d' % i for i in range(len(regions))]
    n_samples = int(buf.shape[0])
    bad, asked, grid_all, n_chunks, n_patches, chrom_ok = [], {}, set(), 0, 0, 0
    for ri in which:
        a, e = regions[ri]
        eng = Recorder(seg, pcount, max_cpg, max_bp)
        g = grid(a, e, chunk)
        want = tree(eng.segment_many(g, {}), eng)
        same = np.array_equal(np.asarray(res[ri], dtype=np.int64), want)
        chrom_ok += bool(same)
        if not same:
            bad.append('stitched borders of %s differ from the reference tree' % names[ri])
        gs = set(g)
        grid_all |= gs
        n_chunks += len(g)
        n_patches += sum(1 for k in eng.asked if k not in gs)
        asked.update(eng.asked)
    log('[2] stitched regions identical to the reference tree: %d/%d (%d chunks, %d distinct junction patches asked for; %.1f s)'
        % (chrom_ok, len(which), n_chunks, n_patches, time.time() - t0))
    # longest first: the chunks fill the cores, the patches the gaps
    ranges = sorted(((a - 1, b - a) for (a, b) in asked), key=lambda r: -r[1])
    procs, shm, free = pool_size(n_samples, chunk, procs)
    log('    %s has %.1f GB free: %d reference processes at a time' % (shm, free / 1e9, procs))
    t1 = time.time()
    ref = ref_on_ranges(buf, loci, ranges, pcount, max_cpg, max_bp, procs=procs)
    ch_ok = ch_n = pa_ok = pa_n = 0
    for (a, b), r in asked.items():
        same = np.array_equal(r - a, ref[(a - 1, b - a)])
        if not same:
            bad.append('range [%d, %d) differs from the reference binary' % (a, b))
        if (a, b) in grid_all:
            ch_n += 1
            ch_ok += bool(same)
        else:
            pa_n += 1
            pa_ok += bool(same)
    ref_s = time.time() - t1
    log('[3] against the reference binary: %d/%d chunks identical, %d/%d patches identical (%d processes at a time, %.0f s)'
        % (ch_ok, ch_n, pa_ok, pa_n, procs, ref_s))
    return {'chromosomes_identical': chrom_ok, 'chromosomes': len(which), 'chunks_identical': ch_ok, 'chunks': ch_n,
            'patches_identical': pa_ok, 'patches': pa_n, 'differences': len(bad), 'different': bad[:20],
            'reference_seconds': ref_s, 'reference_procs': procs, 'host_cpus': os.cpu_count(), 'check_wall_s': time.time() - t0}
