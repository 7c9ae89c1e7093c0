"""Multi-process sharding of the segment path: one process per GPU (python -m torch.distributed.run), no collective on
the data path.

Chunks are independent DP problems and chromosomes never interact (segment.py:84-86,129-134).  The chunk grid is cut into
`world` contiguous runs of chunks by the planner of the share groups (wgbsseg_plan_shares[_weighted], include/wgbsseg.h:
balanced by the scored blocks the chunks hold, always on the reference's grid so every chunk is the very chunk a one-GPU run
would segment).  Each rank runs, on its own GPU, the chunk DPs of its run and the junction patches the stitcher is known to
ask for first (ShardedRun); the border lists reach rank 0 through slots in /dev/shm (NodeSlots; a gather of Python objects
between hosts), and rank 0 walks the reference's pairwise tree (segment.py:157-165,199-252) over ALL chunks with the native
stitcher (wgbsseg_stitch_regions): the answer does not depend on the number of ranks.
"""


def regions_of_sizes(sizes):
    """Whole chromosomes as 1-based half-open CpG ranges (segment.py:115-119)."""
    out, pos = [], 1
    for sz in sizes:
        out.append((pos, pos + int(sz)))
        pos += int(sz)
    return out


def chunk_grid(regions, chunk):
    """[(region idx, start, end)] 1-based half-open: the reference's grid (segment.py:124-135) over the regions."""
    chunks = []
    for ri, (a, b) in enumerate(regions):
        for s in range(int(a), int(b), chunk):
            chunks.append((ri, s, min(s + chunk, int(b))))
    return chunks


def plan(regions, chunk, world, loci, params):
    """wgbsseg_plan_shares for `world` shares -> dict of arrays (own_lo, own_hi, win_lo, win_hi, chunks, work)."""
    from . import _lib
    return _lib.plan_shares(loci, regions, chunk, params['pcount'], params['max_cpg'], params['max_bp'], world)


def plan_weighted(regions, chunk, world, loci, params, weights):
    """wgbsseg_plan_shares_weighted: contiguous runs of chunks whose work is proportional to `weights` (equal weights: `plan`)."""
    from . import _lib
    return _lib.plan_shares(loci, regions, chunk, params['pcount'], params['max_cpg'], params['max_bp'], world, weights=list(weights))


def chunks_of_rank(regions, chunk, world, rank, loci, params, shares=None):
    """The chunks (1-based half-open) rank `rank` of `world` segments: a contiguous run of the reference's chunk grid."""
    sh = shares or plan(regions, chunk, world, loci, params)
    lo, hi = int(sh['own_lo'][rank]), int(sh['own_hi'][rank])
    return [(s, e) for _, s, e in chunk_grid(regions, chunk) if hi > lo and lo <= s - 1 and e - 1 <= hi]


def pieces_of_rank(regions, chunk, world, rank, loci, params, shares=None):
    """The same chunks as maximal runs inside one region: [(region idx, start, end)], every piece starting on the
    region's chunk grid (what a rank hands to wgbsseg_segment_regions when it times its own share: bench.py)."""
    sh = shares or plan(regions, chunk, world, loci, params)
    lo, hi = int(sh['own_lo'][rank]), int(sh['own_hi'][rank])
    out = []
    for ri, s, e in chunk_grid(regions, chunk):
        if not (hi > lo and lo <= s - 1 and e - 1 <= hi):
            continue
        if out and out[-1][0] == ri and out[-1][2] == s:
            out[-1] = (ri, out[-1][1], e)
        else:
            out.append((ri, s, e))
    return out


def gather_to_rank0(local, rank, world):
    """Host-side gather of per-rank chunk results (small int arrays); no-op for world == 1."""
    if world == 1:
        return [local]
    import torch.distributed as dist
    out = [None] * world if rank == 0 else None
    dist.gather_object(local, out, dst=0)
    return out


def env_rank_world():
    """(rank, world, local_rank) from the torch.distributed.run environment; (0, 1, 0) outside it."""
    import os
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def init_host_group():
    """The only communication of a multi-process segment run is the final host-side gather of border lists: a gloo group."""
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group('gloo')
    return dist


# ------------------------------------------------------------------------------------------------------------
# One process per GPU, the fast way: the first batch of the native stitcher is cut by owner
# ------------------------------------------------------------------------------------------------------------
def item_owners(starts, n_chunks, shares):
    """Owner rank of every item of the stitcher's first batch (wgbsseg_first_batch_items: chunks, then junction patches): a
    chunk belongs to the share that the planner gave it to, a patch to the owner of the chunk its first site lies in (its
    other end stays inside that share's halo: patches planned up front span at most 200 sites)."""
    import numpy as np
    lo, hi = np.asarray(shares['own_lo']), np.asarray(shares['own_hi'])
    cs = np.asarray(starts[:n_chunks]) - 1                      # 0-based first sites of the chunks, ascending
    holders = np.flatnonzero(hi > lo)
    rank_of_chunk = holders[np.searchsorted(lo[holders], cs, side='right') - 1]
    owner = np.empty(len(starts), dtype=np.int64)
    owner[:n_chunks] = rank_of_chunk
    owner[n_chunks:] = rank_of_chunk[np.searchsorted(cs, np.asarray(starts[n_chunks:]) - 1, side='right') - 1]
    return owner


class NodeSlots:
    """Hand-over of the ranks' border lists to rank 0.  On one node: every rank owns two slots (files under /dev/shm that rank 0 maps
    as well; steps alternate between them) and NO collective is needed per step: a rank stamps its slot with the step's number
    when its lists are complete, rank 0 waits for the stamps, and stamps its own header with the number of steps it has consumed —
    which is what a rank waits for before it reuses a slot (it may run one step ahead of rank 0's stitching, no further).
    Ranks on several hosts fall back to a gather of Python objects.
    The slot FILES are unlinked as soon as every rank has mapped them (the mappings stay valid): a run that crashes or is killed
    leaks nothing under /dev/shm.  A waiting rank checks every tenth of a second that the ranks it waits for are still alive (their
    process ids are exchanged once; only when all ranks share one pid namespace) and gives up at once when one is gone, after
    TIMEOUT_S otherwise (TIMEOUT_NO_PROBE_S when the probe cannot be used; WGBSSEG_SLOT_TIMEOUT_S sets it).
    Memory ordering: a slot's stamp is a plain 8-byte store issued after the stores of the lists it covers, and the reader loads the
    stamp before the lists; both rest on x86-64's total store order / ordered loads (the hosts of MI355X nodes) — on a weakly
    ordered host the stamp would need a release store and an acquire load.
    Slot layout: int64 header [8] (0: step stamp; rank 0 only, 1: steps consumed), int64 offsets [items + 1], int32 borders [cap]."""
    HDR = 64
    TIMEOUT_S = 120.0                 # with the liveness probe (a dead peer is noticed within 0.1 s; this only catches a hung one)
    TIMEOUT_NO_PROBE_S = 600.0        # without it; WGBSSEG_SLOT_TIMEOUT_S overrides both

    @staticmethod
    def _pid_namespace():
        """What tells two ranks that they share a machine AND a process table: the boot id of the kernel they run on (the initial pid namespace
        has the same inode on every Linux host, and two machines may share a default hostname: ADVICE r05) + the pid namespace."""
        import os
        try:
            with open('/proc/sys/kernel/random/boot_id') as f:
                boot = f.read().strip()
            return boot + ' ' + os.readlink('/proc/self/ns/pid')
        except OSError:
            return None

    @staticmethod
    def _boot_id():
        try:
            with open('/proc/sys/kernel/random/boot_id') as f:
                return f.read().strip()
        except OSError:
            return None

    @classmethod
    def _timeout_from_env(cls, default):
        import os
        v = os.environ.get('WGBSSEG_SLOT_TIMEOUT_S')
        if v is None or v == '':
            return default
        try:
            t = float(v)
        except ValueError:
            raise ValueError('WGBSSEG_SLOT_TIMEOUT_S must be a number of seconds, not %r' % v) from None
        if not t > 0:
            raise ValueError('WGBSSEG_SLOT_TIMEOUT_S must be positive, not %r' % v)
        return t

    def __init__(self, dist, rank, world, n_items, caps):
        import os
        import socket
        import uuid
        import numpy as np
        self.dist, self.rank, self.world = dist, rank, world
        hosts = [None] * world
        # (the timeout travels too: every rank waits as long as rank 0 says, whatever its own environment holds)
        dist.all_gather_object(hosts, (socket.gethostname() + ' ' + str(self._boot_id()), os.getpid(), self._pid_namespace()))
        self.pids = [p for _, p, _ in hosts]
        # the liveness probe (signal 0 to a peer's pid) means something only when every rank lives in ONE pid namespace: ranks in separate
        # containers that share a hostname and /dev/shm would see a stranger's pid, or none, and report a live peer as gone
        spaces = set(ns for _, _, ns in hosts)
        self.liveness = len(spaces) == 1 and None not in spaces
        # without the probe a dead peer shows only as a timeout, and a slow step (a large cohort, the plain path of disordered chunks) must
        # not: the short timeout is kept for runs that can tell the two apart
        tmo = [self._timeout_from_env(self.TIMEOUT_S if self.liveness else self.TIMEOUT_NO_PROBE_S) if rank == 0 else None]
        dist.broadcast_object_list(tmo, src=0)
        self.timeout_s = float(tmo[0])
        hosts = [h for h, _, _ in hosts]
        tag = [uuid.uuid4().hex[:12] if rank == 0 else None]
        dist.broadcast_object_list(tag, src=0)
        self.shared = len(set(hosts)) == 1 and os.path.isdir('/dev/shm') and not os.environ.get('WGBSSEG_NO_SHM')
        self.n_items, self.caps = [int(x) for x in n_items], [int(x) for x in caps]
        self.paths, self.maps, self.step_no = [], {}, 0
        if self.shared:
            def path(r, k):
                return '/dev/shm/wgbsseg_%s_r%d_%d.bin' % (tag[0], r, k)
            # the slots are sized for the worst case (every site a border) and RESERVED, not just mapped: a /dev/shm too small for
            # them (containers often get 64 MB) must show now, as an error every rank can agree on, not later as a SIGBUS in the
            # middle of a step.  One rank that cannot reserve sends all of them to the gather of objects.
            ok = True
            try:
                for k in range(2):
                    pth = path(rank, k)
                    self.paths.append(pth)
                    fd = os.open(pth, os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
                    try:
                        self._reserve(fd, self._bytes(rank))                    # zero-filled: stamp 0 = nothing yet
                    finally:
                        os.close(fd)
            except OSError:
                ok = False
            oks = [None] * world
            dist.all_gather_object(oks, ok)
            if not all(oks):
                self._unlink()
                self.shared = False
        if self.shared:
            for r in range(world):
                for k in range(2):
                    if rank == 0 or r == rank or (r == 0 and k == 0):                             # (everybody reads rank 0's header)
                        self.maps[(r, k)] = np.memmap(path(r, k), dtype=np.uint8, mode='r+', shape=(self._bytes(r),))
            dist.barrier()
            self._unlink()                                       # every rank holds its mappings: the names can go (nothing to leak on a crash)

    @staticmethod
    def _reserve(fd, size):
        import os
        os.posix_fallocate(fd, 0, size)

    def _unlink(self):
        import os
        for pth in self.paths:
            try:
                os.unlink(pth)
            except OSError:
                pass
        self.paths = []

    def _bytes(self, r):
        return self.HDR + 8 * (self.n_items[r] + 1) + 4 * max(1, self.caps[r]) + 8

    def _hdr(self, r, k):
        import numpy as np
        return self.maps[(r, k)][:self.HDR].view(np.int64)

    def _views(self, r, k):
        import numpy as np
        m = self.maps[(r, k)]
        n = self.n_items[r]
        o = self.HDR
        return m[o:o + 8 * (n + 1)].view(np.int64), m[o + 8 * (n + 1):o + 8 * (n + 1) + 4 * max(1, self.caps[r])].view(np.int32)

    def _wait(self, ready, what, peer):
        import os
        import time
        t0 = time.monotonic()
        spins, checked = 0, t0
        while not ready():
            spins += 1
            if spins > 2000:
                time.sleep(0.00002)
                now = time.monotonic()
                if now - checked > 0.1:
                    checked = now
                    if self.liveness:
                        try:
                            os.kill(self.pids[peer], 0)           # signal 0: does the process still exist?
                        except ProcessLookupError:
                            raise RuntimeError('rank %d (pid %d) is gone while this rank waits for %s' % (peer, self.pids[peer], what))
                        except PermissionError:
                            pass
                    if now - t0 > self.timeout_s:
                        raise RuntimeError('timed out after %.0f s waiting for %s' % (self.timeout_s, what))

    def mine(self):
        """(off, borders) views of this rank's slot of the current step, to be filled in place (None, None without /dev/shm).
        Waits until rank 0 has consumed what the slot held two steps ago."""
        if not self.shared:
            return None, None
        if self.step_no >= 2 and self.rank != 0:
            ack = self._hdr(0, 0)
            self._wait(lambda: int(ack[1]) >= self.step_no - 1, 'rank 0 to consume step %d' % (self.step_no - 2), 0)
        return self._views(self.rank, self.step_no & 1)

    def publish(self, off, flat):
        """This rank's CSR of the current step is complete.  -> on rank 0: [(off, flat)] of every rank; else None."""
        k = self.step_no & 1
        self.step_no += 1
        if self.shared:
            self._hdr(self.rank, k)[0] = self.step_no              # the stamp goes last (x86 keeps the stores in order)
            if self.rank != 0:
                return None
            for r in range(1, self.world):
                h = self._hdr(r, k)
                self._wait(lambda: int(h[0]) >= self.step_no, 'rank %d to deliver step %d' % (r, self.step_no - 1), r)
            return [self._views(r, k) for r in range(self.world)]
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object((off, flat), out, dst=0)
        return out

    def consumed(self):
        """rank 0: the lists of the step just published have been stitched; their slots may be reused."""
        if self.shared and self.rank == 0:
            self._hdr(0, 0)[1] = self.step_no

    def close(self):
        if self.shared and self.rank == 0:
            self._hdr(0, 0)[1] = 1 << 60                          # nobody waits for a rank 0 that has left
        self.maps = {}
        self._unlink()


class ShardedRun:
    """The N-process form of wgbsseg_segment_regions: every rank computes the items of the native stitcher's FIRST batch that it
    owns (chunks and the junction patches planned up front: a patch DP is a pure function of its site range), rank 0 collects
    them and runs the one tree (wgbsseg_stitch_regions) over all chunks; only the few patches the rehearsal still misses are
    computed afterwards, by rank 0's `patch_csr`.  No collective on the data path; the borders do not depend on `world`."""

    def __init__(self, dist, regions, chunk_size, loci, params, rank, world, speculate=True, rank0_weight=None):
        import os
        import numpy as np
        from . import _lib
        self.dist, self.rank, self.world = dist, rank, world
        self.regions, self.chunk_size, self.speculate = list(regions), int(chunk_size), speculate
        self.starts, self.ends, self.n_chunks = _lib.first_batch_items(self.regions, chunk_size, speculate)
        # rank 0 also runs the tree (and the follow-up patches) of every step while the other ranks are already computing the
        # next one: it takes a smaller share of the chunks (WGBSSEG_RANK0_WEIGHT; 1 = equal shares)
        if rank0_weight is None:
            rank0_weight = float(os.environ.get('WGBSSEG_RANK0_WEIGHT', '0.75' if world >= 4 else '0.9' if world > 1 else '1'))
        self.shares = plan_weighted(self.regions, chunk_size, world, loci, params, [rank0_weight] + [1.0] * (world - 1))
        self.owner = item_owners(self.starts, self.n_chunks, self.shares)
        self.idx = [np.flatnonzero(self.owner == r) for r in range(world)]
        lens = self.ends - self.starts
        caps = [int(lens[i].sum()) + int(i.size) for i in self.idx]
        self.slots = NodeSlots(dist, rank, world, [i.size for i in self.idx], caps)
        self.my_starts, self.my_ends = self.starts[self.idx[rank]], self.ends[self.idx[rank]]
        self.last_stats = None

    def window(self):
        """0-based resident window [lo, hi) this rank needs (own chunks + halo); (0, 0) when it owns nothing."""
        return int(self.shares['win_lo'][self.rank]), int(self.shares['win_hi'][self.rank])

    def step(self, compute_csr, patch_csr, copy=True):
        """compute_csr(starts, ends, off, out) -> (off, flat): relative border CSR of this rank's items (into the given slot views
        when they are not None); patch_csr(starts, ends) -> (off, flat) for follow-up patches anywhere (rank 0 only).
        -> merged absolute border list per region on rank 0, None elsewhere."""
        import numpy as np
        from . import _lib
        off_v, out_v = self.slots.mine()
        if self.my_starts.size:
            off, flat = compute_csr(self.my_starts, self.my_ends, off_v, out_v)
        else:
            off, flat = (off_v, out_v) if off_v is not None else (np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int32))
            off[0] = 0
        parts = self.slots.publish(off, flat)
        if self.rank != 0:
            return None
        n = self.starts.size
        keep, calls = [parts], [0]

        def batch(st, en):
            calls[0] += 1
            if calls[0] == 1:
                if st.size != n or not (np.array_equal(st, self.starts) and np.array_equal(en, self.ends)):
                    raise RuntimeError('the stitcher asked for a different first batch than wgbsseg_first_batch_items listed')
                ptr, cnt = np.empty(n, dtype=np.uint64), np.empty(n, dtype=np.int64)
                for r, (o, f) in enumerate(parts):
                    i = self.idx[r]
                    if i.size:
                        o = np.asarray(o[:i.size + 1])
                        ptr[i] = np.uint64(f.ctypes.data) + (4 * o[:-1]).astype(np.uint64)
                        cnt[i] = np.diff(o)
                return ptr, cnt
            o, f = patch_csr(st, en)
            keep.append((o, f))
            o = np.asarray(o[:st.size + 1])
            return np.uint64(f.ctypes.data) + (4 * o[:-1]).astype(np.uint64), np.diff(o)
        merged, self.last_stats = _lib.stitch_regions_csr(self.regions, self.chunk_size, batch, speculate=self.speculate, copy=copy)
        self.slots.consumed()
        return merged

    def close(self):
        self.slots.close()


def csr_engine(engine, params):
    """compute_csr / patch_csr of ShardedRun.step over a chunk engine: its own `segment_csr` when it has one (HipEngine,
    GatherEngine: straight into the slot), else built from `segment_many` (the CPU engines of the test-suite)."""
    import numpy as np

    def run(starts, ends, off=None, out=None):
        if hasattr(engine, 'segment_csr'):
            return engine.segment_csr(starts, ends, params, off=off, out=out)
        res = engine.segment_many(list(zip(np.asarray(starts).tolist(), np.asarray(ends).tolist())), params)
        n = len(res)
        if off is None:
            off = np.empty(n + 1, dtype=np.int64)
        off[0] = 0
        np.cumsum([len(r) for r in res], out=off[1:n + 1])
        if out is None:
            out = np.empty(max(1, int(off[n])), dtype=np.int32)
        for i, r in enumerate(res):
            out[off[i]:off[i + 1]] = np.asarray(r, dtype=np.int64) - int(starts[i])
        return off, out
    return run
