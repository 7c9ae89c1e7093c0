"""Multi-process sharding of the segment path: one process per GPU (python -m torch.distributed.run), no collective on
the data path.

Chunks are independent DP problems and chromosomes never interact (segment.py:84-86,129-134).  The chunk grid is cut into
`world` contiguous runs of chunks by the planner of the share groups (wgbsseg_plan_shares, include/wgbsseg.h: balanced by the
scored blocks the chunks hold, always on the reference's grid so every chunk is the very chunk a one-GPU run would segment).
Each rank runs its chunk DPs on its own GPU; the per-chunk border lists are gathered as Python objects on the host of rank
0, which walks the reference's pairwise tree (segment.py:157-165,199-252) over ALL chunks with the native stitcher
(wgbsseg_stitch_regions): the answer does not depend on the number of ranks.
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
