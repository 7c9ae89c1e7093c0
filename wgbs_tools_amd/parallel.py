"""Multi-GPU sharding of the segment path: one process per GPU, no collective on the data path.

Chunks are independent DP problems and chromosomes never interact (segment.py:84-86,129-134), so the chunk grid is
cut into `world` contiguous runs of chunks (balanced by site count, always on the reference's grid so every chunk is
the very chunk a single-GPU run would segment).  Each rank segments and stitches its own pieces; the per-rank border
lists are gathered as Python objects on rank 0 (host side), which stitches the at most world-1 junctions that fall
between two ranks with the reference's own rule (segment.py:199-232) and concatenates.
"""
import numpy as np


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


def shard_regions(regions, chunk, world):
    """-> (pieces per rank: [[(region idx, start, end), ...], ...], number of chunks).  A piece is a maximal run of
    consecutive chunks of one region owned by one rank; pieces start on the region's chunk grid."""
    chunks = chunk_grid(regions, chunk)
    total = sum(e - s for _, s, e in chunks)
    out, acc, r = [[] for _ in range(world)], 0, 0
    for ri, s, e in chunks:
        while r < world - 1 and acc >= total * (r + 1) / world:
            r += 1
        p = out[r]
        if p and p[-1][2] == s and p[-1][0] == ri:
            p[-1] = (ri, p[-1][1], e)
        else:
            p.append((ri, s, e))
        acc += e - s
    return out, len(chunks)


def shard_pieces(sizes, chunk, world):
    """shard_regions over whole chromosomes of the given CpG counts."""
    return shard_regions(regions_of_sizes(sizes), chunk, world)


def stitch_across_ranks(gathered, stitch_fn):
    """gathered: list over ranks of [(chrom idx, start, end, borders ndarray), ...].  Joins pieces of the same
    chromosome in order with `stitch_fn(b1, b2)` (= stitch_2_dfs bound to an engine).  -> {chrom idx: borders}."""
    by_chrom = {}
    for plist in gathered:
        for ci, s, e, b in plist:
            by_chrom.setdefault(ci, []).append((s, e, np.asarray(b)))
    merged = {}
    for ci, lst in by_chrom.items():
        lst.sort(key=lambda x: x[0])
        cur = lst[0][2]
        for s, e, b in lst[1:]:
            assert cur[-1] == s and b[0] == s, 'pieces of a chromosome must tile it'
            cur = stitch_fn(cur, b)
        merged[ci] = cur
    return merged


def gather_to_rank0(local, rank, world):
    """Host-side gather of per-rank piece results (small int arrays); no-op for world == 1."""
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
    """The only communication of a multi-GPU segment run is the final host-side gather of border lists: a gloo group."""
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group('gloo')
    return dist
