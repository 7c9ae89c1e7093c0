"""Multi-GPU sharding of the segment path: one process per GPU, no collective on the data path.

Chunks are independent DP problems and chromosomes never interact (segment.py:84-86,129-134), so the chunk grid is
cut into `world` contiguous runs of chunks (balanced by site count, always on the reference's grid so every chunk is
the very chunk a single-GPU run would segment).  Each rank segments and stitches its own pieces; the per-rank border
lists are gathered as Python objects on rank 0 (host side), which stitches the at most world-1 junctions that fall
between two ranks with the reference's own rule (segment.py:199-232) and concatenates.
"""
import numpy as np


def chunk_grid(sizes, chunk):
    """[(chrom idx, start, end)] 1-based half-open, the reference's grid (segment.py:124-135) over whole chromosomes."""
    chunks, pos = [], 1
    for ci, sz in enumerate(sizes):
        sz = int(sz)
        for s in range(pos, pos + sz, chunk):
            chunks.append((ci, s, min(s + chunk, pos + sz)))
        pos += sz
    return chunks


def shard_pieces(sizes, chunk, world):
    """-> (pieces per rank: [[(chrom idx, start, end), ...], ...], number of chunks).  A piece is a maximal run of
    consecutive chunks of one chromosome owned by one rank."""
    chunks = chunk_grid(sizes, chunk)
    total = sum(e - s for _, s, e in chunks)
    out, acc, r = [[] for _ in range(world)], 0, 0
    for ci, s, e in chunks:
        while r < world - 1 and acc >= total * (r + 1) / world:
            r += 1
        p = out[r]
        if p and p[-1][2] == s and p[-1][0] == ci:
            p[-1] = (ci, p[-1][1], e)
        else:
            p.append((ci, s, e))
        acc += e - s
    return out, len(chunks)


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
