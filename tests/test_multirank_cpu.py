"""The N>1 path on CPU: two gloo ranks shard the chunk grid, each segments + stitches its own pieces, rank 0 gathers
and joins across ranks; the result must equal the single-rank run.  The chunk engine here is the oracle (test
infrastructure) — the sharding, gathering and cross-rank stitching code under test is the product's."""
import os
import os.path as op
import subprocess
import sys
import textwrap

import numpy as np

ROOT = op.dirname(op.dirname(op.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
    import torch.distributed as dist
    from wgbs_tools_amd import synth, parallel, segment as S
    from test_driver_cpu import OracleEngine
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    sizes = [50017, 30040, 9000, 7]
    chunk = 8000
    loci = synth.synth_loci(77, sizes)
    total = sum(sizes)
    betas = [synth.synth_betas(77, s, 0, total) for s in range(3)]
    eng = OracleEngine(betas, loci)
    params = dict(pcount=15.0, max_cpg=1000, max_bp=2000, engine=eng)

    def run_pieces(pieces):
        out = []
        for ci, s, e in pieces:
            bords = list(range(s, e, chunk)) + [e]
            arr = eng.segment_many(list(zip(bords[:-1], bords[1:])), params)
            sbc = S.SegmentByChunks.__new__(S.SegmentByChunks)
            sbc.param_dict = params
            out.append((ci, s, e, sbc.merge_df_list(arr)))
        return out
    pieces, nch = parallel.shard_pieces(sizes, chunk, world)
    assert sum(len(p) for p in pieces) >= len(sizes) and nch == sum(-(-s // chunk) for s in sizes)
    local = run_pieces(pieces[rank])
    dist.barrier()
    gathered = parallel.gather_to_rank0(local, rank, world)
    if rank == 0:
        merged = parallel.stitch_across_ranks(gathered, lambda a, b: S.stitch_2_dfs(a, b, params))
        single = parallel.stitch_across_ranks([run_pieces(parallel.shard_pieces(sizes, chunk, 1)[0][0])], None)
        ok = all(np.array_equal(merged[c], single[c]) for c in single) and set(merged) == set(single)
        print('MULTIRANK_OK' if ok else 'MULTIRANK_DIFF', {c: len(v) for c, v in merged.items()}, flush=True)
    dist.barrier()
    dist.destroy_process_group()
''')


def test_two_gloo_ranks_equal_single_rank(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29531', str(script)]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = res.stdout.decode()
    assert res.returncode == 0, out[-3000:]
    assert 'MULTIRANK_OK' in out, out[-3000:]


def test_shard_pieces_tile_the_grid():
    from wgbs_tools_amd import parallel
    sizes = [249, 1000, 37, 5000, 1]
    for world in (1, 2, 3, 8):
        pieces, nch = parallel.shard_pieces(sizes, 100, world)
        flat = sorted(p for pl in pieces for p in pl)
        # pieces tile every chromosome, start on the chunk grid, and never cross chromosomes
        pos = 1
        for ci, sz in enumerate(sizes):
            mine = [p for p in flat if p[0] == ci]
            assert mine[0][1] == pos and mine[-1][2] == pos + sz
            for a, b in zip(mine, mine[1:]):
                assert a[2] == b[1]
            for _, s, e in mine:
                assert (s - pos) % 100 == 0
            pos += sz
        assert nch == sum(-(-s // 100) for s in sizes)


WORKER2 = textwrap.dedent('''
    import os, sys, argparse, io, contextlib
    import numpy as np
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
    from wgbs_tools_amd import synth, segment as S, parallel
    from test_driver_cpu import OracleEngine
    rank, world, local = parallel.env_rank_world()
    d = %(tmp)r
    names = ['chr1', 'chr2', 'chr3']; sizes = [30017, 21040, 9007]
    loci = synth.synth_loci(78, sizes); total = sum(sizes)
    betas = [synth.synth_betas(78, s, 0, total) for s in range(3)]
    refdir = os.path.join(d, 'references', 'g')
    paths = [os.path.join(d, 's%%d.beta' %% i) for i in range(3)]
    if rank == 0:
        synth.write_genome(refdir, names, sizes, loci)
        for p, b in zip(paths, betas): synth.write_beta(p, b)
    parallel.init_host_group().barrier()
    args = argparse.Namespace(sites=None, region=None, array_id=None, bed_file=None, genome=refdir, betas=paths, beta_file=None,
                              chunk_size=7000, pcount=15, min_cpg=1, max_cpg=1000, max_bp=2000,
                              out_path=os.path.join(d, 'sharded.bed'), threads=1, device=0)
    with contextlib.redirect_stderr(io.StringIO()):
        S.SegmentByChunks(args, paths, engine=None).run_sharded(rank, world, local, engine_factory=lambda sr: OracleEngine(betas, loci))
        if rank == 0:
            args.out_path = os.path.join(d, 'single.bed')
            os.environ['WORLD_SIZE'] = '1'
            S.SegmentByChunks(args, paths, engine=OracleEngine(betas, loci)).run()
    if rank == 0:
        a, b = open(os.path.join(d, 'sharded.bed')).read(), open(os.path.join(d, 'single.bed')).read()
        print('SHARDED_CLI_OK' if (a == b and a.count('\\n') > 1000) else 'SHARDED_CLI_DIFF', a.count('\\n'), b.count('\\n'), flush=True)
''')


def test_sharded_driver_equals_single_rank(tmp_path):
    script = tmp_path / 'worker2.py'
    script.write_text(WORKER2 % dict(root=ROOT, tmp=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29532', str(script)]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = res.stdout.decode()
    assert res.returncode == 0, out[-3000:]
    assert 'SHARDED_CLI_OK' in out, out[-3000:]
