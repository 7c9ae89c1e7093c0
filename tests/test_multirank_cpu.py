"""The N>1 path on CPU: two gloo ranks take the two shares the planner cuts, each runs its own chunk DPs, rank 0 gathers
the per-chunk border lists and walks the reference's pairwise tree over all of them with the native stitcher; the
result must equal a plain single-process walk of that tree.  The chunk engines here are test infrastructure (the oracle,
or an adversarial engine that forces patch doubling) — the planning, gathering and stitching under test is the product's."""
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
    from wgbs_tools_amd import synth, parallel, segment as S, _lib
    from test_driver_cpu import OracleEngine, FickleEngine, _tree_merge
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    sizes = [50017, 30040, 9000, 7]
    chunk = 8000
    loci = synth.synth_loci(77, sizes)
    total = sum(sizes)
    betas = [synth.synth_betas(77, s, 0, total) for s in range(3)]
    regions = parallel.regions_of_sizes(sizes)
    params = dict(pcount=15.0, max_cpg=1000, max_bp=2000)
    verdicts = []
    # engines: the oracle, and fickle engines whose borders depend on where a DP started (patches must double, sometimes
    # past a whole chunk: the situation in which per-rank trees would differ from the reference's tree)
    engines = [('oracle', OracleEngine(betas, loci))] + [('fickle%%d' %% k, FickleEngine(k, d, a)) for k, (d, a) in enumerate([(3, 170), (7, 400), (20, 80)])]
    for name, eng in engines:
        pr = dict(params, engine=eng)
        mine = parallel.chunks_of_rank(regions, chunk, world, rank, loci, params)
        local = list(zip(mine, eng.segment_many(mine, pr)))
        gathered = parallel.gather_to_rank0(local, rank, world)
        want = None
        if rank == 0:
            assert sorted(k for part in gathered for k, _ in part) == sorted((s, e) for _, s, e in parallel.chunk_grid(regions, chunk))
            assert all(len(part) > 0 for part in gathered)
            cache = {tuple(k): np.asarray(v) for part in gathered for k, v in part}
            asked = []
            def many(sites):
                need = [s for s in sites if s not in cache]
                asked.extend(need)
                got = dict(zip(need, eng.segment_many(need, pr)))
                return [cache[s] if s in cache else got[s] for s in sites]
            merged, _ = _lib.stitch_regions(regions, chunk, many)
            want = []
            for a, b in regions:
                bords = list(range(a, b, chunk)) + [b]
                want.append(_tree_merge(eng.segment_many(list(zip(bords[:-1], bords[1:])), pr), pr))
            ok = all(np.array_equal(m, w) for m, w in zip(merged, want))
            grown = any(b - a > 100 for a, b in asked)
            verdicts.append((name, ok, grown))
        dist.barrier()
        # the product's form (parallel.ShardedRun: every rank computes the items of the stitcher's first batch that it owns,
        # slots in /dev/shm or - second pass - a gather of objects), three steps each so that the two slots alternate
        # third pass: /dev/shm too small on ONE rank (its reservation fails): every rank must fall back to the gather of objects
        for no_shm in ('', '1', 'full'):
            os.environ['WGBSSEG_NO_SHM'] = no_shm if no_shm != 'full' else ''
            reserve = parallel.NodeSlots._reserve
            if no_shm == 'full' and rank == 1:
                def refuse(fd, size):
                    raise OSError(28, 'No space left on device')
                parallel.NodeSlots._reserve = staticmethod(refuse)
            run = parallel.ShardedRun(dist, regions, chunk, loci, params, rank, world)
            parallel.NodeSlots._reserve = staticmethod(reserve)
            assert run.slots.shared == (no_shm == '')
            assert no_shm != 'full' or run.slots.paths == []           # (the slot files of the failed attempt are gone)
            late = []
            def patches(st, en, _e=parallel.csr_engine(eng, pr)):
                late.extend((en - st).tolist())
                return _e(st, en)
            for step in range(3):
                merged = run.step(parallel.csr_engine(eng, pr), patches)
                if rank == 0:
                    ok = all(np.array_equal(m, w) for m, w in zip(merged, want))
                    verdicts.append((name + ' sharded' + (' objects' if no_shm == '1' else ' no room in shm' if no_shm else ' shm') + ' step %%d' %% step, ok, any(x > 100 for x in late)))
                else:
                    assert merged is None
            assert sum(i.size for i in run.idx) == run.starts.size and all(i.size for i in run.idx)
            run.close()
            dist.barrier()
        os.environ.pop('WGBSSEG_NO_SHM', None)
    if rank == 0:
        good = all(ok for _, ok, _ in verdicts) and any(g for n, _, g in verdicts if n.startswith('fickle'))
        print('MULTIRANK_OK' if good else 'MULTIRANK_DIFF', verdicts, flush=True)
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


def test_planned_shares_tile_the_grid_and_balance_work():
    from wgbs_tools_amd import parallel, synth
    sizes = [24900, 100000, 3700, 50000, 1]
    loci = synth.synth_loci(5, sizes, islands=True)
    regions = parallel.regions_of_sizes(sizes)
    params = dict(pcount=15.0, max_cpg=1000, max_bp=2000)
    grid = [(s, e) for _, s, e in parallel.chunk_grid(regions, 1000)]
    for world in (1, 2, 3, 8, 64):
        sh = parallel.plan(regions, 1000, world, loci, params)
        got = []
        for r in range(world):
            mine = parallel.chunks_of_rank(regions, 1000, world, r, loci, params, shares=sh)
            assert len(mine) == sh['chunks'][r]
            got += mine
            pieces = parallel.pieces_of_rank(regions, 1000, world, r, loci, params, shares=sh)
            assert sum(e - s for _, s, e in pieces) == sum(e - s for s, e in mine)
            for ri, s, e in pieces:                                # pieces start on their region's grid, never cross regions
                assert (s - regions[ri][0]) % 1000 == 0 and regions[ri][0] <= s < e <= regions[ri][1]
            if mine:                                               # the resident window covers the share and its halo
                assert sh['win_lo'][r] <= mine[0][0] - 1 and mine[-1][1] - 1 <= sh['win_hi'][r] and sh['win_lo'][r] % 128 == 0
        assert got == grid                                         # contiguous, in order, nothing lost
        assert sh['work'].sum() == parallel.plan(regions, 1000, 2, loci, params)['work'].sum() or world == 1
        if world <= 8:                                             # 179 chunks over <= 8 shares: within one chunk's work of even
            assert sh['work'].max() - sh['work'].min() <= 2 * sh['work'].sum() / len(grid) * 3


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
                              out_path=os.path.join(d, 'sharded.bed'), threads=1, device=0, gpus=1)
    with contextlib.redirect_stderr(io.StringIO()):
        S.SegmentByChunks(args, paths, engine=None).run_sharded(rank, world, local, engine_factory=lambda sr: OracleEngine(betas, loci),
                                                                patch_engine_factory=lambda: OracleEngine(betas, loci))
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


def test_first_batch_items_are_what_the_stitcher_asks_for_first_and_weighted_plans():
    """wgbsseg_first_batch_items == the first request of wgbsseg_stitch_regions (chunks, then the up-front patches), with and without
    speculation; item owners follow the planner's shares; wgbsseg_plan_shares_weighted gives rank 0 its smaller share."""
    from wgbs_tools_amd import parallel, synth, _lib
    sizes = [24900, 100000, 3700, 50000, 1]
    loci = synth.synth_loci(5, sizes)
    regions = parallel.regions_of_sizes(sizes)
    params = dict(pcount=15.0, max_cpg=1000, max_bp=2000)
    for chunk in (1000, 7000, 60000):
        for spec in (True, False):
            st, en, nch = _lib.first_batch_items(regions, chunk, spec)
            grid = parallel.chunk_grid(regions, chunk)
            assert nch == len(grid) and list(zip(st[:nch].tolist(), en[:nch].tolist())) == [(a, b) for _, a, b in grid]
            assert len(set(zip(st.tolist(), en.tolist()))) == st.size and ((en - st)[nch:] <= 200).all()
            seen = []

            class Stop(Exception):
                pass

            def many(sites):
                seen.append(list(sites))
                raise Stop()
            try:
                _lib.stitch_regions(regions, chunk, many, speculate=spec)
            except Stop:
                pass
            assert seen and seen[0] == list(zip(st.tolist(), en.tolist()))
        st, en, nch = _lib.first_batch_items(regions, chunk, True)
        for world in (1, 2, 8):
            w = [0.6] + [1.0] * (world - 1)
            sh = parallel.plan_weighted(regions, chunk, world, loci, params, w)
            eq = parallel.plan(regions, chunk, world, loci, params)
            assert sh['chunks'].sum() == eq['chunks'].sum() == nch
            if world > 1 and nch >= 16 * world:
                assert sh['work'][0] < eq['work'][0] and sh['work'][0] / sh['work'].sum() < 0.9 * eq['work'][0] / eq['work'].sum()
            owner = parallel.item_owners(st, nch, sh)
            for r in range(world):
                mine = parallel.chunks_of_rank(regions, chunk, world, r, loci, params, shares=sh)
                assert [(int(a), int(b)) for a, b in zip(st[:nch][owner[:nch] == r], en[:nch][owner[:nch] == r])] == mine
            lo, hi = sh['win_lo'][owner], sh['win_hi'][owner]                 # every item lies inside its owner's resident window
            assert (st - 1 >= lo).all() and (en - 1 <= hi).all()


WORKER8 = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
    import torch.distributed as dist
    from wgbs_tools_amd import synth, parallel
    from test_driver_cpu import FickleEngine, _tree_merge
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    assert world == 8
    params = dict(pcount=15.0, max_cpg=1000, max_bp=2000)
    verdicts = []
    # (a) 40 chunks over 8 ranks, rank 0 at its default weight of 0.75 of an even share; (b) 5 chunks: ranks that own NOTHING publish empty slots;
    # (c) one region of a single short chunk: seven ranks idle.  Three steps each: the two slots of every rank alternate and a rank may run one
    # step ahead of rank 0's stitching, no further.
    for tag, sizes, chunk in (('40 chunks', [90017, 60040, 30000, 17001, 7], 5000), ('5 chunks', [9000, 2001, 7], 4000), ('1 chunk', [900], 4000)):
        loci = synth.synth_loci(79, sizes)
        regions = parallel.regions_of_sizes(sizes)
        eng = FickleEngine(3, 7, 170)
        pr = dict(params, engine=eng)
        want = None
        if rank == 0:
            want = []
            for a, b in regions:
                bords = list(range(a, b, chunk)) + [b]
                want.append(_tree_merge(eng.segment_many(list(zip(bords[:-1], bords[1:])), pr), pr))
        run = parallel.ShardedRun(dist, regions, chunk, loci, params, rank, world)
        assert run.slots.shared, 'eight ranks of one host hand over through /dev/shm'
        owns = [int(i.size) for i in run.idx]
        assert sum(owns) == run.starts.size
        if tag == '40 chunks':
            w = run.shares['work'].astype(float)
            assert all(o > 0 for o in owns) and w[0] < 0.9 * w[1:].mean(), (owns, w.tolist())      # rank 0 keeps room for the tree
        else:
            assert min(owns) == 0                                                                   # somebody owns nothing
        for step in range(3):
            merged = run.step(parallel.csr_engine(eng, pr), parallel.csr_engine(eng, pr))
            if rank == 0:
                verdicts.append((tag, step, all(np.array_equal(m, w) for m, w in zip(merged, want)), owns))
            else:
                assert merged is None
        run.close()
        dist.barrier()
    if rank == 0:
        print('WORLD8_OK' if all(ok for _, _, ok, _ in verdicts) and len(verdicts) == 9 else 'WORLD8_DIFF', verdicts, flush=True)
    dist.barrier()
    dist.destroy_process_group()
''')


def test_eight_gloo_ranks_equal_single_rank(tmp_path):
    """everything an 8-GPU launch does besides the kernels — WGBSSEG_RANK0_WEIGHT's default of 0.75, eight pairs of
    NodeSlots, the one-step-ahead protocol, ranks that own nothing — before the first real 8-GPU launch does."""
    script = tmp_path / 'worker8.py'
    script.write_text(WORKER8 % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    env.pop('WGBSSEG_RANK0_WEIGHT', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=8', '--master-addr', '127.0.0.1',
           '--master-port', '29533', str(script)]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = res.stdout.decode()
    assert res.returncode == 0, out[-3000:]
    assert 'WORLD8_OK' in out, out[-3000:]
