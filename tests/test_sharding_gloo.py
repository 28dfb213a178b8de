"""Multi-process (world_size 2, gloo, CPU) check of the sharding + gather code bench.py runs on N GPUs
(mpc_motion_planner_amd/sharding.py: shard_bounds, ShardedBatch.pack_and_gather / assemble): rank r owns a contiguous slice of
the global seeded batch, solves it independently (here the CPU oracle stands in for the device solve, as the checker), and one
gather to rank 0 reassembles the global result in order.  Both scaling modes: weak (B problems per rank) and strong (B problems
in the whole job, uneven split)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
MARGINS = (0.9, 0.9, 0.5, 0.9, 0.1)
N = 13


def _oracle_solve(x0, xf):
    import oracle_py as o
    cfg = o.default_config(4, 1, margins=MARGINS, qp_iters=50)
    B = x0.shape[0]
    sx = np.zeros((B, N, 14)); su = np.zeros((B, N, 7)); sT = np.zeros(B)
    for b in range(B):
        xg, ug, Tg = o.warm_start(cfg, x0[b], xf[b])
        sx[b], su[b], sT[b], _ = o.solve(cfg, x0[b], xf[b], xg, ug, Tg)
    return sx, su, sT


def _worker(rank, world, port, scaling, batch, q):
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mpc_motion_planner_amd import scenarios, sharding
    total = sharding.global_total(scaling, batch, world)
    sb = sharding.ShardedBatch(total, rank, world, N, torch.device("cpu"), dist)
    x0, xf = scenarios.make_batch(sb.count, MARGINS, stream_offset=sb.lo)       # this rank's slice of the global batch
    sx, su, sT = _oracle_solve(x0, xf)
    dist.barrier()
    sb.pack_and_gather(torch.from_numpy(sx), torch.from_numpy(su), torch.from_numpy(sT))
    if rank == 0:
        q.put(sb.assemble().numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def _run(scaling, batch, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (7 if scaling == "strong" else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, scaling, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def _reference(total):
    sys.path.insert(0, HERE)
    from mpc_motion_planner_amd import scenarios
    x0, xf = scenarios.make_batch(total, MARGINS)
    sx, su, sT = _oracle_solve(x0, xf)
    return np.concatenate([sx.reshape(total, -1), su.reshape(total, -1), sT[:, None]], axis=1)


def test_shard_bounds_cover_the_batch_exactly():
    from mpc_motion_planner_amd import sharding
    for total in (1, 5, 1024, 65536, 1000):
        for world in (1, 2, 3, 4, 8):
            cuts = [sharding.shard_bounds(r, world, total) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_bounds(3, 8, 1024) == (384, 512)           # [r*B/G, (r+1)*B/G)
    assert sharding.global_total("weak", 1024, 8) == 8192 and sharding.global_total("strong", 1024, 8) == 1024


def test_two_rank_weak_scaling_matches_single_process():
    got = _run("weak", 3)                       # 3 problems per rank -> 6 in the job
    assert np.array_equal(got, _reference(6))


def test_two_rank_strong_scaling_uneven_split_matches_single_process():
    got = _run("strong", 5)                     # 5 problems in the job -> shards of 2 and 3 (padded gather)
    assert got.shape == (5, 21 * N + 1)
    assert np.array_equal(got, _reference(5))
