#!/usr/bin/env python3
"""bench.py — trajectories/s of the batched min-time OCP solver on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (init -> 20 x [QP, line search + re-linearisation]) over one batch of
1024 synthetic (start,target) pairs per GPU (BASELINE.json configs[1]: 7-DoF Panda, N=13 nodes, 20 SQP
iterations, <=700 ADMM iterations).  Inputs are resident in HBM before the timed region; weak scaling: every
rank solves its own 1024-problem shard (no data-path collective), then one RCCL gather of the solutions to
rank 0 inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MARGINS = (0.9, 0.9, 0.5, 0.9, 0.1)      # examples/offline_trajectory.cpp:9
NUM_SEG, SQP_ITERS = 4, 20               # BASELINE.json configs[1]
BYTES_PER_TRAJ = 4640                    # compulsory HBM I/O per trajectory at N=13
FP64_PEAK_TFLOPS = 78.6                  # MI355X FP64 vector peak (datasheet), SURVEY.md §8(d)
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: 8 TB/s spec


def canonical_flops(N, sqp_iters, admm_iters_total):
    """SURVEY.md §8(d) dense-equivalent FP64 flop count of one trajectory."""
    n, m = 21 * N + 1, 14 * (N - 1) + 8 * N
    f_rb, f_fact = 2.0e4, n ** 3 / 3.0
    f_iter = 2.0 * n * n + 4.0 * 245 * N + 12.0 * (n + m)
    return sqp_iters * (N * f_rb + f_fact) + admm_iters_total * f_iter


def cpu_baseline(x0, xf, sample, threads, warm):
    """Time the CPU oracle (same algorithm) on a bounded sample of the same workload. Checker/baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as o
    cfg = o.default_config(NUM_SEG, SQP_ITERS, margins=MARGINS)
    N = 3 * NUM_SEG + 1
    wx = np.zeros((sample, N, 14)); wu = np.zeros((sample, N, 7)); wT = np.zeros(sample)
    lim = o.default_limits()
    for b in range(sample):    # (warm start outside the timed part: microseconds per problem, single thread)
        if warm == "jerk":
            wx[b], wu[b], wT[b] = o.warm_start_jerk(NUM_SEG, MARGINS[1] * lim["vmax"], MARGINS[2] * lim["amax"], MARGINS[4] * lim["jmax"], x0[b], xf[b])
        else:
            wx[b], wu[b], wT[b] = o.warm_start(cfg, x0[b], xf[b])
    t0 = time.perf_counter()
    _, _, T, info = o.solve_batch(cfg, x0[:sample], xf[:sample], wx, wu, wT, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": sample / dt, "unit": "trajectories/s", "cores": threads, "kind": "port",
            "sample": "%d of the batch's %d problems, oracle/liboracle.so (C, -O3), %d pthreads, %.1f s wall"
                      % (sample, x0.shape[0], threads, dt)}, T


def bench_receding_horizon(args, M, scenarios, local):
    """BASELINE.json configs[4]: 512 parallel Panda instances x 200 warm-started re-solves, hipGraph-captured step.
    (reference-as-shipped solver depth: 2 SQP iterations per re-solve, motionPlanner.cpp:15; N = 13; dt = 10 ms)"""
    B, resolves, dt = 512, 200, 0.01
    cfg = M.default_config(NUM_SEG, 2, margins=MARGINS)
    s = M.Solver(cfg, B, device=local)
    x0, xf = scenarios.make_batch(B, MARGINS)
    out = {}
    for graph in (False, True):
        s.rh_init(x0, xf)
        s.rh_run(2, dt, use_graph=graph)              # first (cold) solve + graph instantiation are warm-up
        t0 = time.perf_counter()
        s.rh_run(resolves, dt, use_graph=graph)
        el = time.perf_counter() - t0
        out["graph" if graph else "eager"] = B * resolves / el
    xg, sx, su, sT, info = s.rh_get()
    print(json.dumps({"metric": "re-solves/sec, receding-horizon MPC, 512 instances x 200 warm-started re-solves", "value": out["graph"],
                      "unit": "re-solves/s", "n_gpus": 1, "eager_value": out["eager"], "dtype": "f64", "data": "synthetic",
                      "config": {"workload": "512 Panda instances x 200 re-solves, N=13, 2 SQP iters/re-solve, dt=10 ms, hipGraph replay"},
                      "quality": {"status_ok_frac": float((info["status"] == 0).mean()), "T_mean_remaining": float(sT.mean())}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="problems per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=192)
    ap.add_argument("--warm", choices=["jerk", "quintic"], default="jerk",
                    help="initial guess of every OCP: jerk = the jerk-limited time-synchronised trajectory the reference gets from Ruckig "
                         "(solve_trajectory(true), motionPlanner.cpp:146-175), computed on the GPU inside the timed step; quintic = k_init's fallback")
    ap.add_argument("--workload", choices=["batch", "rh", "shipped"], default="batch",
                    help="batch: BASELINE configs[1] (default, the contract line); rh: configs[4] receding horizon, extra line; "
                         "shipped: the reference-as-shipped solver depth (N=19, 2 SQP iterations; SURVEY.md 8d), extra line")
    args = ap.parse_args()
    global NUM_SEG, SQP_ITERS, BYTES_PER_TRAJ
    if args.workload == "shipped":
        NUM_SEG, SQP_ITERS = 6, 2            # robot_ocp.hpp:32 NUM_SEG as shipped (N=19), motionPlanner.cpp:15 max_iter 2
        BYTES_PER_TRAJ = 6656                # compulsory I/O per trajectory at N=19

    import torch
    import mpc_motion_planner_amd as M
    from mpc_motion_planner_amd import scenarios

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 "
                     "bench.py --gpus %d ..." % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the mpcmp product path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    if args.workload == "rh":
        return bench_receding_horizon(args, M, scenarios, local)
    B, N = args.batch, 3 * NUM_SEG + 1
    cfg = M.default_config(NUM_SEG, SQP_ITERS, margins=MARGINS)
    solver = M.Solver(cfg, B, device=local)
    # this rank's shard of the global seeded batch: problems [rank*B, (rank+1)*B)
    x0_h, xf_h = scenarios.make_batch(B, MARGINS, stream_offset=rank * B)
    dev = torch.device("cuda", local)
    x0 = torch.from_numpy(x0_h).to(dev); xf = torch.from_numpy(xf_h).to(dev)
    sol = torch.zeros(B, 21 * N + 1, dtype=torch.float64, device=dev)      # [xs | us | T] per problem
    sol_x = torch.zeros(B, N, 14, dtype=torch.float64, device=dev)
    sol_u = torch.zeros(B, N, 7, dtype=torch.float64, device=dev)
    sol_T = torch.zeros(B, dtype=torch.float64, device=dev)
    info = torch.zeros(B, 64, dtype=torch.uint8, device=dev)               # mpcmp_info records (64 B each)
    gathered = [torch.zeros_like(sol) for _ in range(world)] if (world > 1 and rank == 0) else None
    stream = torch.cuda.current_stream(dev)
    jmax = MARGINS[4] * M.default_limits()["jmax"]                          # motionPlanner.cpp:86-88
    warm_x = torch.zeros(B, N, 14, dtype=torch.float64, device=dev); warm_u = torch.zeros(B, N, 7, dtype=torch.float64, device=dev)
    warm_T = torch.zeros(B, dtype=torch.float64, device=dev)

    def step():
        warm = (0, 0, 0)
        if args.warm == "jerk":
            solver.warm_start_jerk_device(B, x0.data_ptr(), xf.data_ptr(), jmax, warm_x.data_ptr(), warm_u.data_ptr(), warm_T.data_ptr(),
                                          stream=stream.cuda_stream)
            warm = (warm_x.data_ptr(), warm_u.data_ptr(), warm_T.data_ptr())
        solver.solve_device(B, x0.data_ptr(), xf.data_ptr(), sol_x.data_ptr(), sol_u.data_ptr(), sol_T.data_ptr(),
                            info.data_ptr(), warm=warm, stream=stream.cuda_stream)
        if world > 1:
            torch.cat([sol_x.reshape(B, -1), sol_u.reshape(B, -1), sol_T[:, None]], dim=1, out=sol)
            dist.gather(sol, gathered, dst=0)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    solver.kernel_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kname, k_ms, k_launches = solver.kernel_timing(reset=True)

    if rank == 0:
        inf = np.frombuffer(info.cpu().numpy().tobytes(), dtype=M.INFO_DTYPE)
        total = world * B * args.steps
        value = total / elapsed
        admm_mean = float(inf["qp_iters_total"].mean())
        flops_traj = canonical_flops(N, SQP_ITERS, admm_mean)
        k_avg_s = (k_ms / max(k_launches, 1)) * 1e-3
        # a large batch is solved as two half-batches on two streams (mpcmp.hip: solve_impl), so a launch of the dominant kernel
        # covers B / launches_per_sqp problems; its duration is measured with HIP events on the stream it was launched on
        launches_per_sqp = max(1, round(k_launches / float(args.steps * SQP_ITERS)))
        problems_per_launch = B / launches_per_sqp
        alg_bytes_launch = problems_per_launch * BYTES_PER_TRAJ / SQP_ITERS
        achieved_gbs = alg_bytes_launch / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
        traffic = None
        try:    # HBM bytes per k_qp2 launch from the committed rocprofv3 PMC passes (see profiles/r01_traffic.json)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if tj.get("kernel") == kname and tj.get("problems_per_launch") == problems_per_launch:
                traffic = tj["traffic_bytes_per_launch"]
        except Exception:
            traffic = None
        out = {
            "metric": "trajectories/sec, 7-DoF Panda min-time OCP, 1k batch @ 1/2/4/8 GPU" if args.workload == "batch"
                      else "trajectories/sec, 7-DoF Panda min-time OCP, reference-as-shipped depth (N=19, 2 SQP), 1k batch",
            "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d-problem random batch per GPU, 7-DoF Panda, N=%d Chebyshev nodes (cubic x %d segments), "
                                   "%d SQP iters, <=700 ADMM iters, %s warm start computed in the timed step (%s)"
                                   % (B, N, NUM_SEG, SQP_ITERS, "jerk-limited (Ruckig-equivalent)" if args.warm == "jerk" else "quintic",
                                      "BASELINE.json configs[1]" if args.workload == "batch"
                                      else "reference as shipped: robot_ocp.hpp:32, motionPlanner.cpp:15"),
                       "batch_per_gpu": B, "seed": scenarios.SEED, "margins": list(MARGINS)},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": 1e3 * k_avg_s, "launches": k_launches, "problems_per_launch": problems_per_launch,
                         "algorithmic_bytes_per_launch": alg_bytes_launch,
                         "note": "path is FP64-VALU/LDS bound, not HBM bound (SURVEY.md 8d); see fp64"},
            "fp64": {"achieved_tflops": (value / world) * flops_traj / 1e12, "peak_tflops": FP64_PEAK_TFLOPS,
                     "frac": (value / world) * flops_traj / 1e12 / FP64_PEAK_TFLOPS,
                     "canonical_gflop_per_traj": flops_traj / 1e9, "admm_iters_per_traj": admm_mean,
                     "note": "whole solve on the wall clock, per GPU (canonical dense-equivalent flops, SURVEY.md 8d)"},
            "quality": {"status_ok_frac": float((inf["status"] == 0).mean()), "T_mean": float(inf["T"].mean()),
                        "defect_inf_median": float(np.median(inf["defect_inf"])),
                        "term_err_inf_median": float(np.median(inf["term_err_inf"])),
                        "path_viol_inf_max": float(inf["path_viol_inf"].max())},
        }
        if world == 1:
            # PCIe-inclusive rate through the host-buffer entry point (reported, never `value`)
            t1 = time.perf_counter()
            solver.solve(x0_h, xf_h, solver.warm_start_jerk(x0_h, xf_h, jmax) if args.warm == "jerk" else None)
            out["host_buffers_traj_per_s"] = B / (time.perf_counter() - t1)
            if not args.no_cpu_baseline:
                threads = min(os.cpu_count() or 1, 16)
                cb, T_cpu = cpu_baseline(x0_h, xf_h, min(args.cpu_sample, B), threads, args.warm)
                out["cpu_baseline"] = cb
                out["quality"]["max_rel_dT_vs_cpu_sample"] = float(np.max(np.abs(inf["T"][:len(T_cpu)] - T_cpu) / T_cpu))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
