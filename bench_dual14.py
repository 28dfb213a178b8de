"""bench.py --workload dual14 — BASELINE.json configs[3]: 14-DoF dual-Panda min-time OCP, N = 25 nodes (cubic x 8 segments),
4,096-problem batch, 20 SQP iterations, <= 700 ADMM iterations.  Two arm workgroups per OCP (k_qp3<8,2>): the arms couple only
through the final time, exchanged once per ADMM iteration.  The dual-arm robot is synthetic (two Pandas on one base facing each
other, 1 m apart: the reference ships one arm); the warm start is the per-arm jerk-limited (Ruckig-equivalent) trajectory
stretched to the slower arm's duration, computed on the GPU inside the timed step."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


def dual_states(B, margins, offset=0):
    """[B][28] = [q_A; q_B; qd_A; qd_B]: arm A from stream `offset + i`, arm B from stream `offset + i + 10^6` of the seeded sampler"""
    from mpc_motion_planner_amd import scenarios
    a0, af = scenarios.make_batch(B, margins, stream_offset=offset)
    b0, bf = scenarios.make_batch(B, margins, stream_offset=offset + 1000000)
    x0 = np.concatenate([a0[:, :7], b0[:, :7], a0[:, 7:], b0[:, 7:]], axis=1)
    xf = np.concatenate([af[:, :7], bf[:, :7], af[:, 7:], bf[:, 7:]], axis=1)
    return np.ascontiguousarray(x0), np.ascontiguousarray(xf)


def run(args, rank, world, local, dist, steps=None, warmup=None, batch=None):
    """rank 0 returns the JSON line (a dict); steps / warmup / batch override the command line (bench.py's brief secondary run)"""
    import torch
    import bench as Bn
    import mpc_motion_planner_amd as M
    from mpc_motion_planner_amd import scenarios, sharding

    nseg, sqp, narm, bytes_per_traj, metric = Bn.WORKLOADS["dual14"]
    margins = Bn.MARGINS
    batch = batch or args.batch or 4096
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    N = 3 * nseg + 1
    total = sharding.global_total(args.scaling, batch, world)
    lo, hi = sharding.shard_bounds(rank, world, total)
    B = hi - lo
    dev = torch.device("cuda", local)
    cfg = M.default_config(nseg, sqp, margins=margins)
    solver = M.Solver(cfg, max(B, 1), device=local, models=M.arm_models(M.DUAL_BASES))
    x0_h, xf_h = dual_states(B, margins, offset=lo)
    x0 = torch.from_numpy(x0_h).to(dev); xf = torch.from_numpy(xf_h).to(dev)
    sol_x = torch.zeros(B, N, 28, dtype=torch.float64, device=dev); sol_u = torch.zeros(B, N, 14, dtype=torch.float64, device=dev)
    sol_T = torch.zeros(B, dtype=torch.float64, device=dev)
    info = torch.zeros(B, 64, dtype=torch.uint8, device=dev)
    warm_x = torch.zeros(B, N, 28, dtype=torch.float64, device=dev); warm_u = torch.zeros(B, N, 14, dtype=torch.float64, device=dev)
    warm_T = torch.zeros(B, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)
    jmax = margins[4] * M.default_limits()["jmax"]
    cap = max(sharding.shard_bounds(r, world, total)[1] - sharding.shard_bounds(r, world, total)[0] for r in range(world))
    rec = torch.zeros(cap, 42 * N + 1, dtype=torch.float64, device=dev)
    gathered = [torch.zeros_like(rec) for _ in range(world)] if (world > 1 and rank == 0) else None

    def step():
        solver.warm_start_jerk_device(B, x0.data_ptr(), xf.data_ptr(), jmax, warm_x.data_ptr(), warm_u.data_ptr(), warm_T.data_ptr(),
                                      stream=stream.cuda_stream)
        solver.solve_device(B, x0.data_ptr(), xf.data_ptr(), sol_x.data_ptr(), sol_u.data_ptr(), sol_T.data_ptr(), info.data_ptr(),
                            warm=(warm_x.data_ptr(), warm_u.data_ptr(), warm_T.data_ptr()), stream=stream.cuda_stream)
        if world > 1:
            torch.cat([sol_x.reshape(B, -1), sol_u.reshape(B, -1), sol_T[:, None]], dim=1, out=rec[:B])
            dist.gather(rec, gathered, dst=0)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(warmup):
        step()
    sync()
    solver.kernel_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kname, k_ms, k_launches = solver.kernel_timing(reset=True)
    if rank != 0:
        return None
    inf = np.frombuffer(info.cpu().numpy().tobytes(), dtype=M.INFO_DTYPE)
    value = total * steps / elapsed
    admm_mean = float(inf["qp_iters_total"].mean())
    flops_traj = Bn.canonical_flops(N, sqp, admm_mean, narm)
    flops_qp = flops_traj - sqp * N * 2.0e4 * narm
    per_gpu = value / world
    achieved = per_gpu * flops_qp / 1e12
    parts = max(1, round(k_launches / float(steps * sqp)))
    k_avg_s = (k_ms / max(k_launches, 1)) * 1e-3
    feasible = (inf["status"] & (1 | 2 | 4 | 16 | 32)) == 0
    peak_meas = Bn.measured_fp64_peak()
    cc = Bn.committed_counters("dual14")
    if cc is not None:
        executed = None if cc["stale"] else per_gpu * admm_mean * cc["flops_per_admm_iter"] / 1e12      # (stale counter profile: no fraction)
    else:
        executed = per_gpu * 2.0 * narm * (admm_mean * Bn.EXECUTED_FMA[nseg][0] + sqp * Bn.EXECUTED_FMA[nseg][1]) / 1e12
    traffic, _, traffic_src = Bn.committed_traffic(kname, B / parts, "dual14")
    out = {
        "metric": metric, "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%d-problem random batch %s, synthetic 14-DoF dual Panda (two arms on one base), N=%d Chebyshev nodes (cubic x %d "
                               "segments), %d SQP iters, <=700 ADMM iters, per-arm jerk-limited warm start merged to the slower arm's duration in "
                               "the timed step (BASELINE.json configs[3])" % (batch, "per GPU" if args.scaling == "weak" else "in the whole job", N, nseg, sqp),
                   "batch": batch, "problems_total": total, "problems_rank0": B, "rccl_world_size": world, "seed": scenarios.SEED,
                   "margins": list(margins), "n_variables": 42 * N + 1},
        "roofline": {"bound": "fp64_valu", "kernel": kname, "achieved": achieved, "peak": Bn.FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": (executed / Bn.FP64_PEAK_TFLOPS) if executed is not None else None,
                     "frac_def": "FP64 flops the QP kernels executed (instruction counters x 64 lanes, inactive lanes counted: an upper bound) per GPU over the wall clock of the timed region / peak; null when the committed counter profile was measured on other kernel sources (executed_source.stale)",
                     "executed_source": cc if cc is not None else "hand count (bench.py EXECUTED_FMA): no counter profile committed",
                     "canonical_frac": achieved / Bn.FP64_PEAK_TFLOPS,
                     "peak_measured": peak_meas,
                     "frac_of_measured_peak": (achieved / peak_meas) if peak_meas else None, "traffic": traffic, "traffic_source": traffic_src, "mfma_busy": 0.0,
                     "mfma_factor_kernel": Bn.committed_mfma("dual14"),
                     "avg_launch_ms": 1e3 * k_avg_s, "launches": k_launches, "launches_in_flight": parts,
                     "problems_per_launch": B / parts, "workgroups_per_problem": 2,
                     "canonical_gflop_per_traj": flops_traj / 1e9, "admm_iters_per_traj": admm_mean,
                     "hbm": {"algorithmic_bytes_per_traj": bytes_per_traj, "frac": per_gpu * bytes_per_traj / 1e9 / Bn.HBM_PEAK_GBS},
                     "note": "canonical dense-equivalent FP64 flops of the whole 1051-variable QP (SURVEY.md 8d) per GPU over the wall clock; the "
                             "kernel exploits that the KKT matrix is two arm blocks bordered by T, so its executed flops are about a quarter of "
                             "the dense-equivalent count"},
        "quality": {**Bn.status_fractions(inf["status"]), "feasible_frac": float(feasible.mean()), "feasible_traj_per_s": value * float(feasible.mean()), "T_mean": float(inf["T"].mean()),
                    "defect_inf_median": float(np.median(inf["defect_inf"])), "term_err_inf_median": float(np.median(inf["term_err_inf"])),
                    "path_viol_inf_max": float(inf["path_viol_inf"].max())},
    }
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_py as o
        ocfg = o.default_config(nseg, sqp, margins=margins)
        models = o.arm_models(o.DUAL_BASES)
        lim = o.default_limits()
        vmax, amax, jm = margins[1] * lim["vmax"], margins[2] * lim["amax"], margins[4] * lim["jmax"]
        nproc = os.cpu_count() or 1
        quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = max(1, int(round(float(q) / float(per))))
        except Exception:
            quota = None
        threads = max(1, min(nproc, quota if quota else 32))
        n_multi, n_single = min(B, 8 * threads), min(B, 12)
        wx = np.zeros((n_multi, N, 28)); wu = np.zeros((n_multi, N, 14)); wT = np.zeros(n_multi)
        for b in range(n_multi):
            wx[b], wu[b], wT[b] = o.warm_start_jerk_multi(nseg, vmax, amax, jm, x0_h[b], xf_h[b])
        t1 = time.perf_counter()
        _, _, T_cpu, _ = o.solve_batch_multi(models, ocfg, x0_h[:n_multi], xf_h[:n_multi], wx, wu, wT, threads=threads)
        dt_multi = time.perf_counter() - t1
        t1 = time.perf_counter()
        o.solve_batch_multi(models, ocfg, x0_h[:n_single], xf_h[:n_single], wx[:n_single], wu[:n_single], wT[:n_single], threads=1)
        dt_single = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": n_multi / dt_multi, "unit": "trajectories/s", "cores": threads, "kind": "port",
                               "single_thread": n_single / dt_single, "nproc": nproc, "cgroup_cpu_quota": quota,
                               "sample": "oracle/liboracle.so multi-arm form: %d problems on %d pthreads in %.1f s; %d problems on 1 thread in %.1f s"
                                         % (n_multi, threads, dt_multi, n_single, dt_single)}
        out["quality"]["max_rel_dT_vs_cpu_sample"] = float(np.max(np.abs(inf["T"][:n_multi] - T_cpu) / np.abs(T_cpu)))
    return out
