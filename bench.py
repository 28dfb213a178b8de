#!/usr/bin/env python3
"""bench.py — trajectories/s of the batched min-time OCP solver on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (warm start -> init -> K x [QP, line search + re-linearisation]) over one batch of
synthetic (start,target) pairs.  Default workload = BASELINE.json configs[1]: 1024 problems per GPU, 7-DoF Panda, N=13
nodes, 20 SQP iterations, <=700 ADMM iterations.  `value` is measured with inputs and outputs resident in HBM
(device-pointer entry points); the host->host rate of SURVEY.md 8(d) (PCIe-inclusive, median of >=5 warm repeats) is
reported beside it as `host_to_host`, never as `value`.

Multi-GPU: one process per GPU; the global seeded batch is cut into contiguous slices (mpc_motion_planner_amd/sharding.py),
no data-path collective, one RCCL gather of the solutions to rank 0 inside the timed region.
  --scaling weak   (default)  --batch problems PER GPU          (what the driver's N=1,2,4,8 sweep measures)
  --scaling strong            --batch problems in the WHOLE job (slice [r*B/G, (r+1)*B/G))

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--scaling weak|strong] [--workload ...]
    (N>1 without WORLD_SIZE in the environment: bench.py starts its own N ranks — `python -m torch.distributed.run --nnodes=1
     --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N` as a child process, before anything touches the GPU — and
     forwards rank 0's JSON line and the exit code; under torchrun it is a rank.)
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
FP64_PEAK_TFLOPS = 78.6                  # MI355X FP64 vector peak: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (datasheet)
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: 8 TB/s spec

# workload -> (num_seg, sqp_iters, n_arms, compulsory HBM bytes per trajectory, metric string)
WORKLOADS = {
    "batch":   (4, 20, 1, 4640, "trajectories/sec, 7-DoF Panda min-time OCP, 1k batch @ 1/2/4/8 GPU"),
    "shipped": (6, 2, 1, 6656, "trajectories/sec, 7-DoF Panda min-time OCP, reference-as-shipped depth (N=19, 2 SQP), 1k batch"),
    "dual14":  (8, 20, 2, 17296, "trajectories/sec, 14-DoF dual-Panda min-time OCP, N=25, 4096-problem batch"),
}
# FP64 FMAs the dominant kernel EXECUTES (tools/isa_mix.py on the ISA of the ADMM loop / the sweeps), per ADMM iteration and
# per factorisation; None where it has not been counted
# FMAs the kernels execute on useful data: (per ADMM iteration, per factorisation) of one arm.  N = 13: counted in k_qp2's ISA
# (round 1).  N = 19 / 25 (k_qp3f + k_qp3): products of the algorithm — two G products (NSEG x 49^2 each), S^-1 (n_I^2), the sparse
# and dense K_JC parts, A x~ and A^T w (22 per path row, 6 per dynamics row), the updates; factorisation: Gauss-Jordan sweeps of
# the NSEG + 1 interior blocks (49^3 each), the Schur complement (28 columns x (49^2 + 200) per segment), the sweep of S (n_I^3).
EXECUTED_FMA = {4: (34.0e3, 0.85e6), 6: (55.0e3, 2.2e6), 8: (66.0e3, 3.6e6)}


def canonical_flops(N, sqp_iters, admm_iters_total, narm=1):
    """SURVEY.md 8(d) dense-equivalent FP64 flop count of one trajectory (n, m of the whole NLP)."""
    n, m = 21 * N * narm + 1, (14 * (N - 1) + 8 * N) * narm
    f_rb, f_fact = 2.0e4 * narm, n ** 3 / 3.0
    f_iter = 2.0 * n * n + 4.0 * 245 * N * narm + 12.0 * (n + m)
    return sqp_iters * (N * f_rb + f_fact) + admm_iters_total * f_iter


def measured_fp64_peak():
    """FP64 FMA peak measured on this pool by tools/micro/fp64_peak.hip (profiles/r02_fp64_peak.json), None if absent"""
    try:
        return float(json.load(open(os.path.join(ROOT, "profiles", "r02_fp64_peak.json")))["tflops"])
    except Exception:
        return None


def _tree_sha():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from src_hash import csrc_sha16
        return csrc_sha16(ROOT)
    except Exception:
        return None


def _newest_profile(suffix):
    """newest committed profiles/rNN_<suffix> (round number descending), or None"""
    import glob
    import re
    c = [(int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)), f) for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix))]
    return max(c)[1] if c else None


def committed_traffic(kname, problems_per_launch, workload=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes: (bytes, MFMA busy cycles, source record).
    The newest round's file (profiles/rNN_traffic.json, tools/traffic_from_summary.py) is keyed by workload and kernel and names the kernel sources it was
    measured on; the figure is withheld (None) when the tree differs.  Older files (one kernel each, no hash) are used for an unchanged kernel only."""
    try:
        tf = _newest_profile("traffic.json")
        tj = json.load(open(tf))
        rec = tj["workloads"].get(workload or "", {})
        e = rec.get("kernels", {}).get(kname)
        if e is not None and abs((rec.get("problems_per_launch") or -1e9) - problems_per_launch) < 1.0 and "traffic_bytes_per_launch" in e:
            stale = tj.get("csrc_sha16") != _tree_sha()
            src = {"file": "profiles/" + os.path.basename(tf), "csrc_sha16": tj.get("csrc_sha16"), "stale": stale,
                   "what": "FETCH_SIZE x 2 + WRITE_SIZE of the dominant kernel, per launch (committed profile, not this run)"}
            busy = e.get("SQ_VALU_MFMA_BUSY_CYCLES")
            if busy is not None and e.get("avg_us_kernel_trace"):
                # fraction of the kernel's SIMD time (256 CUs x 4 SIMDs, 2.4 GHz nominal) in which a matrix-core instruction was executing
                src["mfma_busy_frac_of_simd_time"] = busy / (e["avg_us_kernel_trace"] * 1e-6 * 2.4e9 * 1024)
                src["mfma_insts_per_launch"] = e.get("SQ_INSTS_MFMA")
            return (None if stale else e["traffic_bytes_per_launch"]), busy, src
    except Exception:
        pass
    for f in ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", f)))
            if tj.get("kernel") == kname and tj.get("problems_per_launch") == problems_per_launch:
                return tj["traffic_bytes_per_launch"], tj.get("mfma_busy_cycles_per_launch"), {"file": "profiles/" + f, "stale": None,
                        "what": "FETCH_SIZE x 2 + WRITE_SIZE of the dominant kernel, per launch (committed profile of an earlier round, no source hash)"}
        except Exception:
            pass
    return None, None, None


def committed_mfma(workload):
    """MFMA counters of the factorisation kernel k_qp3f (its Schur complement products run on the matrix cores) from the committed
    newest committed rocprofv3 PMC pass (profiles/rNN_traffic.json, else r03_mfma.json)"""
    try:
        tf = _newest_profile("traffic.json")
        tj = json.load(open(tf))
        rec = tj["workloads"].get(workload, {})
        for kf in ("k_qp3f", "k_qp2"):      # N >= 19: the factorisation kernel; N = 13: k_qp2 factorises itself (Schur products on the matrix cores since round 5)
            e = rec.get("kernels", {}).get(kf)
            if e is not None and e.get("SQ_INSTS_MFMA"):
                return {"kernel": kf, "insts_mfma_per_launch": e["SQ_INSTS_MFMA"], "mfma_busy_cycles_per_launch": e.get("SQ_VALU_MFMA_BUSY_CYCLES"),
                        "mops_f64_per_launch": e.get("SQ_INSTS_VALU_MFMA_MOPS_F64"), "problems_per_launch": rec.get("problems_per_launch"),
                        "source": "committed_profile: profiles/" + os.path.basename(tf), "stale": tj.get("csrc_sha16") != _tree_sha()}
    except Exception:
        pass
    try:
        mj = json.load(open(os.path.join(ROOT, "profiles", "r03_mfma.json")))
        e = mj.get(workload)
        if e is None:
            return None
        return {"kernel": mj["kernel"], "insts_mfma_per_launch": e["SQ_INSTS_MFMA"], "mfma_busy_cycles_per_launch": e["SQ_VALU_MFMA_BUSY_CYCLES"],
                "busy_frac_of_simd_time": e.get("mfma_busy_frac_of_simd_time"), "source": "committed_profile: profiles/r03_mfma.json"}
    except Exception:
        return None


def _finite(x):
    """the line must be strict JSON: a non-finite number (e.g. the mean final time of a batch with a failed instance) becomes null"""
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if isinstance(x, float) and not np.isfinite(x):
        return None
    if isinstance(x, np.generic):
        return _finite(x.item())
    return x


def status_fractions(st):
    """what the per-problem status words (include/mpcmp.h MPCMP_STATUS_*) say about a batch"""
    st = np.asarray(st)
    return {"status_ok_frac": float((st == 0).mean()),                       # converged QPs AND an iterate inside every tolerance
            "hard_fail_frac": float(((st & 7) != 0).mean()),                 # NaN / lost positive definiteness / dead exchange
            "qp_capped_frac": float(((st & 8) != 0).mean()),                 # at least one QP stopped at qp_iters (truncated ADMM)
            "outside_tol_frac": float(((st & 16) != 0).mean()),              # defect / path violation > eps_abs or terminal error > eps_target + eps_abs
            "T_out_of_box_frac": float(((st & 32) != 0).mean())}


def committed_counters(workload):
    """FP64 VALU instruction counters of the dominant kernels from the newest committed rocprofv3 PMC pass (profiles/r0N_fp64_counters.json, written
    by tools/fp64_counters.py from `tools/profile_round.sh`): flops the ISA executed per ADMM iteration of one trajectory (factorisation included in
    the ratio; wave-level counts x 64 lanes, inactive lanes included: an upper bound).  `stale` is True when the kernel sources of the working tree
    (tools/src_hash.py) are not the ones the profile was measured on: the fraction is then withheld.  None if no profile has the workload."""
    now = _tree_sha()
    import glob
    for f in sorted((os.path.basename(g) for g in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_fp64_counters.json"))), reverse=True):
        try:
            cj = json.load(open(os.path.join(ROOT, "profiles", f)))
            e = cj["workloads"].get(workload)
            if e is None:
                continue
            sha = e.get("csrc_sha16") or cj.get("csrc_sha16")
            return {"flops_per_admm_iter": float(e["fp64_flops_per_traj_admm_iter"]), "source": "profiles/" + f, "csrc_sha16": sha, "tree_sha16": now,
                    "stale": bool(sha is None or now is None or sha != now),
                    "counters": "64 x (2 SQ_INSTS_VALU_FMA_F64 + SQ_INSTS_VALU_ADD_F64 + SQ_INSTS_VALU_MUL_F64) + MFMA MOPS, of every QP kernel; inactive lanes counted (upper bound)"}
        except Exception:
            pass
    return None


def cpu_baseline(nseg, sqp, x0, xf, warm, n_multi, n_single, qp_warm_start=0):
    """Time the CPU oracle (same algorithm, same warm start; oracle/liboracle.so, C -O3) on a bounded sample of the same
    workload: (i) one thread = the reference's execution model (examples/benchmark.cpp:16), (ii) a pthread pool over
    problems on the host cores this process may use.  Checker/baseline only — never the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as o
    cfg = o.default_config(nseg, sqp, margins=MARGINS, qp_warm_start=int(qp_warm_start))
    N = 3 * nseg + 1
    nproc = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = nproc
    # a container may see every host core but be throttled to a CPU quota (cgroup v2 cpu.max): size the pool by the quota
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(round(float(q) / float(per))))
    except Exception:
        quota = None
    threads = max(1, min(usable, quota if quota else 32))
    n_multi = min(n_multi, x0.shape[0]); n_single = min(n_single, x0.shape[0])
    lim = o.default_limits()
    wx = np.zeros((n_multi, N, 14)); wu = np.zeros((n_multi, N, 7)); wT = np.zeros(n_multi)
    for b in range(n_multi):    # (warm start outside the timed part: microseconds per problem)
        if warm == "jerk":
            wx[b], wu[b], wT[b] = o.warm_start_jerk(nseg, MARGINS[1] * lim["vmax"], MARGINS[2] * lim["amax"], MARGINS[4] * lim["jmax"], x0[b], xf[b])
        else:
            wx[b], wu[b], wT[b] = o.warm_start(cfg, x0[b], xf[b])
    t0 = time.perf_counter()
    _, _, T, _ = o.solve_batch(cfg, x0[:n_multi], xf[:n_multi], wx, wu, wT, threads=threads)
    dt_multi = time.perf_counter() - t0
    t0 = time.perf_counter()
    o.solve_batch(cfg, x0[:n_single], xf[:n_single], wx[:n_single], wu[:n_single], wT[:n_single], threads=1)
    dt_single = time.perf_counter() - t0
    return {"value": n_multi / dt_multi, "unit": "trajectories/s", "cores": threads, "kind": "port",
            "single_thread": n_single / dt_single, "nproc": nproc, "usable_cores": usable, "cgroup_cpu_quota": quota,
            "sample": "oracle/liboracle.so (C, -O3): %d problems of the batch on %d pthreads in %.1f s; %d problems on 1 thread "
                      "(the reference's execution model, examples/benchmark.cpp:16) in %.1f s; os.cpu_count() = %d"
                      % (n_multi, threads, dt_multi, n_single, dt_single, nproc)}, T


RH_REC = 18      # doubles per instance in the receding-horizon gather: final state (14) | T | status | ADMM iterations of the last re-solve | spare


def rh_shard(args, rank, world, B):
    """configs[4] across ranks (no collective inside the loop, gather only final statistics): weak = B instances PER rank,
    strong = B instances in the whole job; rank r owns the contiguous slice [lo, hi) of the global seeded instance list"""
    from mpc_motion_planner_amd import sharding
    total = sharding.global_total(args.scaling, B, world)
    lo, hi = sharding.shard_bounds(rank, world, total)
    return total, lo, hi


def rh_collect(rank, world, dist, device, total, x_final, sT, status, qp_iters, done, arrived, elapsed):
    """after the loop: ONE gather of the per-instance final records to rank 0, one sum of the two per-rank counters, the maximum of the ranks'
    loop times.  Returns (records [total][RH_REC] on rank 0 else None, done_total, arrived_total, elapsed_max).  CPU stub and GPU ranks share it."""
    import torch
    from mpc_motion_planner_amd import sharding
    c = x_final.shape[0]
    rows = torch.zeros(c, RH_REC, dtype=torch.float64)
    rows[:, :14] = torch.as_tensor(np.asarray(x_final, dtype=np.float64)); rows[:, 14] = torch.as_tensor(np.asarray(sT, dtype=np.float64))
    rows[:, 15] = torch.as_tensor(np.asarray(status, dtype=np.float64)); rows[:, 16] = torch.as_tensor(np.asarray(qp_iters, dtype=np.float64))
    sb = sharding.ShardedBatch(total, rank, world, 13, device, dist, width=RH_REC)
    assert sb.count == c
    sb.gather_rows(rows.to(device))
    cnt = torch.tensor([float(done), float(arrived), 0.0], dtype=torch.float64, device=device)
    tmax = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    rec = sb.assemble().cpu().numpy() if rank == 0 else None
    return rec, int(cnt[0].item()), int(cnt[1].item()), float(tmax.item())


def bench_receding_horizon(args, M, scenarios, rank, world, local, dist, flags=None, B=None):
    """BASELINE.json configs[4]: 512 parallel Panda instances x 200 warm-started re-solves, hipGraph-captured step, sharded over the ranks.
    (reference-as-shipped solver depth: 2 SQP iterations per re-solve, motionPlanner.cpp:15; N = 13; dt = 10 ms.)  flags: None = the driver's
    defaults (carried multipliers + warm QP duals + arrival, include/mpcmp.h mpcmp_rh_run), "cold" = both start flags off (-1)."""
    import torch
    nseg, sqp = 4, 2
    resolves, dt = 200, 0.01
    total, lo, hi = rh_shard(args, rank, world, B or args.batch or 512)
    Bl = hi - lo
    fl = -1 if flags == "cold" else 0
    cfg = M.default_config(nseg, sqp, margins=MARGINS, qp_warm_start=fl, carry_multipliers=fl)
    s = M.Solver(cfg, max(Bl, 1), device=local)
    x0, xf = scenarios.make_batch(Bl, MARGINS, stream_offset=lo)
    dev = torch.device("cuda", local)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    out, live = {}, {}
    kname = k_ms = k_n = None
    for graph in ((False, True) if world == 1 else (True,)):
        s.rh_init(x0, xf)
        s.rh_run(2, dt, use_graph=graph)              # first (cold) solve + graph instantiation are warm-up
        d0 = s.rh_stats()[0]
        if not graph:
            s.kernel_timing(reset=True)               # HIP events around the QP launches (eager mode only; a graph replay has none)
        sync()
        t0 = time.perf_counter()
        s.rh_run(resolves, dt, use_graph=graph)       # (synchronises its stream before it returns)
        sync()
        el = time.perf_counter() - t0
        if not graph:
            kname, k_ms, k_n = s.kernel_timing(reset=True)
        out["graph" if graph else "eager"] = el
        live["graph" if graph else "eager"] = s.rh_stats()[0] - d0
    xg, sx, su, sT, info = s.rh_get()
    done, arrived = s.rh_stats()
    rec, done_all, arrived_all, el_max = rh_collect(rank, world, dist, dev, total, xg, sT, info["status"], info["qp_iters_total"], live["graph"], arrived, out["graph"])
    if rank != 0:
        return None
    N = 3 * nseg + 1
    st_all = rec[:, 15].astype(np.int64)
    alive = (st_all & 64) == 0
    value = total * resolves / el_max                                  # instance-steps per second, arrived instances included (they cost nothing)
    live_value = done_all / el_max                                     # re-solves actually executed per second
    admm_per_resolve = float(rec[:, 16].mean())                        # of every instance's last executed re-solve
    flops = canonical_flops(N, sqp, admm_per_resolve)
    cc_rh = committed_counters("batch")
    line = {"metric": "re-solves/sec, receding-horizon MPC, 512 instances x 200 warm-started re-solves", "value": value,
            "unit": "re-solves/s", "n_gpus": world, "steps": resolves, "warmup": 2, "ms_per_step": 1e3 * el_max / resolves,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "live_resolves_per_s": live_value, "resolves_executed": done_all, "instances_arrived": arrived_all,
            "eager_value": (Bl * resolves / out["eager"]) if "eager" in out else None,
            "config": {"workload": "%d Panda instances %s x 200 re-solves, N=13, 2 SQP iters/re-solve, dt=10 ms, hipGraph replay of the two-stream step, %s "
                                   "(BASELINE.json configs[4])" % (total if args.scaling == "strong" else Bl, "in the whole job" if args.scaling == "strong" else "per GPU",
                                                                  "start flags off (-1): every re-solve from lambda = 0, cold QPs" if flags == "cold"
                                                                  else "driver defaults: carried multipliers, warm QP duals, arrival handling"),
                       "instances_total": total, "instances_rank0": Bl, "rccl_world_size": world, "flags": flags or "default"},
            "roofline": {"bound": "fp64_valu", "kernel": kname, "achieved": live_value * flops / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": (live_value / world * admm_per_resolve * cc_rh["flops_per_admm_iter"] / 1e12 / FP64_PEAK_TFLOPS) if (cc_rh and not cc_rh["stale"]) else None,
                         "frac_def": "FP64 flops the QP kernel executed on LIVE re-solves (instruction counters of the N = 13 kernel x 64 lanes, inactive lanes counted: an upper bound) per GPU over the wall clock / peak; null when the committed counter profile is stale (executed_source.stale)",
                         "executed_source": cc_rh,
                         "canonical_frac": live_value / world * flops / 1e12 / FP64_PEAK_TFLOPS, "traffic": None, "mfma_busy": 0.0,
                         "avg_launch_ms_eager": (k_ms / max(k_n, 1)) if k_n else None, "launches_eager": k_n,
                         "admm_iters_per_resolve": admm_per_resolve, "canonical_gflop_per_resolve": flops / 1e9,
                         "note": "live re-solves on the wall clock (canonical dense-equivalent flops, SURVEY.md 8d); the kernel uses no MFMA"},
            "quality": dict(status_fractions(st_all & 63), arrived_frac=float((~alive).mean()),
                            T_mean_remaining_live=float(np.nanmean(rec[alive, 14])) if alive.any() else 0.0,
                            T_nan_frac=float(np.isnan(rec[:, 14]).mean()), state_nonfinite_frac=float((~np.isfinite(rec[:, :14]).all(axis=1)).mean()),
                            note="status of every instance's LAST executed re-solve (an arrived instance keeps the record of the solve it arrived with)")}
    if world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle's loop (same rule: re-guess, carried multipliers, advance + arrival) on a bounded sample of the instances
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_py as o
        on = 0 if flags == "cold" else 1
        ocfg = o.default_config(nseg, sqp, margins=MARGINS, qp_warm_start=on, carry_multipliers=on)
        ns, reps, nsolve = 24, 40, 0
        t0 = time.perf_counter()
        for b in range(ns):
            wx, wu, wT = o.rh_start_guess(ocfg, x0[b], xf[b]); xc = x0[b].copy(); lam = None
            for r in range(reps):
                xs, us, T, oi, lam = o.solve_carry(ocfg, xc, xf[b], wx, wu, wT, lam=lam)
                nsolve += 1
                xc, retired = o.rh_advance(ocfg, xs, us, T, oi.status, dt, xf[b], xc)
                if retired:
                    break
                if oi.status & (1 | 2 | 4 | 32):                  # a failed solve is neither a trajectory to follow nor a guess (k_advance, k_init)
                    wx, wu, wT = o.rh_start_guess(ocfg, xc, xf[b])
                else:
                    wx, wu, wT = xs.copy(), us, T
                    wx[0] = xc; wx[-1] = xf[b]
        el = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": nsolve / el, "unit": "re-solves/s", "cores": 1, "kind": "port", "single_thread": nsolve / el,
                                "nproc": os.cpu_count(), "sample": "oracle loop (same rule), %d instances x <= %d re-solves = %d re-solves, 1 thread, %.1f s" % (ns, reps, nsolve, el)}
    return line


def stub_rh_rank(args, rank, world):
    """CPU stand-in of one rank of the receding-horizon workload (tests/test_bench_launcher.py): gloo instead of RCCL, the CPU oracle's loop as the
    stand-in of the device loop on a few instances — the sharding, the single gather of final records, the counter sum and the max-over-ranks time
    are the code of the real bench (rh_shard, rh_collect)."""
    import torch
    import torch.distributed as dist
    from mpc_motion_planner_amd import scenarios
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as o
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    total, lo, hi = rh_shard(args, rank, world, args.batch or 2)
    x0, xf = scenarios.make_batch(hi - lo, MARGINS, stream_offset=lo)
    ocfg = o.default_config(4, 1, margins=MARGINS, qp_iters=50, carry_multipliers=1, qp_warm_start=1)
    dt, c = 0.4, hi - lo
    xfin, sT, st, its, done, arrived = np.zeros((c, 14)), np.zeros(c), np.zeros(c), np.zeros(c), 0, 0
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for b in range(c):
        wx, wu, wT = o.rh_start_guess(ocfg, x0[b], xf[b]); xc = x0[b].copy(); lam = None
        for r in range(args.steps):
            xs, us, T, oi, lam = o.solve_carry(ocfg, xc, xf[b], wx, wu, wT, lam=lam)
            done += 1
            xc, retired = o.rh_advance(ocfg, xs, us, T, oi.status, dt, xf[b], xc)
            sT[b], st[b], its[b] = T, oi.status | (64 if retired else 0), oi.qp_iters_total
            if retired:
                arrived += 1
                break
            wx, wu, wT = xs.copy(), us, T
            wx[0] = xc; wx[-1] = xf[b]
        xfin[b] = xc
    if world > 1:
        dist.barrier()
    rec, done_all, arrived_all, el = rh_collect(rank, world, dist if world > 1 else None, torch.device("cpu"), total, xfin, sT, st, its, done, arrived,
                                                time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "stub (launcher self-test of the rh workload, no GPU work)", "value": total * args.steps / max(el, 1e-9), "unit": "re-solves/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": args.scaling, "stub": True,
                          "resolves_executed": done_all, "instances_arrived": arrived_all,
                          "records": rec.tolist(), "config": {"instances_total": total, "rccl_world_size": world, "backend": "gloo"}}))
    if world > 1:
        dist.destroy_process_group()


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside torchrun: start N ranks as a CHILD process (torch.distributed.run, one rank per GPU, rendezvous on
    127.0.0.1) and forward rank 0's JSON line (the ranks inherit stdout) and the exit code.  Nothing here touches the GPU: a process that
    has initialised it must not be replaced or re-executed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def stub_rank(args, rank, world):
    """CPU stand-in of one rank (tests/test_bench_launcher.py): gloo instead of RCCL, a deterministic fill instead of the device solve —
    everything else (sharding, the single gather, barrier + max-over-ranks timing, the JSON line) is the code path of the real bench."""
    import torch
    import torch.distributed as dist
    from mpc_motion_planner_amd import sharding
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    N, batch = 13, args.batch or 8
    total = sharding.global_total(args.scaling, batch, world)
    sb = sharding.ShardedBatch(total, rank, world, N, torch.device("cpu"), dist if world > 1 else None)
    idx = torch.arange(sb.lo, sb.hi, dtype=torch.float64)
    sol_x = idx[:, None, None] + torch.zeros(sb.count, N, 14, dtype=torch.float64)
    sol_u = -idx[:, None, None] + torch.zeros(sb.count, N, 7, dtype=torch.float64)
    sol_T = 0.5 * idx
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sb.pack_and_gather(sol_x, sol_u, sol_T)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        g = sb.assemble()
        ok = bool(torch.equal(g[:, 0], torch.arange(total, dtype=torch.float64)) and torch.equal(g[:, -1], 0.5 * torch.arange(total, dtype=torch.float64)))
        print(json.dumps({"metric": "stub (launcher self-test, no GPU work)", "value": total * args.steps / max(elapsed, 1e-9), "unit": "records/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": args.scaling, "stub": True, "gather_ok": ok,
                          "config": {"batch": batch, "problems_total": total, "rccl_world_size": world, "backend": "gloo"}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="problems per GPU (weak) or in the whole job (strong); default 1024 (4096 for dual14)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=512, help="problems of the multi-thread CPU leg (single-thread leg: a quarter)")
    ap.add_argument("--warm", choices=["jerk", "quintic"], default="jerk",
                    help="initial guess of every OCP: jerk = the jerk-limited time-synchronised trajectory the reference gets from Ruckig "
                         "(solve_trajectory(true), motionPlanner.cpp:146-175), computed on the GPU inside the timed step; quintic = k_init's fallback")
    ap.add_argument("--workload", choices=["batch", "rh", "shipped", "dual14"], default="batch",
                    help="batch: BASELINE configs[1] (default, the contract line); rh: configs[4] receding horizon; shipped: the reference-as-shipped "
                         "solver depth (N=19, 2 SQP iterations; SURVEY.md 8d); dual14: configs[3], 14-DoF dual-Panda, N=25")
    ap.add_argument("--rh-cold", action="store_true", help="rh workload: both start flags off (-1): every re-solve from lambda = 0 with cold QPs; default = the driver's defaults (carried multipliers, warm QP duals)")
    ap.add_argument("--qp-warm-start", action="store_true", help="mpcmp_config.qp_warm_start = 1 (opt-in: QPs start from the NLP multipliers); the contract line keeps the default 0")
    ap.add_argument("--no-secondary", action="store_true", help="default workload only: skip the brief runs of the three other workloads")
    ap.add_argument("--stub-cpu", action="store_true", help=argparse.SUPPRESS)      # launcher self-test on a CPU-only box (gloo, no solve)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under torchrun: this process only starts the ranks and waits (no GPU call before or after)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE = %d" % (args.gpus, world))
    if args.stub_cpu:
        return stub_rh_rank(args, rank, world) if args.workload == "rh" else stub_rank(args, rank, world)

    import torch
    import mpc_motion_planner_amd as M
    from mpc_motion_planner_amd import scenarios, sharding

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the mpcmp product path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    if args.workload == "rh":
        line = bench_receding_horizon(args, M, scenarios, rank, world, local, dist, flags="cold" if args.rh_cold else None)
    elif args.workload == "dual14":
        import bench_dual14
        line = bench_dual14.run(args, rank, world, local, dist)
    else:
        line = run_batch_workload(args, args.workload, args.steps, args.warmup, rank, world, local, dist, args.cpu_sample, args.batch or 1024)
    if line is not None and args.workload == "batch" and world == 1 and not args.no_secondary:
        # the other three workloads, briefly, inside the ONE contract line: value, ms_per_step, roofline fractions, CPU baseline
        import bench_dual14
        sec = {}
        for name, fn in (("batch_qp_warm_start", lambda: run_batch_workload(args, "batch", 5, 1, 0, 1, local, None, 128, 1024, host_to_host=False, qp_warm_start=1)),
                         ("shipped", lambda: run_batch_workload(args, "shipped", 5, 1, 0, 1, local, None, 128, 1024, host_to_host=False)),
                         ("rh", lambda: bench_receding_horizon(args, M, scenarios, 0, 1, local, None, B=512)),
                         ("rh_cold", lambda: bench_receding_horizon(args, M, scenarios, 0, 1, local, None, flags="cold", B=512)),
                         ("dual14", lambda: bench_dual14.run(args, 0, 1, local, None, steps=2, warmup=1, batch=4096))):
            t0 = time.perf_counter()
            try:
                d = fn()
                sec[name] = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "ms_per_step": d["ms_per_step"],
                             "roofline": {k: d["roofline"].get(k) for k in ("kernel", "frac", "canonical_frac", "avg_launch_ms", "admm_iters_per_traj", "admm_iters_per_resolve", "traffic", "problems_per_launch")
                                          if k in d["roofline"]},
                             "cpu_baseline": {k: d["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "single_thread")} if "cpu_baseline" in d else None,
                             "quality": d.get("quality"), "config": d["config"]["workload"], "wall_s": time.perf_counter() - t0}
                for k in ("live_resolves_per_s", "resolves_executed", "instances_arrived"):
                    if k in d:
                        sec[name][k] = d[k]
            except Exception as e:      # a secondary workload must never cost the contract line
                sec[name] = {"error": repr(e)}
        line["secondary"] = sec
    if line is not None:
        print(json.dumps(_finite(line), allow_nan=False))
    if world > 1:
        dist.destroy_process_group()


def run_batch_workload(args, workload, steps, warmup, rank, world, local, dist, cpu_sample, batch, host_to_host=True, qp_warm_start=None):
    """one batch workload (`batch` = BASELINE.json configs[1], `shipped` = the reference-as-shipped depth) on this rank; rank 0 returns the line"""
    import torch
    import mpc_motion_planner_amd as M
    from mpc_motion_planner_amd import scenarios, sharding
    nseg, sqp, narm, bytes_per_traj, metric = WORKLOADS[workload]
    N = 3 * nseg + 1
    total = sharding.global_total(args.scaling, batch, world)
    dev = torch.device("cuda", local)
    sb = sharding.ShardedBatch(total, rank, world, N, dev, dist)        # this rank's slice [lo, hi) of the global seeded batch
    B = sb.count
    qws = int(args.qp_warm_start if qp_warm_start is None else qp_warm_start)
    cfg = M.default_config(nseg, sqp, margins=MARGINS, qp_warm_start=qws)
    solver = M.Solver(cfg, max(B, 1), device=local)
    x0_h, xf_h = scenarios.make_batch(B, MARGINS, stream_offset=sb.lo)
    x0 = torch.from_numpy(x0_h).to(dev); xf = torch.from_numpy(xf_h).to(dev)
    sol_x = torch.zeros(B, N, 14, dtype=torch.float64, device=dev)
    sol_u = torch.zeros(B, N, 7, dtype=torch.float64, device=dev)
    sol_T = torch.zeros(B, dtype=torch.float64, device=dev)
    info = torch.zeros(B, 64, dtype=torch.uint8, device=dev)               # mpcmp_info records (64 B each)
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
            sb.pack_and_gather(sol_x, sol_u, sol_T)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(warmup):
        step()
    sync()
    solver.kernel_timing(reset=True)          # switches the HIP-event timing of the dominant kernel on
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

    if rank == 0:
        inf = np.frombuffer(info.cpu().numpy().tobytes(), dtype=M.INFO_DTYPE)
        value = total * steps / elapsed
        admm_mean = float(inf["qp_iters_total"].mean())
        flops_traj = canonical_flops(N, sqp, admm_mean, narm)
        flops_qp_traj = flops_traj - sqp * N * 2.0e4 * narm                 # the dominant kernel's share: factorisations + ADMM iterations
        k_avg_s = (k_ms / max(k_launches, 1)) * 1e-3
        # a large batch is solved as several parts on as many streams (mpcmp.hip: solve_impl), so a launch of the dominant kernel
        # covers B / parts problems and `parts` launches are in flight together; durations are HIP events on the launch's own stream
        parts = max(1, round(k_launches / float(steps * sqp)))
        problems_per_launch = B / parts
        flops_launch = problems_per_launch * flops_qp_traj / sqp
        per_gpu = value / world
        achieved = per_gpu * flops_qp_traj / 1e12                           # on the wall clock of the timed region (includes the other kernels)
        alg_bytes_launch = problems_per_launch * bytes_per_traj / sqp
        traffic, mfma_busy, traffic_src = committed_traffic(kname, problems_per_launch, workload)
        peak_meas = measured_fp64_peak()
        # executed FP64 flops: from the committed counter pass (SQ_INSTS_VALU_{FMA,ADD,MUL}_F64 of the QP kernels), per ADMM iteration actually run;
        # the round-1/2 hand count (EXECUTED_FMA) only if no counter profile is committed
        cc = committed_counters(workload)
        ex = EXECUTED_FMA.get(nseg)
        if cc is not None:
            executed = None if cc["stale"] else per_gpu * admm_mean * cc["flops_per_admm_iter"] / 1e12      # (a stale profile yields no fraction: frac null, executed_source.stale true)
        else:
            executed = per_gpu * 2.0 * (admm_mean * ex[0] + sqp * ex[1]) / 1e12 if ex else None
        feasible = (inf["status"] & (1 | 2 | 4 | 16 | 32)) == 0           # inside every tolerance and no hard failure (capped QPs allowed)
        out = {
            "metric": metric, "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d-problem random batch %s, 7-DoF Panda, N=%d Chebyshev nodes (cubic x %d segments), "
                                   "%d SQP iters, <=700 ADMM iters, %s warm start computed in the timed step (%s)"
                                   % (batch, "per GPU" if args.scaling == "weak" else "in the whole job, sliced [r*B/G,(r+1)*B/G)", N, nseg, sqp,
                                      "jerk-limited (Ruckig-equivalent)" if args.warm == "jerk" else "quintic",
                                      "BASELINE.json configs[1]" if workload == "batch" else "reference as shipped: robot_ocp.hpp:32, motionPlanner.cpp:15"),
                       "batch": batch, "problems_total": total, "problems_rank0": B, "rccl_world_size": world, "qp_warm_start": qws,
                       "seed": scenarios.SEED, "margins": list(MARGINS), "timed": "device-resident inputs and outputs (value); host_to_host beside it"},
            # the binding resource is FP64 vector issue + LDS + workgroup barriers: not HBM, not MFMA
            "roofline": {"bound": "fp64_valu", "kernel": kname, "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": (executed / FP64_PEAK_TFLOPS) if executed is not None else None,
                         "frac_def": "FP64 flops the QP kernels executed (instruction counters x 64 lanes, inactive lanes counted: an upper bound) per GPU over the wall clock of the timed region / peak; null when the committed counter profile was measured on other kernel sources (executed_source.stale)",
                         "executed_source": cc if cc is not None else "hand count (bench.py EXECUTED_FMA): no counter profile committed",
                         "canonical_frac": achieved / FP64_PEAK_TFLOPS,
                         "peak_measured": peak_meas, "frac_of_measured_peak": (achieved / peak_meas) if peak_meas else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "mfma_busy": (traffic_src or {}).get("mfma_busy_frac_of_simd_time", 0.0) if mfma_busy else 0.0,
                         "mfma_busy_cycles_per_launch": mfma_busy,
                         "mfma_factor_kernel": committed_mfma(workload),
                         "avg_launch_ms": 1e3 * k_avg_s, "launches": k_launches, "launches_in_flight": parts, "problems_per_launch": problems_per_launch,
                         "canonical_gflop_per_launch": flops_launch / 1e9,
                         "per_launch_tflops": flops_launch / k_avg_s / 1e12 if k_avg_s > 0 else None,
                         "canonical_gflop_per_traj": flops_traj / 1e9, "admm_iters_per_traj": admm_mean,
                         "hbm": {"algorithmic_bytes_per_launch": alg_bytes_launch, "achieved_gbs": alg_bytes_launch / k_avg_s / 1e9 if k_avg_s > 0 else None,
                                 "peak_gbs": HBM_PEAK_GBS, "frac": alg_bytes_launch / k_avg_s / 1e9 / HBM_PEAK_GBS if k_avg_s > 0 else None},
                         "note": "frac = executed flops (counters) / peak; achieved / canonical_frac = canonical dense-equivalent FP64 flops (SURVEY.md 8d) of the "
                                 "QP kernel per GPU over the wall clock of the timed region (an algorithm-speed figure, not a utilisation); "
                                 "`launches_in_flight` launches overlap, so avg_launch_ms is not exclusive GPU time; mfma_busy = fraction of the dominant kernel's SIMD time with a matrix-core instruction executing (committed PMC pass): the ADMM loop has one right-hand side per problem and issues no MFMA, the Schur-complement GEMMs of the factorisation run on the matrix cores (N = 13: inside k_qp2 since round 5); "
                                 "for N >= 19 the factorisation kernel k_qp3f runs the QP's block GEMMs (Schur complement products) on the matrix cores: mfma_factor_kernel"},
            "quality": {**status_fractions(inf["status"]), "feasible_frac": float(feasible.mean()),
                        "feasible_traj_per_s": value * float(feasible.mean()),
                        "T_mean": float(inf["T"].mean()),
                        "defect_inf_median": float(np.median(inf["defect_inf"])),
                        "term_err_inf_median": float(np.median(inf["term_err_inf"])),
                        "path_viol_inf_max": float(inf["path_viol_inf"].max()),
                        "feasible_def": "no hard failure, collocation defect and path violation <= eps_abs (1e-3), terminal error <= eps_target + eps_abs "
                                        "(1.1e-2), T inside its box: status bits 1, 2, 4, 16, 32 clear (include/mpcmp.h)"},
        }
        if world == 1 and host_to_host:
            # host -> host (inputs and outputs in host memory, PCIe inclusive): warm, median of 5 repeats
            ts = []
            for _ in range(6):
                t1 = time.perf_counter()
                solver.solve(x0_h, xf_h, solver.warm_start_jerk(x0_h, xf_h, jmax) if args.warm == "jerk" else None)
                ts.append(time.perf_counter() - t1)
            ts = sorted(ts[1:])
            out["host_to_host"] = {"trajectories_per_s": B / ts[len(ts) // 2], "repeats": len(ts), "min_ms": 1e3 * ts[0], "max_ms": 1e3 * ts[-1],
                                   "note": "SURVEY.md 8(d) defines the metric host->host; `value` is the device-resident rate the bench contract asks for"}
        if world == 1 and not args.no_cpu_baseline:
            cb, T_cpu = cpu_baseline(nseg, sqp, x0_h, xf_h, args.warm, cpu_sample, max(16, cpu_sample // 4), qp_warm_start=qws)
            out["cpu_baseline"] = cb
            out["quality"]["max_rel_dT_vs_cpu_sample"] = float(np.max(np.abs(inf["T"][:len(T_cpu)] - T_cpu) / np.abs(T_cpu)))
        return out
    return None


if __name__ == "__main__":
    main()
