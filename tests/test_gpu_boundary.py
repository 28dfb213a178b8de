"""GPU tests of the drop-in boundary beyond the raw solve: the C++ MotionPlanner shim (warm_start / solve_trajectory /
get_MPC_point with its clamp / get_RK_point), the Python BatchMotionPlanner, stream semantics of the *_device entry
points, and the full-size shapes of BASELINE configs #3 (8,192 problems per GPU) and #5 (512 receding-horizon instances,
two-stream hipGraph capture).  Run with -m gpu."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_py as o

pytestmark = pytest.mark.gpu

MARGINS = (0.9, 0.9, 0.5, 0.9, 0.1)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def M():
    import mpc_motion_planner_amd as M
    return M


def _cfgs(M, nseg, sqp, **kw):
    return M.default_config(nseg, sqp, margins=MARGINS, **kw), o.default_config(nseg, sqp, margins=MARGINS, **kw)


def _jerk_limits():
    lim = o.default_limits()
    return MARGINS[1] * lim["vmax"], MARGINS[2] * lim["amax"], MARGINS[4] * lim["jmax"]


def test_cpp_shim_warm_start_solve_points_vs_oracle(M, tmp_path):
    """examples/shim_selftest.cpp drives the header-only MotionPlanner like a reference caller; every number it prints is
    recomputed by the oracle: warm_start (motionPlanner.hpp:145-172, nearest-sample pick) -> solve_trajectory(false) ->
    get_MPC_point below and beyond T (the clamp of motionPlanner.hpp:120-121) -> solve_trajectory(true) -> get_RK_point
    (motionPlanner.hpp:130-142, clamped to the duration) -> robot.data / frame_id look-alike."""
    exe = str(tmp_path / "shim_selftest")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "shim_selftest.cpp"),
                           "-L" + os.path.join(ROOT, "mpc_motion_planner_amd"), "-lmpcmp",
                           "-Wl,-rpath," + os.path.join(ROOT, "mpc_motion_planner_amd"), "-o", exe])
    r = json.loads(subprocess.check_output([exe]).decode())
    x0, xf = np.array(r["x0"]), np.array(r["xf"])
    _, ocfg = _cfgs(M, 4, 3)
    # the guess the shim's warm_start() builds: straight line over 41 samples, picked at round(tau * 40)
    nP, Tg, tn = 41, 2.0, o.time_nodes(4)
    idx = np.floor(tn * (nP - 1) + 0.5).astype(int)            # std::round: halves away from zero (0.0625 * 40 = 2.5 -> 3)
    line = x0[None, :7] + (xf[:7] - x0[:7])[None, :] * (np.arange(nP) / (nP - 1.0))[:, None]
    xg = np.concatenate([line[idx], np.tile((xf[:7] - x0[:7]) / Tg, (13, 1))], axis=1); ug = np.zeros((13, 7))
    xs, us, T, oi = o.solve(ocfg, x0, xf, xg, ug, Tg)
    assert abs(r["T_warm"] - T) <= 1e-6 * T and r["iters_warm"] == oi.qp_iters_total
    assert r["mpc_p"] == r["T_warm"] and r["mpc_iter"] == 3
    ref_in = o.mpc_point(4, xs, us, T, 0.37 * r["T_warm"])
    assert np.abs(np.array(r["mpc_point_in"]) - ref_in).max() < 1e-6
    ref_beyond = o.mpc_point(4, xs, us, T, r["T_warm"] + 0.3)      # normalised time := T (> 1): extrapolation of the last segment
    assert np.abs(np.array(r["mpc_point_beyond"]) - ref_beyond).max() < 1e-5 * (1 + np.abs(ref_beyond).max())
    assert np.abs(ref_beyond[:14] - xs[-1]).max() > 1e-3           # i.e. NOT the terminal state: the quirk is kept
    # solve_trajectory(true): jerk-limited warm start
    vmax, amax, jmax = _jerk_limits()
    wx, wu, wT = o.warm_start_jerk(4, vmax, amax, jmax, x0, xf)
    xs2, us2, T2, oi2 = o.solve(ocfg, x0, xf, wx, wu, wT)
    assert abs(r["T_rk_solve"] - T2) <= 1e-6 * T2 and r["iters_rk"] == oi2.qp_iters_total
    tr, Trk = o.jerk_trajectory(vmax, amax, jmax, x0, xf, 1000)
    i = int(round(0.4 / Trk * 1000))
    assert abs(tr[i, 0] - 0.4) < Trk / 1000
    pin = np.array(r["rk_point_in"])
    assert np.abs(pin[:14] - tr[i, 1:15]).max() < 0.05              # coarse: neighbouring sample of the oracle trajectory
    assert np.abs(pin[21:] - o.rnea(pin[:7], pin[7:14], pin[14:21])).max() < 1e-9      # torque = RNEA of the point itself
    pb = np.array(r["rk_point_beyond"])                             # time clamped to the duration: the target, at rest in acceleration
    assert np.abs(pb[:14] - xf).max() < 1e-9 and np.abs(pb[14:21]).max() < 1e-6
    assert np.abs(pb[21:] - o.rnea(xf[:7], xf[7:], pb[14:21])).max() < 1e-9
    assert abs(r["tool_z"] - o.fk(xf[:7])[3][2]) < 1e-12
    # write side of `mpc` (motionPlanner.hpp:28-29): the same guess handed over through x_guess / u_guess / p_guess + mpc.solve()
    # is the same solve, bit for bit
    assert r["T_guess_api"] == r["T_warm"] and r["iters_guess_api"] == r["iters_warm"]
    # mpc.control_bounds (motionPlanner.cpp:75) at 40 % of the limits, solved from the previous solution with its end states
    # re-pinned (motionPlanner.cpp:199-207): the oracle with the same box
    lim = o.default_limits()
    _, obox = _cfgs(M, 4, 3)
    for j in range(7):
        obox.lbu[j] = -0.4 * lim["amax"][j]; obox.ubu[j] = 0.4 * lim["amax"][j]
    gx = xs.copy(); gx[0] = x0; gx[-1] = xf
    xs3, us3, T3, oi3 = o.solve(obox, x0, xf, gx, us, T)
    assert abs(r["T_ctrl_box"] - T3) <= 1e-6 * T3 and r["iters_ctrl_box"] == oi3.qp_iters_total
    assert np.abs(np.array(r["u_max_ctrl_box"]) - np.abs(us3).max(axis=0) / lim["amax"]).max() < 1e-6
    assert np.abs(us3).max() > 0 and (np.abs(us3).max(axis=0) / lim["amax"]).max() < 0.45       # the box is what binds
    # the Ruckig members: otg.calculate(input, trajectory) / trajectory.at_time are get_RK_point's trajectory; caller-written
    # limits in the input record are honoured (half the jerk, 80 % of the velocity: the oracle generator with those limits)
    assert r["otg_result"] == 0 and abs(r["rk_duration"] - Trk) <= 1e-9 * Trk
    assert np.abs(np.array(r["rk_at_time"]) - pin[:21]).max() == 0.0
    _, Tslow = o.jerk_trajectory(0.8 * vmax, amax, 0.5 * jmax, x0, xf, 10)
    assert abs(r["rk_duration_slow"] - Tslow) <= 1e-9 * Tslow and Tslow > Trk
    # a non-positive limit: calculate() returns Error (-1) and leaves no trajectory behind; `input` is honoured in full by solve_trajectory(true)
    assert r["otg_result_bad"] == -1 and r["at_time_threw"] == 1
    _, Tin = o.jerk_trajectory(0.8 * vmax, amax, jmax, x0, xf, 10, acc0=0.3 * amax, accT=np.zeros(7))
    assert abs(r["rk_duration_input"] - Tin) <= 1e-9 * Tin and r["guess_T_input"] == r["rk_duration_input"] and abs(r["rk_a0_ratio"] - 0.3) <= 1e-12


def test_python_batch_motion_planner_vs_oracle(M):
    """mpc_motion_planner_amd.BatchMotionPlanner keeps MotionPlanner's member names with a leading batch axis."""
    from mpc_motion_planner_amd import scenarios
    B = 5
    x0, xf = scenarios.make_batch(B, MARGINS, stream_offset=900)
    pl = M.BatchMotionPlanner(None, max_batch=B, num_seg=4, sqp_iters=3)
    pl.set_constraint_margins(*MARGINS)
    pl.set_current_state(x0[:, :7], x0[:, 7:]); pl.set_target_state(xf[:, :7], xf[:, 7:])
    assert np.all(pl.check_state_in_bounds(x0[:, :7], x0[:, 7:]) == 0)
    info = pl.solve_trajectory(True)                               # jerk-limited warm start, as the reference's `true`
    _, ocfg = _cfgs(M, 4, 3)
    vmax, amax, jmax = _jerk_limits()
    sx, su, sT = pl.solution()
    t_q = np.array([0.3, 0.5, 1.0, 5.0, 0.01])                      # problem 3: beyond T (clamp)
    q, v, a, tau = pl.get_MPC_point(t_q)
    qr, vr, ar, taur = pl.get_RK_point(t_q)
    for b in range(B):
        wx, wu, wT = o.warm_start_jerk(4, vmax, amax, jmax, x0[b], xf[b])
        xs, us, T, oi = o.solve(ocfg, x0[b], xf[b], wx, wu, wT)
        assert abs(sT[b] - T) <= 1e-6 * T and np.abs(sx[b] - xs).max() <= 1e-6 and info["qp_iters_total"][b] == oi.qp_iters_total
        ref = o.mpc_point(4, xs, us, T, t_q[b])
        got = np.concatenate([q[b], v[b], a[b], tau[b]])
        assert np.abs(got - ref).max() < 1e-5 * (1 + np.abs(ref).max()), b
        assert np.abs(taur[b] - o.rnea(qr[b], vr[b], ar[b])).max() < 1e-9
    assert sT[3] < 5.0 and np.abs(qr[3] - xf[3, :7]).max() < 1e-9   # get_RK_point clamps to the duration
    # solve_trajectory(false) after a solve: re-guess with exact end states (motionPlanner.cpp:199-207)
    info2 = pl.solve_trajectory(False)
    for b in (0, 4):
        gx = sx[b].copy(); gx[0] = x0[b]; gx[-1] = xf[b]
        xs, us, T, oi = o.solve(ocfg, x0[b], xf[b], gx, su[b], sT[b])
        assert abs(pl.solution()[2][b] - T) <= 1e-6 * T and info2["qp_iters_total"][b] == oi.qp_iters_total
    # generic warm_start from a regularly spaced trajectory
    tr, _ = pl._solver.jerk_trajectory(x0, xf, jmax, 100)
    pl.warm_start(tr[:, -1, 0], tr[..., 1:8], tr[..., 8:15], tr[..., 15:22])
    pl.solve_trajectory(False)
    idx = np.floor(o.time_nodes(4) * 100 + 0.5).astype(int)
    b = 2
    xs, us, T, _ = o.solve(ocfg, x0[b], xf[b], tr[b, idx, 1:15], tr[b, idx, 15:22], tr[b, -1, 0])
    assert abs(pl.solution()[2][b] - T) <= 1e-6 * T
    with pytest.raises(ValueError):
        pl._solver.solve(x0, xf[:, :13])                            # mismatched shapes are refused before the C call


def test_device_entry_points_share_the_given_stream(M):
    """warm_start_jerk_batch_device followed by solve_batch_device on the SAME stream handle (0 = the legacy default stream,
    what torch.cuda.current_stream().cuda_stream is) with fresh states on every call: the solve must see this call's warm
    start, never the previous one (round-1 advisor finding)."""
    import torch
    from mpc_motion_planner_amd import scenarios
    cfg, _ = _cfgs(M, 4, 2)
    B, N = 256, 13
    s = M.Solver(cfg, B)
    dev = torch.device("cuda", 0)
    jmax = MARGINS[4] * M.default_limits()["jmax"]
    wx = torch.zeros(B, N, 14, dtype=torch.float64, device=dev); wu = torch.zeros(B, N, 7, dtype=torch.float64, device=dev)
    wT = torch.zeros(B, dtype=torch.float64, device=dev)
    sx = torch.zeros(B, N, 14, dtype=torch.float64, device=dev); su = torch.zeros(B, N, 7, dtype=torch.float64, device=dev)
    sT = torch.zeros(B, dtype=torch.float64, device=dev)
    for rep, stream in enumerate((0, 0, torch.cuda.Stream(dev).cuda_stream)):
        x0_h, xf_h = scenarios.make_batch(B, MARGINS, stream_offset=5000 + 1000 * rep)
        x0 = torch.from_numpy(x0_h).to(dev); xf = torch.from_numpy(xf_h).to(dev)
        torch.cuda.synchronize(dev)
        s.warm_start_jerk_device(B, x0.data_ptr(), xf.data_ptr(), jmax, wx.data_ptr(), wu.data_ptr(), wT.data_ptr(), stream=stream)
        s.solve_device(B, x0.data_ptr(), xf.data_ptr(), sx.data_ptr(), su.data_ptr(), sT.data_ptr(),
                       warm=(wx.data_ptr(), wu.data_ptr(), wT.data_ptr()), stream=stream)
        torch.cuda.synchronize(dev)
        hx, hu, hT, _ = s.solve(x0_h, xf_h, s.warm_start_jerk(x0_h, xf_h, jmax))      # host-buffer path, internally ordered
        assert np.array_equal(sT.cpu().numpy(), hT), rep
        assert np.array_equal(sx.cpu().numpy(), hx) and np.array_equal(su.cpu().numpy(), hu)


def test_set_config_invalidates_captured_graph(M):
    """a captured receding-horizon step holds the configuration by value: after set_config the next replay must use the
    new bounds (round-1 advisor finding)."""
    from mpc_motion_planner_amd import scenarios
    cfg, _ = _cfgs(M, 4, 2)
    B, dt = 4, 0.02
    x0, xf = scenarios.make_batch(B, MARGINS, stream_offset=60)
    tight = M.default_config(4, 2, margins=(0.9, 0.9, 0.25, 0.9, 0.1))              # half the acceleration range
    res = {}
    for mode in ("graph", "eager"):
        s = M.Solver(cfg, B); s.rh_init(x0, xf)
        s.rh_run(3, dt, use_graph=(mode == "graph"))
        s.set_config(tight)
        s.rh_run(3, dt, use_graph=(mode == "graph"))
        res[mode] = s.rh_get()
    assert np.array_equal(res["graph"][0], res["eager"][0]) and np.array_equal(res["graph"][3], res["eager"][3])
    s = M.Solver(cfg, B); s.rh_init(x0, xf); s.rh_run(6, dt, use_graph=True)                  # same six steps, bounds never changed
    assert not np.array_equal(s.rh_get()[3], res["graph"][3])                                # i.e. the new bounds did take effect


def test_timing_is_off_by_default_and_bounded(M):
    cfg, _ = _cfgs(M, 4, 1, qp_iters=25)
    from mpc_motion_planner_amd import scenarios
    x0, xf = scenarios.make_batch(8, MARGINS)
    s = M.Solver(cfg, 8)
    for _ in range(3):
        s.solve(x0, xf)
    name, ms, n = s.kernel_timing(reset=True)
    assert name == "k_qp2" and n == 0 and ms == 0.0                # nothing was recorded before the first call
    s.solve(x0, xf)
    name, ms, n = s.kernel_timing(reset=True)
    assert n == 1 and ms > 0.0


def test_config3_share_of_one_gpu_full_size(M):
    """BASELINE.json configs[2]: 65,536 problems over 8 GPUs = 8,192 per GPU (N=13, 20 SQP).  Size-independent properties at
    that size on one GPU: every problem reported, bitwise reproducible, results independent of the batch a problem travels
    in (the same problems as the first 1,024 of the global seeded batch), oracle parity on a sample."""
    cfg, ocfg = _cfgs(M, 4, 20)
    from mpc_motion_planner_amd import scenarios
    B = 8192
    x0, xf = scenarios.make_batch(B, MARGINS)
    s = M.Solver(cfg, B)
    jmax = MARGINS[4] * M.default_limits()["jmax"]
    warm = s.warm_start_jerk(x0, xf, jmax)
    sx, su, sT, info = s.solve(x0, xf, warm)
    assert np.all(np.isfinite(sT)) and np.all(np.isfinite(sx)) and np.all(info["sqp_iters"] == 20)
    assert ((info["status"] & 7) == 0).mean() >= 0.999      # no hard failure (NaN, factorisation, exchange)
    sx2, su2, sT2, info2 = s.solve(x0, xf, warm)
    assert np.array_equal(sT, sT2) and np.array_equal(sx, sx2) and np.array_equal(info["qp_iters_total"], info2["qp_iters_total"])
    sub = slice(4096, 4096 + 512)
    sxs, _, sTs, _ = s.solve(x0[sub], xf[sub], tuple(w[sub] for w in warm))
    assert np.array_equal(sTs, sT[sub]) and np.array_equal(sxs, sx[sub])
    vmax, amax, jm = _jerk_limits()
    for b in (0, 4095, 8191):
        wx, wu, wT = o.warm_start_jerk(4, vmax, amax, jm, x0[b], xf[b])
        xs, us, T, oi = o.solve(ocfg, x0[b], xf[b], wx, wu, wT)
        assert abs(sT[b] - T) <= 1e-6 * abs(T) and np.abs(sx[b] - xs).max() <= 1e-6 and info["qp_iters_total"][b] == oi.qp_iters_total
    ok = (info["defect_inf"] < 1e-3) & (info["path_viol_inf"] < 1e-3) & (info["term_err_inf"] <= 1.1e-2)
    assert ok.mean() > 0.5


def test_receding_horizon_at_config5_shape(M):
    """BASELINE.json configs[4] shape: 512 instances, so that the step is the two-stream capture path (B >= 512).  Six
    graph-replayed steps equal six eagerly enqueued ones bit for bit, and three instances are followed by the oracle."""
    cfg = M.default_config(4, 2, margins=MARGINS)                   # (the driver's defaults: carried multipliers, warm QP duals)
    ocfg = o.default_config(4, 2, margins=MARGINS, carry_multipliers=1, qp_warm_start=1)
    from mpc_motion_planner_amd import scenarios
    B, steps, dt = 512, 6, 0.01
    x0, xf = scenarios.make_batch(B, MARGINS, stream_offset=2000)
    out = {}
    for mode in (False, True):
        s = M.Solver(cfg, B); s.rh_init(x0, xf)
        s.rh_run(steps, dt, use_graph=mode)
        out[mode] = s.rh_get()
    for k in range(4):
        assert np.array_equal(out[False][k], out[True][k]), k
    assert np.array_equal(out[False][4]["qp_iters_total"], out[True][4]["qp_iters_total"])
    for b in (0, 255, 511):                                         # one from each half-batch, and the last
        xc = x0[b].copy(); prev = None; lam = None
        for st in range(steps):
            if prev is None or (prev[3] & (1 | 2 | 4 | 32)):
                wx, wu, wT = o.rh_start_guess(ocfg, xc, xf[b])
            else:
                wx, wu, wT = prev[0].copy(), prev[1], prev[2]
                wx[0] = xc; wx[-1] = xf[b]
            xs, us, T, oi, lam = o.solve_carry(ocfg, xc, xf[b], wx, wu, wT, lam=lam)
            prev = (xs, us, T, oi.status)
            xc, retired = o.rh_advance(ocfg, xs, us, T, oi.status, dt, xf[b], xc)
            assert not retired
        assert np.abs(out[True][0][b] - xc).max() < 1e-6 and abs(out[True][3][b] - T) < 1e-6
        assert out[True][4]["qp_iters_total"][b] == oi.qp_iters_total
