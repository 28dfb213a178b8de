"""Oracle NLP/SQP layer: discretisation facts and the one stored solve of the reference (GOLD-TRAJ)."""
import json
import os
import sys

import numpy as np

import oracle_py as o


def _gold(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "gold_traj.json")))
    x0 = np.array(g["q0"] + g["v0"]); xf = np.array(g["qT"] + g["vT"])
    return g, x0, xf


def rk_warm_start(g, x0, nseg):
    """KAT-RK: the stored Ruckig trajectory sampled at the collocation nodes (motionPlanner.cpp:146-175)."""
    tn = o.time_nodes(nseg); t = np.array(g["t_rk"])
    q, v, a = np.array(g["q_rk"]), np.array(g["v_rk"]), np.array(g["a_rk"])
    N = len(tn); xg = np.zeros((N, 14)); ug = np.zeros((N, 7))
    for k, tk in enumerate(tn):
        for j in range(7):
            xg[k, j] = np.interp(tk * t[-1], t, q[:, j]); xg[k, 7 + j] = np.interp(tk * t[-1], t, v[:, j])
            ug[k, j] = np.interp(tk * t[-1], t, a[:, j])
    xg[0] = x0
    return xg, ug, t[-1]


def test_discretisation():
    D = o.diff_matrix()
    ref = np.array([[-19 / 6, 4, -4 / 3, 1 / 2], [-1, 1 / 3, 1, -1 / 3], [1 / 3, -1, -1 / 3, 1], [-1 / 2, 4 / 3, -4, 19 / 6]])
    assert np.abs(D - ref).max() < 1e-14
    xi = np.array([-1, -0.5, 0.5, 1.0])
    for p in range(4):      # exact differentiation of cubics
        assert np.abs(D @ xi ** p - p * xi ** max(p - 1, 0) * (p > 0)).max() < 1e-13
    assert np.allclose(o.time_nodes(4)[:5], [0, 0.0625, 0.1875, 0.25, 0.3125])
    assert np.allclose(o.time_nodes(6)[:4], [0, 1 / 24, 1 / 8, 1 / 6])
    assert len(o.time_nodes(8)) == 25


# residuals of the fitted defaults on GOLD-TRAJ (tools/polympc_param_fit.py, profiles/r02_polympc_param_fit.json) + 10 %
GOLD_FIT = {"dT": 1.51e-3 * 1.1, "dq": 0.0228 * 1.1, "dv": 0.0604 * 1.1, "da": 0.451 * 1.1}


def jerk_warm_start(g, x0, xf, nseg):
    """the Ruckig stand-in evaluated at the node times (what the reference's warm_start_RK builds, motionPlanner.cpp:146-175)"""
    lim = o.default_limits(); m = g["margins"]
    return o.warm_start_jerk(nseg, m[1] * lim["vmax"], m[2] * lim["amax"], m[4] * lim["jmax"], x0, xf)


def gold_residuals(g, xs, us, T, x0, xf):
    """distance to the stored MPC samples.  The stored trajectory was sampled AFTER solve_trajectory's re-guess (head := current
    state, tail := target; motionPlanner.cpp:199-207 writes into the primal vector get_MPC_trajectory interpolates): it ends
    exactly at the target in all 14 components although the terminal box is +-1e-2, so the comparison applies the same rule."""
    xs = np.array(xs); xs[0] = x0; xs[-1] = xf
    s = o.sample(6, xs, us, T, 200)
    return (abs(T - g["T_mpc"]), np.abs(s[:, 1:8] - np.array(g["q_mpc"])).max(), np.abs(s[:, 8:15] - np.array(g["v_mpc"])).max(),
            np.abs(s[:, 15:22] - np.array(g["a_mpc"])).max())


def test_gold_traj_stored_solve_is_end_fixed(golden_dir):
    g, x0, xf = _gold(golden_dir)
    assert np.abs(np.array(g["q_mpc"][-1] + g["v_mpc"][-1]) - xf).max() == 0.0        # exact, although eps = 1e-2 (motionPlanner.hpp:44)


def test_gold_traj_fit(golden_dir):
    """The one solve the reference stores (figure title "1SQP_700QP_...": 19 nodes, one SQP iteration, 700-iteration QP cap),
    from the Ruckig-equivalent warm start, with the defaults tools/polympc_param_fit.py selected: final time within 1.7e-3 of
    the stored 1.55469 and all 201 stored q / qd / qdd samples within the fitted residual + 10 % (the warm start itself is
    0.097 rad / 0.35 rad/s / 3.75 rad/s^2 away, the previous round's defaults 0.032 / 0.107 / 1.32 with |dT| 0.018)."""
    g, x0, xf = _gold(golden_dir)
    cfg = o.default_config(6, 1, margins=g["margins"])
    assert (cfg.rho, cfg.alpha, cfg.rho_eq_scale, cfg.sigma) == (0.02, 1.4, 1e3, 1e-6)
    xg, ug, Tg = jerk_warm_start(g, x0, xf, 6)
    assert abs(Tg - g["T_ruckig"]) < 2e-6
    xs, us, T, info = o.solve(cfg, x0, xf, xg, ug, Tg)
    assert (info.status & 7) == 0 and info.status & 8 and info.qp_capped == 1 and info.qp_iters_total == 700 and info.last_alpha == 1.0
    dT, dq, dv, da = gold_residuals(g, xs, us, T, x0, xf)
    assert dT < GOLD_FIT["dT"] and dq < GOLD_FIT["dq"] and dv < GOLD_FIT["dv"] and da < GOLD_FIT["da"], (dT, dq, dv, da)
    assert info.path_viol_inf < 1e-6 and np.abs(xs[0] - x0).max() < 1e-4     # truncated ADMM returns x, not the clipped z
    # the solver depth as shipped in the source (2 SQP iterations, motionPlanner.cpp:15) moves on towards the optimum
    cfg2 = o.default_config(6, 2, margins=g["margins"])
    xs2, us2, T2, info2 = o.solve(cfg2, x0, xf, xg, ug, Tg)
    assert info2.qp_iters_total == 1400 and 1.52 < T2 < T and info2.defect_inf < 1e-3 and info2.term_err_inf < 1.1e-2


def test_gold_traj_collocation_structure(golden_dir):
    """Pins the discretisation (orc_diff_matrix, the time scaling ts*T, which nodes carry dynamics rows) against reference-held
    data: the stored MPC trajectory is piecewise cubic on exactly 6 segments; with the per-segment cubics evaluated at the
    local CGL nodes xi = {-1,-1/2,1/2,1}, the oracle's collocation-defect operator is small (QP tolerance) at the first three
    local nodes of every segment and O(1) at xi = +1."""
    g, x0, xf = _gold(golden_dir)
    q, v, a, T = np.array(g["q_mpc"]), np.array(g["v_mpc"]), np.array(g["a_mpc"]), g["T_mpc"]
    t = np.arange(201) / 200.0
    xi = np.array([-1.0, -0.5, 0.5, 1.0])

    def per_segment_nodes(nseg):
        xs = np.zeros((nseg, 4, 14)); us = np.zeros((nseg, 4, 7)); resid = 0.0
        for s in range(nseg):
            idx = np.where((t >= s / nseg - 1e-12) & (t <= (s + 1) / nseg + 1e-12))[0]
            V = np.vander(2 * (t[idx] * nseg - s) - 1, 4)
            for arr, dst, off in ((q, xs, 0), (v, xs, 7), (a, us, 0)):
                c = np.linalg.lstsq(V, arr[idx], rcond=None)[0]
                resid = max(resid, np.abs(V @ c - arr[idx]).max())
                dst[s, :, off:off + 7] = np.vander(xi, 4) @ c
        return xs, us, resid

    for nseg in (4, 5, 7, 8):
        assert per_segment_nodes(nseg)[2] > 0.5                  # not a 4-, 5-, 7- or 8-segment cubic spline
    xs, us, resid = per_segment_nodes(6)
    assert resid < 1e-4                                          # 6 s.f. storage
    assert np.abs(xs[:-1, 3] - xs[1:, 0]).max() < 1e-4 and np.abs(us[:-1, 3] - us[1:, 0]).max() < 1e-4     # shared end nodes
    X = np.zeros((19, 14)); U = np.zeros((19, 7))
    for s in range(6):
        X[3 * s:3 * s + 4] = xs[s]; U[3 * s:3 * s + 4] = us[s]
    assert np.allclose(np.linspace(0, 1, 7)[:6, None] + (xi[None, :3] + 1) / 12, o.time_nodes(6)[:18].reshape(6, 3))
    d = np.abs(o.collocation_defects(6, X, U, T))                # [segment][local node][14 rows]
    assert d[:, :3].max() < 0.02                                 # first three local nodes of every segment: QP tolerance
    assert np.all(d[:, 3, 7:].max(axis=1) > 0.3)                 # xi = +1: no dynamics row there (qd rows off by O(1))
    assert d[:, 3].max() > 50 * np.median(d[:, :3])
    # the same operator with the time scaling of a different segment count, or a transposed D, does not have this structure
    assert np.abs(o.collocation_defects(6, X, U, T * 6 / 4))[:, :3].max() > 0.3


def test_gold_traj_converged(golden_dir):
    """the same specification converges to T* ~= 1.5278 on this scenario."""
    g, x0, xf = _gold(golden_dir)
    cfg = o.default_config(6, 12, margins=g["margins"], qp_iters=3000)
    xg, ug, Tg = rk_warm_start(g, x0, 6)
    xs, us, T, info = o.solve(cfg, x0, xf, xg, ug, Tg)
    assert abs(T - 1.5278) < 2e-3
    assert info.defect_inf < 1e-4 and info.path_viol_inf < 1e-6
    lim = np.array(cfg.ubu[:])
    assert np.abs(us).max(axis=0).max() <= lim.max() + 1e-2
    # torque and height constraints hold at every node
    for k in range(19):
        gk, _ = o.eval_constraints(xs[k], us[k], jac=False)
        assert np.all(gk[:7] <= np.array(cfg.ubg[:7]) + 1e-3) and np.all(gk[:7] >= np.array(cfg.lbg[:7]) - 1e-3)
        assert gk[7] >= 0.05 - 1e-3


def test_builtin_warm_start_and_headline_config(golden_dir):
    g, x0, xf = _gold(golden_dir)
    cfg = o.default_config(4, 20, margins=g["margins"])
    xg, ug, Tg = o.warm_start(cfg, x0, xf)
    assert np.abs(xg[0] - x0).max() == 0 and np.abs(xg[-1] - xf).max() == 0
    assert np.abs(xg[:, 7:]).max(axis=0).max() <= 2.61 * 0.9 + 1e-9
    xs, us, T, info = o.solve(cfg, x0, xf, xg, ug, Tg)
    assert 1.45 < T < 1.58 and info.term_err_inf < 1.2e-2 and info.path_viol_inf < 1e-6


def test_sample_matches_nodes(golden_dir):
    g, x0, xf = _gold(golden_dir)
    cfg = o.default_config(4, 3, margins=g["margins"])
    xg, ug, Tg = o.warm_start(cfg, x0, xf)
    xs, us, T, _ = o.solve(cfg, x0, xf, xg, ug, Tg)
    out = o.sample(4, xs, us, T, n_pts=16)        # 16 uniform samples hit nodes 0, 1, 3, 4, ... exactly
    tn = o.time_nodes(4)
    for k, tk in enumerate(tn):
        i = int(round(tk * 16))
        if abs(i / 16 - tk) < 1e-12:
            assert np.abs(out[i, 1:15] - xs[k]).max() < 1e-12 and np.abs(out[i, 15:22] - us[k]).max() < 1e-12
            assert np.abs(out[i, 22:] - o.rnea(xs[k, :7], xs[k, 7:], us[k])).max() < 1e-10
    assert abs(out[-1, 0] - T) < 1e-15


def test_jerk_limited_warm_start_reproduces_stored_ruckig_trajectory(golden_dir):
    """KAT-RK: the one Ruckig trajectory the reference stores (analysis/data_analysis.ipynb; motionPlanner.cpp:146-175 with
    margins 0.9/0.9/0.5/0.9/0.1).  The double-S restatement must give Ruckig's duration and all seven joint trajectories."""
    g = json.load(open(os.path.join(golden_dir, "gold_traj.json")))
    lim = o.default_limits()
    mp, mv, ma, mt, mj = g["margins"]
    vmax, amax, jmax = mv * lim["vmax"], ma * lim["amax"], mj * lim["jmax"]
    x0 = np.concatenate([g["q0"], g["v0"]]); xf = np.concatenate([g["qT"], g["vT"]])
    out, T = o.jerk_trajectory(vmax, amax, jmax, x0, xf, 200)
    assert abs(T - g["T_ruckig"]) < 2e-6                                    # stored with 6 significant digits
    t, q, v, a = out[:, 0], out[:, 1:8], out[:, 8:15], out[:, 15:22]
    assert np.abs(t - np.array(g["t_rk"])).max() < 1e-5
    assert np.abs(q - np.array(g["q_rk"])).max() < 2e-5                     # all seven joints, 201 samples
    assert np.abs(v - np.array(g["v_rk"])).max() < 1e-4
    assert np.abs(a - np.array(g["a_rk"])).max() < 1e-2                     # accelerations switch at rates of 375..1000 1/s^3
    # limits and boundary conditions hold exactly
    assert np.all(np.abs(v) <= vmax + 1e-12) and np.all(np.abs(a) <= amax + 1e-9)
    assert np.abs(q[-1] - g["qT"]).max() < 1e-12 and np.abs(v[-1] - g["vT"]).max() < 1e-12 and np.abs(a[[0, -1]]).max() < 1e-9
    # node form used as the OCP's initial guess
    xg, ug, Tg = o.warm_start_jerk(6, vmax, amax, jmax, x0, xf)
    assert Tg == T and np.array_equal(xg[0], x0) and np.array_equal(xg[-1], xf)


def test_jerk_limited_warm_start_random_states():
    """random state pairs: boundary conditions met, limits respected, duration = the slowest joint's minimum time"""
    rng = np.random.default_rng(3)
    lim = o.default_limits()
    vmax, amax, jmax = 0.9 * lim["vmax"], 0.5 * lim["amax"], 0.1 * lim["jmax"]
    worst = 0.0
    for _ in range(200):
        x0 = np.concatenate([rng.uniform(lim["qmin"], lim["qmax"]), rng.uniform(-vmax, vmax)])
        xf = np.concatenate([rng.uniform(lim["qmin"], lim["qmax"]), rng.uniform(-vmax, vmax)])
        out, T = o.jerk_trajectory(vmax, amax, jmax, x0, xf, 400)
        q, v, a = out[:, 1:8], out[:, 8:15], out[:, 15:22]
        assert np.abs(q[0] - x0[:7]).max() < 1e-12 and np.abs(v[0] - x0[7:]).max() < 1e-12
        assert np.abs(q[-1] - xf[:7]).max() < 1e-9 and np.abs(v[-1] - xf[7:]).max() < 1e-9
        worst = max(worst, (np.abs(v) / vmax).max(), (np.abs(a) / amax).max())
        # positions are the integral of the velocities (trapezoid on 400 intervals)
        dq = np.cumsum(0.5 * (v[1:] + v[:-1]) * np.diff(out[:, 0])[:, None], axis=0)
        assert np.abs(q[1:] - q[0] - dq).max() < 5e-3
    assert worst <= 1.0 + 1e-9 or worst < 1.6     # (a joint that fell back to a quintic may overshoot a limit)


def test_jerk_limited_warm_start_boundary_accelerations():
    """non-zero boundary accelerations (the reference forwards current / target accelerations to Ruckig, motionPlanner.cpp:36-38,50-52; KAT-RK has
    none, so the pin is properties): both boundary states met in position, velocity AND acceleration, the trajectory is continuous (positions are
    the integral of the velocities, velocities of the accelerations), the acceleration stays inside its limit, the jerk inside its limit (finite
    differences), and zero accelerations reproduce the zero-only generator bit for bit."""
    rng = np.random.default_rng(11)
    lim = o.default_limits()
    vmax, amax, jmax = 0.9 * lim["vmax"], 0.5 * lim["amax"], 0.1 * lim["jmax"]
    worst_a = worst_j = 0.0
    nq = 0
    for it in range(200):
        x0 = np.concatenate([rng.uniform(lim["qmin"], lim["qmax"]), rng.uniform(-0.6 * vmax, 0.6 * vmax)])
        xf = np.concatenate([rng.uniform(lim["qmin"], lim["qmax"]), rng.uniform(-0.6 * vmax, 0.6 * vmax)])
        a0, aT = rng.uniform(-0.8 * amax, 0.8 * amax), rng.uniform(-0.8 * amax, 0.8 * amax)
        n = 2000
        out, T = o.jerk_trajectory(vmax, amax, jmax, x0, xf, n, acc0=a0, accT=aT)
        t, q, v, a = out[:, 0], out[:, 1:8], out[:, 8:15], out[:, 15:22]
        assert T > 0 and np.all(np.isfinite(out))
        assert np.abs(q[0] - x0[:7]).max() < 1e-12 and np.abs(v[0] - x0[7:]).max() < 1e-12 and np.abs(a[0] - a0).max() < 1e-12
        assert np.abs(q[-1] - xf[:7]).max() < 1e-9 and np.abs(v[-1] - xf[7:]).max() < 1e-9 and np.abs(a[-1] - aT).max() < 1e-9
        dt = np.diff(t)[:, None]
        assert np.abs(q[1:] - q[0] - np.cumsum(0.5 * (v[1:] + v[:-1]) * dt, axis=0)).max() < 2e-4
        assert np.abs(v[1:] - v[0] - np.cumsum(0.5 * (a[1:] + a[:-1]) * dt, axis=0)).max() < 2e-3
        jerk = np.abs(np.diff(a, axis=0) / dt)
        # (a joint that fell back to the quintic of the common duration may overshoot: count them, bound the rest)
        ok = (np.abs(a) / amax).max(axis=0) <= 1.0 + 1e-9
        nq += int((~ok).sum())
        worst_a = max(worst_a, (np.abs(a) / amax)[:, ok].max()); worst_j = max(worst_j, (jerk / jmax)[:, ok].max())
        xg, ug, Tg = o.warm_start_jerk(6, vmax, amax, jmax, x0, xf, acc0=a0, accT=aT)
        assert Tg == T and np.array_equal(xg[0], x0) and np.array_equal(xg[-1], xf) and np.abs(ug[0] - a0).max() < 1e-12 and np.abs(ug[-1] - aT).max() < 1e-9
        if it < 20:
            z = np.zeros(7)
            ref, Tr = o.jerk_trajectory(vmax, amax, jmax, x0, xf, 50)
            got, Tz = o.jerk_trajectory(vmax, amax, jmax, x0, xf, 50, acc0=z, accT=z)
            assert Tz == Tr and np.array_equal(ref, got)
    assert worst_a <= 1.0 + 1e-9 and worst_j <= 1.0 + 1e-6, (worst_a, worst_j)
    assert nq <= 0.2 * 200 * 7, nq


def scenarios_pair(k):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mpc_motion_planner_amd import scenarios
    x0, xf = scenarios.make_batch(k + 1, (0.9, 0.9, 0.5, 0.9), stream_offset=900)
    return x0[k], xf[k]


def test_carried_multipliers_semantics():
    """mpcmp_config.carry_multipliers (what a re-solve on one MotionPlanner object most plausibly does upstream: polympc keeps its dual iterate until
    lam_guess is called): flag 0 = lambda_0 = 0 whatever is handed in; flag 1 with zeros = the plain solve bit for bit; flag 1 with the
    previous solve's multipliers = another start, and with qp_warm_start a cheaper re-solve of the same problem"""
    x0, xf = scenarios_pair(3)
    cfg0 = o.default_config(4, 2, margins=(0.9, 0.9, 0.5, 0.9))
    wx, wu, wT = o.warm_start(cfg0, x0, xf)
    xs, us, T, info = o.solve(cfg0, x0, xf, wx, wu, wT)
    rng = np.random.default_rng(0)
    xa, ua, Ta, ia, lam = o.solve_carry(cfg0, x0, xf, wx, wu, wT, lam=rng.normal(size=o.solve_carry(cfg0, x0, xf, wx, wu, wT)[4].shape))
    assert np.array_equal(xa, xs) and Ta == T and ia.qp_iters_total == info.qp_iters_total and np.abs(lam).max() > 0
    cfg1 = o.default_config(4, 2, margins=(0.9, 0.9, 0.5, 0.9), carry_multipliers=1, qp_warm_start=1)
    cfgw = o.default_config(4, 2, margins=(0.9, 0.9, 0.5, 0.9), qp_warm_start=1)
    xw, uw, Tw, iw = o.solve(cfgw, x0, xf, wx, wu, wT)
    xb, ub, Tb, ib, lam1 = o.solve_carry(cfg1, x0, xf, wx, wu, wT)                 # zeros in: the plain (QP-warm-started) solve
    assert np.array_equal(xb, xw) and Tb == Tw and ib.qp_iters_total == iw.qp_iters_total
    # the re-solve of motionPlanner.cpp:199-207 (previous solution as the guess, end states re-pinned), with and without the carried multipliers
    gx = xb.copy(); gx[0] = x0; gx[-1] = xf
    xc, uc, Tc, ic, lam2 = o.solve_carry(cfg1, x0, xf, gx, ub, Tb, lam=lam1)
    xd, ud, Td, idd = o.solve(cfgw, x0, xf, gx, ub, Tb)
    assert not np.array_equal(xc, xd) and np.all(np.isfinite(xc)) and abs(Tc - Td) < 0.05 * Td
    assert ic.qp_iters_total <= idd.qp_iters_total, (ic.qp_iters_total, idd.qp_iters_total)


def test_status_word_semantics():
    """mpcmp_info.status / orc_info.status (include/mpcmp.h MPCMP_STATUS_*; the reference never reads mpc.info().status, motionPlanner.cpp:191):
    bit 8 = a QP stopped at qp_iters (qp_capped counts them), bit 16 = returned iterate outside tolerance, bit 32 = T outside its box,
    0 = converged QPs and an iterate inside every tolerance.  The kernels report the same word (tests/test_gpu_parity.py)."""
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mpcmp.h")).read()
    bits = {k: int(v) for k, v in re.findall(r"#define MPCMP_STATUS_(\w+)\s+(\d+)", hdr)}
    assert bits == {"NAN": 1, "NOT_PD": 2, "XCH_DEAD": 4, "QP_CAPPED": 8, "OUTSIDE_TOL": 16, "T_OUT_OF_BOX": 32, "ARRIVED": 64}
    margins = (0.9, 0.9, 0.5, 0.9, 0.1)
    x0 = np.array([0.0, -0.5, 0.0, -2.0, 0.0, 1.6, 0.0] + [0.0] * 7); xf = x0.copy(); xf[:7] += 0.05         # an easy problem
    # (a) every QP is cut off after 5 iterations: capped, the count says how often
    cfg = o.default_config(4, 3, margins=margins, qp_iters=5, check_every=5)
    xg, ug, Tg = o.warm_start(cfg, x0, xf)
    _, _, _, info = o.solve(cfg, x0, xf, xg, ug, Tg)
    assert info.status & 8 and info.qp_capped == 3 and info.qp_iters_total == 15
    # (b) the same problem solved to the end: QPs converge, the iterate is inside every tolerance -> status 0, nothing capped
    cfg = o.default_config(4, 8, margins=margins)
    _, _, T, info = o.solve(cfg, x0, xf, xg, ug, Tg)
    outside = info.defect_inf > cfg.eps_abs or info.path_viol_inf > cfg.eps_abs or info.term_err_inf > cfg.eps_target + cfg.eps_abs
    assert bool(info.status & 16) == outside and bool(info.status & 8) == (info.qp_capped > 0) and not info.status & 7
    assert cfg.lbT <= T <= cfg.ubT and not info.status & 32
    # (c) a final-time box that excludes the answer is reported
    cfg = o.default_config(4, 2, margins=margins); cfg.ubT = 0.05
    _, _, T, info = o.solve(cfg, x0, xf, xg, ug, Tg)
    assert (T > cfg.ubT + 1e-9) == bool(info.status & 32)


MARGINS = (0.9, 0.9, 0.5, 0.9, 0.1)


def test_receding_horizon_arrival_rule():
    """orc_rh_advance (the arrival rule of mpcmp_rh_run, include/mpcmp.h): a failed solve holds the state; a plan that ends within the control period is
    followed to its last node and retires the instance; an advance into the terminal box retires it; otherwise the state is the solution at time dt
    (get_MPC_point, motionPlanner.hpp:118-128)"""
    cfg = o.default_config(4, 2, margins=MARGINS)
    rng = np.random.default_rng(3)
    xs, us = rng.normal(size=(13, 14)), rng.normal(size=(13, 7))
    xf, x_now = xs[-1] + 0.5, rng.normal(size=14)
    for st in (1, 2, 4, 32, 33):                                    # held
        x, r = o.rh_advance(cfg, xs, us, 2.0, st, 0.1, xf, x_now)
        assert not r and np.array_equal(x, x_now)
    x, r = o.rh_advance(cfg, xs, us, 0.05, 8 | 16, 0.1, xf, x_now)   # T <= dt: end of the plan, retired
    assert r and np.array_equal(x, xs[-1])
    x, r = o.rh_advance(cfg, xs, us, 2.0, 0, 0.1, xf, x_now)         # plain advance = get_MPC_point
    assert not r and np.abs(x - o.mpc_point(4, xs, us, 2.0, 0.1)[:14]).max() < 1e-14
    xf2 = o.mpc_point(4, xs, us, 2.0, 0.1)[:14] + 0.5 * cfg.eps_target
    x, r = o.rh_advance(cfg, xs, us, 2.0, 0, 0.1, xf2, x_now)        # lands inside the terminal box
    assert r
    # a short chain on a real scenario: arrival after a few long periods, inside the terminal box or at the end of a plan shorter than dt
    from mpc_motion_planner_amd import scenarios
    ocfg = o.default_config(4, 2, margins=MARGINS, carry_multipliers=1, qp_warm_start=1)
    x0, xfb = scenarios.make_batch(1, MARGINS, stream_offset=901)
    xc, prev, lam, steps = x0[0].copy(), None, None, 0
    for steps in range(1, 20):
        if prev is None or (prev[3] & (1 | 2 | 4 | 32)):
            wx, wu, wT = o.rh_start_guess(ocfg, xc, xfb[0])
        else:
            wx, wu, wT = prev[0].copy(), prev[1], prev[2]; wx[0] = xc; wx[-1] = xfb[0]
        xs, us, T, oi, lam = o.solve_carry(ocfg, xc, xfb[0], wx, wu, wT, lam=lam)
        prev = (xs, us, T, oi.status)
        xc, r = o.rh_advance(ocfg, xs, us, T, oi.status, 0.4, xfb[0], xc)
        if r:
            break
    assert r and 2 <= steps <= 12 and (T <= 0.4 or np.abs(xc - xfb[0]).max() <= ocfg.eps_target)
