"""Oracle NLP/SQP layer: discretisation facts and the one stored solve of the reference (GOLD-TRAJ)."""
import json
import os

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


def test_gold_traj_regime(golden_dir):
    """Reference-as-shipped config (19 nodes, 2 SQP, 700 ADMM): T must land between the converged optimum
    and the Ruckig warm start, feasible at the nodes and inside the terminal box (regime-level parity; the
    digit-level result depends on unknowable polympc defaults — SURVEY.md B.4)."""
    g, x0, xf = _gold(golden_dir)
    cfg = o.default_config(6, 2, margins=g["margins"])
    xg, ug, Tg = rk_warm_start(g, x0, 6)
    assert abs(Tg - g["T_ruckig"]) < 1e-12
    xs, us, T, info = o.solve(cfg, x0, xf, xg, ug, Tg)
    assert info.status == 0 and info.qp_iters_total == 1400
    assert 1.52 < T < g["T_ruckig"]
    assert abs(T - g["T_mpc"]) < 0.03            # stored solve: 1.55469
    assert info.path_viol_inf < 1e-6 and info.term_err_inf < 1.1e-2 and info.defect_inf < 1e-3
    assert np.abs(xs[0] - x0).max() < 1e-4       # truncated ADMM returns x, not the clipped z


def test_gold_traj_converged(golden_dir):
    """the same specification converges to T* ~= 1.5278 on this scenario."""
    g, x0, xf = _gold(golden_dir)
    cfg = o.default_config(6, 8, margins=g["margins"])
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
