"""Pin the oracle's rigid-body layer against the reference's stored Pinocchio outputs."""
import json
import os

import numpy as np

import oracle_py as o


def test_kat_rnea(golden_dir):
    # 402 (q,qd,qdd)->tau tuples from analysis/data_analysis.ipynb cell 1; inputs have 6 s.f. so the
    # attainable agreement is ~2e-4 N.m (|tau|max 57.8)
    k = np.loadtxt(os.path.join(golden_dir, "kat_rnea.csv"), delimiter=",")
    assert k.shape == (402, 29)
    err = max(np.abs(o.rnea(r[1:8], r[8:15], r[15:22]) - r[22:29]).max() for r in k)
    assert err < 2.5e-4, err


def test_kat_fk_link8(golden_dir):
    # full double precision FK of frame panda_link8 (data_analysis.ipynb cell 3)
    f = np.loadtxt(os.path.join(golden_dir, "kat_fk_link8.csv"), delimiter=",")
    assert f.shape == (201, 10)
    err = max(np.abs(o.fk(r[:7])[2] - r[7:]).max() for r in f)
    assert err < 1e-12, err


def test_kat_jacobians(golden_dir):
    j = json.load(open(os.path.join(golden_dir, "kat_jac.json")))
    J1 = o.frame_jacobian(j["J1"]["q"], [0, 0, 0]) @ np.array(j["J1"]["qd"])
    assert np.abs(J1 - j["J1"]["world_aligned_joint7_velocity"]).max() < 1e-8
    J2 = o.frame_jacobian(j["J2"]["q"], [0, 0, 0]) @ np.array(j["J2"]["qd"])
    assert np.abs(J2 - j["J2"]["world_aligned_joint7_velocity"]).max() < 6e-4
    p7, R7, _, _ = o.fk(j["FK1"]["q"])
    assert np.abs(p7 - j["FK1"]["oMi7_p"]).max() < 1e-6
    assert np.abs(R7 - np.array(j["FK1"]["oMi7_R"])).max() < 1e-6


def test_rnea_derivatives_vs_central_differences():
    rng = np.random.default_rng(7)
    lim = o.default_limits()
    h = 1e-6
    for _ in range(20):
        q = rng.uniform(lim["qmin"], lim["qmax"]); v = rng.uniform(-1, 1, 7) * lim["vmax"]
        a = rng.uniform(-1, 1, 7) * lim["amax"]
        tau, dq, dv, M = o.rnea_derivatives(q, v, a)
        assert np.abs(tau - o.rnea(q, v, a)).max() == 0.0
        E = np.eye(7) * h
        fd = lambda f: np.stack([(f(E[i]) - f(-E[i])) / (2 * h) for i in range(7)], axis=1)
        scale = 1.0 + np.abs(dq).max()
        assert np.abs(fd(lambda d: o.rnea(q + d, v, a)) - dq).max() < 1e-6 * scale
        assert np.abs(fd(lambda d: o.rnea(q, v + d, a)) - dv).max() < 1e-6 * scale
        assert np.abs(fd(lambda d: o.rnea(q, v, a + d)) - M).max() < 1e-6 * scale
        assert np.abs(M - M.T).max() < 1e-13          # mass matrix symmetric
        assert np.linalg.eigvalsh(M).min() > 0        # and positive definite


def test_rnea_derivatives_directional_vs_analytic_closed_form():
    """the reference takes its torque Jacobians from pinocchio::computeRNEADerivatives (analytic, robot_ocp.hpp:118) and the mass matrix from crba
    (:121-122); the solver here (oracle and kernels) propagates directional derivatives through RNEA.  oracle/rbd.c also holds the analytic closed
    form (world-frame spatial algebra, the published formulation) as an independent implementation: both give the same numbers to round-off, over
    the whole joint, velocity and acceleration ranges, and M is symmetric positive definite"""
    rng = np.random.default_rng(7)
    lim = o.default_limits()
    worst = 0.0
    for _ in range(300):
        q = rng.uniform(lim["qmin"], lim["qmax"]); v = rng.uniform(-lim["vmax"], lim["vmax"]); a = rng.uniform(-lim["amax"], lim["amax"])
        tau, dq, dv, M = o.rnea_derivatives(q, v, a)
        tau2, dq2, dv2, M2 = o.rnea_derivatives_analytic(q, v, a)
        for x, y in ((tau, tau2), (dq, dq2), (dv, dv2), (M, M2)):
            worst = max(worst, np.abs(x - y).max() / max(1.0, np.abs(x).max()))
        assert np.array_equal(M2, M2.T) and np.linalg.eigvalsh(M2).min() > 0
        assert np.abs(tau2 - o.rnea(q, v, a)).max() < 1e-12
    assert worst < 1e-13, worst


def test_eval_constraints_layout_and_quirk():
    # robot_ocp.hpp:98-163: rows 0-6 = [dtau/dq, dtau/dqd, M_sym, quirk], row 7 = [dz/dq, 0...]
    rng = np.random.default_rng(3)
    q, v, a = rng.uniform(-1, 1, 7), rng.uniform(-1, 1, 7), rng.uniform(-3, 3, 7)
    x = np.concatenate([q, v])
    g, G = o.eval_constraints(x, a, quirk=1)
    g0, _ = o.eval_constraints(x, a, jac=False)
    assert np.abs(g - g0).max() < 1e-13
    tau, dq, dv, M = o.rnea_derivatives(q, v, a)
    assert np.allclose(G[:7, :7], dq) and np.allclose(G[:7, 7:14], dv)
    assert np.array_equal(G[:7, 14:21], G[:7, 14:21].T)
    assert np.allclose(G[:7, 21], dv @ v + np.triu(M) @ a, atol=1e-12)
    assert np.all(G[7, 7:] == 0)
    _, G0 = o.eval_constraints(x, a, quirk=0)
    assert np.all(G0[:, 21] == 0)
    # height row = d z_tool / dq by central differences
    h = 1e-6
    fd = [(o.fk(q + h * e)[3][2] - o.fk(q - h * e)[3][2]) / (2 * h) for e in np.eye(7)]
    assert np.abs(G[7, :7] - fd).max() < 1e-8
    assert abs(g[7] - o.fk(q)[3][2]) < 1e-15
