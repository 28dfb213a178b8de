#!/usr/bin/env python3
"""Hypothesis probe: does the reference's boxADMM (polympc, an empty submodule: .gitmodules:1-4) adapt rho the way OSQP does?
Runs the CPU oracle (TEST INFRASTRUCTURE, oracle/ocp.c; the adaptive rule is switched on by the environment variable ORC_ADAPT =
tolerance) on the reference's one stored solve (GOLD-TRAJ, tests/golden/gold_traj.json) and prints the residuals against the stored
201 samples next to those of a fixed rho.  Result: every adaptive setting is 3x - 50x further from the stored
trajectory than the fitted fixed-rho setting, most of them so bad that the line search rejects the step: not adopted.

    python tools/polympc_adaptive_rho_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_py as o  # noqa: E402
import polympc_param_fit as F  # noqa: E402

g, x0, xf = F.gold()
lim = o.default_limits(); m = g["margins"]
xg, ug, Tg = o.warm_start_jerk(6, m[1] * lim["vmax"], m[2] * lim["amax"], m[4] * lim["jmax"], x0, xf)
print("warm start itself: dT %.5f dq %.4f dv %.4f da %.3f" % F.residuals(g, xg, ug, Tg))
for adapt in (None, 5.0, 2.0, 1.2):
    if adapt is None:
        os.environ.pop("ORC_ADAPT", None)
    else:
        os.environ["ORC_ADAPT"] = str(adapt)
    for rho in (0.02, 0.1, 1.0):
        for alpha in (1.0, 1.4, 1.6):
            for eq in (1.0, 1e3):
                c = o.default_config(6, 1, margins=tuple(m[:4]))
                c.rho = rho; c.alpha = alpha; c.rho_eq_scale = eq
                xs, us, T, info = o.solve(c, x0, xf, xg, ug, Tg)
                r = F.residuals(g, xs, us, T)
                print("adaptive tolerance %-4s rho %-4g alpha %.1f eq-scale %-6g -> dT %.5f dq %.4f dv %.4f da %.3f" % (adapt, rho, alpha, eq, r[0], r[1], r[2], r[3]))
os.environ.pop("ORC_ADAPT", None)
