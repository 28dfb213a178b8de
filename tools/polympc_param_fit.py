#!/usr/bin/env python3
"""Fit the free parameters of the restated polympc layer (box-ADMM rho / alpha / equality-row scaling / check interval, SQP
depth, line-search constants) to the ONE solve the reference stores: GOLD-TRAJ (analysis/data_analysis.ipynb cell 1 ->
tests/golden/gold_traj.json; figure title "1SQP_700QP_10accel_90_pos": 19 nodes, 700-iteration QP cap, T_mpc 1.55469).

polympc itself is an empty submodule in the reference (.gitmodules:1-4), so its defaults cannot be read; this sweep runs the
CPU oracle (TEST INFRASTRUCTURE, oracle/ocp.c) from the stored Ruckig trajectory (KAT-RK, the warm start the reference
used) for every setting and records
    |T - 1.55469|  and  max |q - q_mpc|, max |v - v_mpc|, max |a - a_mpc|  over the 201 stored samples,
so that the defaults of orc_default_config / mpcmp_default_config are chosen by the fit, not by recollection, and the
residual of the best setting is the tolerance tests/test_oracle_ocp.py asserts.

    python tools/polympc_param_fit.py [--full] [--out profiles/r02_polympc_param_fit.json]
"""
import argparse
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as o  # noqa: E402


def gold():
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "gold_traj.json")))
    x0 = np.array(g["q0"] + g["v0"]); xf = np.array(g["qT"] + g["vT"])
    return g, x0, xf


def rk_nodes(g, x0, nseg):
    """the stored Ruckig trajectory at the collocation nodes (motionPlanner.cpp:156-168)"""
    tn = o.time_nodes(nseg); t = np.array(g["t_rk"])
    q, v, a = np.array(g["q_rk"]), np.array(g["v_rk"]), np.array(g["a_rk"])
    N = len(tn); xg = np.zeros((N, 14)); ug = np.zeros((N, 7))
    for k, tk in enumerate(tn):
        for j in range(7):
            xg[k, j] = np.interp(tk * t[-1], t, q[:, j]); xg[k, 7 + j] = np.interp(tk * t[-1], t, v[:, j])
            ug[k, j] = np.interp(tk * t[-1], t, a[:, j])
    xg[0] = x0
    return xg, ug, t[-1]


def residuals(g, xs, us, T, nseg=6):
    # The stored trajectory ends EXACTLY at the target (14 components, 6 digits) although the terminal box is +-1e-2: it was
    # sampled after solve_trajectory's re-guess "head := current state, tail := target" (motionPlanner.cpp:199-207), which
    # writes into the solver's primal vector that get_MPC_trajectory then interpolates (examples/offline_trajectory.cpp:58,96).
    xs = np.array(xs); xs[0] = np.array(g["q0"] + g["v0"]); xs[-1] = np.array(g["qT"] + g["vT"])
    s = o.sample(nseg, xs, us, T, 200)
    dq = np.abs(s[:, 1:8] - np.array(g["q_mpc"])).max()
    dv = np.abs(s[:, 8:15] - np.array(g["v_mpc"])).max()
    da = np.abs(s[:, 15:22] - np.array(g["a_mpc"])).max()
    return abs(T - g["T_mpc"]), dq, dv, da


def probe(g, x0, xf):
    # the probes are compiled out of liboracle.so (the specification): this tool loads its own build, liboracle_probe.so
    import subprocess
    import oracle_py
    subprocess.check_call(["make", "-C", os.path.join(oracle_py.ROOT, "oracle"), "-B", "liboracle_probe.so"], stdout=subprocess.DEVNULL)
    oracle_py._SO = os.path.join(oracle_py.ROOT, "oracle", "liboracle_probe.so"); oracle_py._lib = None
    lim = o.default_limits(); m = g["margins"]
    xg, ug, Tg = o.warm_start_jerk(6, m[1] * lim["vmax"], m[2] * lim["amax"], m[4] * lim["jmax"], x0, xf)
    def run(mask, **kw):
        os.environ["ORC_PROBE"] = str(mask)
        try:
            cfg = o.default_config(6, 1, margins=g["margins"], **kw)
            xs, us, T, info = o.solve(cfg, x0, xf, xg, ug, Tg)
        finally:
            os.environ.pop("ORC_PROBE", None)
        return xs, us, T, info
    out = {"scenario": "GOLD-TRAJ, N = 19, ONE SQP iteration, QP cap 700, defaults rho 0.02 / alpha 1.4 / rho_eq_scale 1e3 (profiles/r02_polympc_param_fit.json)",
           "adopt_if": "q'' residual < 0.09 rad/s^2 (5 x below the 0.45 floor) with default-looking parameters", "variants": {}}
    names = {0: "specification (oracle default)",
             4: "(i) u_{N-1} pinned to the warm start (its collocation row is absent, SURVEY section 4: the stored control at tau = 1 stays ~0)",
             1: "(iii) variable boxes keep the plain rho even when they pin a variable (x_0): only general equality rows get rho_eq",
             2: "(iv) the QP returns the projected copy z of the step instead of x",
             5: "(i) + (iii)", 6: "(i) + (iv)", 3: "(iii) + (iv)", 7: "(i) + (iii) + (iv)"}
    print("%-100s %-9s %-8s %-8s %-8s %-8s" % ("variant", "T", "dT", "dq", "dv", "da"))
    for mask, nm in names.items():
        xs, us, T, info = run(mask)
        dT, dq, dv, da = residuals(g, xs, us, T)
        out["variants"][nm] = {"ORC_PROBE": mask, "T": T, "dT": dT, "dq": dq, "dv": dv, "da": da, "status": info.status}
        print("%-100s %-9.5f %-8.5f %-8.4f %-8.4f %-8.3f" % (nm[:100], T, dT, dq, dv, da))
    out["variants"]["(ii) duals of the first QP warm-started instead of cold"] = {
        "not_applicable": "the stored solve is ONE SQP iteration from lambda = 0 (figure title 1SQP_700QP, and the fit): its only QP has no previous duals to start from; "
                          "the ADMM multipliers y start at 0 either way"}
    # where the residual of the specification sits: per sample block (node interval) and joint, accelerations
    xs, us, T, info = run(0)
    xs = np.array(xs); xs[0] = np.array(g["q0"] + g["v0"]); xs[-1] = np.array(g["qT"] + g["vT"])
    smp = o.sample(6, xs, us, T, 200)
    ra = np.abs(smp[:, 15:22] - np.array(g["a_mpc"]))
    seg = [float(ra[int(round(200 * k / 18.0)):int(round(200 * (k + 1) / 18.0)) + 1].max()) for k in range(18)]
    out["q_ddot_residual_max_per_node_interval"] = seg
    out["q_ddot_residual_max_per_joint"] = [float(v) for v in ra.max(axis=0)]
    out["q_ddot_residual_argmax_sample"] = int(ra.max(axis=1).argmax())
    print("max |q'' - stored| per node interval:", " ".join("%.2f" % v for v in seg))
    print("per joint:", " ".join("%.2f" % v for v in out["q_ddot_residual_max_per_joint"]), " worst sample", out["q_ddot_residual_argmax_sample"], "of 200")
    json.dump(out, open(os.path.join(ROOT, "profiles", "r03_polympc_structure_probe.json"), "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="finer grid (minutes instead of seconds)")
    ap.add_argument("--warm", choices=["jerk", "stored"], default="jerk",
                    help="jerk: the oracle's Ruckig stand-in at the node times (default); stored: the 201 stored Ruckig samples, interpolated")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_polympc_param_fit.json"))
    ap.add_argument("--probe", action="store_true",
                    help="structure probe (VERDICT r2 item 7): where the best setting's residual sits (per node and joint), and four structural "
                         "hypotheses (oracle/ocp.c ORC_PROBE) against the stored solve; writes profiles/r03_polympc_structure_probe.json")
    args = ap.parse_args()
    g, x0, xf = gold()
    if args.probe:
        return probe(g, x0, xf)
    if args.warm == "stored":
        xg, ug, Tg = rk_nodes(g, x0, 6)
    else:
        # the oracle's jerk-limited generator evaluated AT the node times (what Ruckig's at_time gives the reference,
        # motionPlanner.cpp:156-168); it reproduces the stored Ruckig samples to 1.2e-5 rad (tests/test_oracle_ocp.py) and
        # avoids interpolating the fast-switching accelerations from 201 six-digit samples
        lim = o.default_limits(); m = g["margins"]
        xg, ug, Tg = o.warm_start_jerk(6, m[1] * lim["vmax"], m[2] * lim["amax"], m[4] * lim["jmax"], x0, xf)
    base = residuals(g, xg, ug, Tg)          # the warm start itself, for scale
    rhos = [0.01, 0.03, 0.1, 0.3, 1.0] if not args.full else [0.01, 0.02, 0.03, 0.05, 0.1, 0.2, 0.3, 0.5, 1.0, 2.0]
    alphas = [1.0, 1.6] if not args.full else [1.0, 1.2, 1.4, 1.6, 1.8]
    eqs = [1.0, 1e3] if not args.full else [1.0, 10.0, 1e2, 1e3]
    sqps = [1, 2]
    checks = [25] if not args.full else [1, 25]
    sigmas = [1e-6]
    rows = []
    for rho, al, eq, sqp, chk, sg in itertools.product(rhos, alphas, eqs, sqps, checks, sigmas):
        cfg = o.default_config(6, sqp, margins=g["margins"], rho=rho, alpha=al, rho_eq_scale=eq, check_every=chk, sigma=sg)
        xs, us, T, info = o.solve(cfg, x0, xf, xg, ug, Tg)
        dT, dq, dv, da = residuals(g, xs, us, T)
        rows.append({"rho": rho, "alpha": al, "rho_eq_scale": eq, "sqp_iters": sqp, "check_every": chk, "sigma": sg, "T": T, "dT": dT,
                     "dq": dq, "dv": dv, "da": da, "qp_iters": info.qp_iters_total, "defect_inf": info.defect_inf,
                     "term_err_inf": info.term_err_inf, "status": info.status})
    # rank: node distance first (scaled by the warm start's own distance), T second
    def score(r):
        return r["dq"] / base[1] + r["dv"] / base[2] + r["da"] / base[3] + r["dT"] / 0.0218
    rows.sort(key=score)
    cur = o.default_config(6, 2, margins=g["margins"])
    out = {"scenario": "GOLD-TRAJ, N=19, QP cap 700, warm start = Ruckig trajectory at the nodes (T 1.57649), source: " + args.warm,
           "stored_T_mpc": g["T_mpc"], "warm_start_residuals": {"dT": base[0], "dq": base[1], "dv": base[2], "da": base[3]},
           "score": "dq/dq_warm + dv/dv_warm + da/da_warm + dT/0.0218", "n_settings": len(rows),
           "current_defaults": {"rho": cur.rho, "alpha": cur.alpha, "rho_eq_scale": cur.rho_eq_scale, "sigma": cur.sigma, "check_every": cur.check_every},
           "best": rows[:12], "all": rows if args.full else None}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print("warm start: dT %.5f dq %.4f dv %.4f da %.3f" % base)
    print("%-6s %-5s %-7s %-3s %-4s | %-9s %-8s %-7s %-7s %-7s %-6s" % ("rho", "alpha", "eqscale", "sqp", "chk", "T", "dT", "dq", "dv", "da", "iters"))
    for r in rows[:16]:
        print("%-6g %-5g %-7g %-3d %-4d | %-9.5f %-8.5f %-7.4f %-7.4f %-7.3f %-6d" % (r["rho"], r["alpha"], r["rho_eq_scale"], r["sqp_iters"], r["check_every"],
                                                                                      r["T"], r["dT"], r["dq"], r["dv"], r["da"], r["qp_iters"]))


if __name__ == "__main__":
    main()
