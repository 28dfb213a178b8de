"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (see oracle/oracle.h)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
_SO = os.path.join(ROOT, "oracle", "liboracle.so")

NJ, NX, NU, NG = 7, 14, 7, 8


class Model(C.Structure):
    _fields_ = [("R0", C.c_double * 9 * NJ), ("p", C.c_double * 3 * NJ), ("mass", C.c_double * NJ),
                ("com", C.c_double * 3 * NJ), ("I", C.c_double * 9 * NJ), ("tool", C.c_double * 3),
                ("link8", C.c_double * 3), ("gravity", C.c_double * 3)]


class Config(C.Structure):
    _fields_ = [("num_seg", C.c_int), ("sqp_iters", C.c_int), ("qp_iters", C.c_int), ("ls_iters", C.c_int),
                ("check_every", C.c_int), ("quirk_dtau_dT", C.c_int),
                ("eps_abs", C.c_double), ("eps_rel", C.c_double),
                ("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double), ("rho_eq_scale", C.c_double),
                ("ls_eta", C.c_double), ("ls_tau", C.c_double), ("hess_reg", C.c_double), ("eps_target", C.c_double),
                ("lbx", C.c_double * NX), ("ubx", C.c_double * NX), ("lbu", C.c_double * NU), ("ubu", C.c_double * NU),
                ("lbg", C.c_double * NG), ("ubg", C.c_double * NG), ("lbT", C.c_double), ("ubT", C.c_double),
                ("qp_warm_start", C.c_int), ("carry_multipliers", C.c_int)]


class Info(C.Structure):
    _fields_ = [("T", C.c_double), ("viol_l1", C.c_double), ("defect_inf", C.c_double),
                ("path_viol_inf", C.c_double), ("term_err_inf", C.c_double), ("last_alpha", C.c_double),
                ("qp_iters_total", C.c_int), ("sqp_iters", C.c_int), ("status", C.c_int), ("qp_capped", C.c_int)]


INFO_DTYPE = np.dtype([("T", "f8"), ("viol_l1", "f8"), ("defect_inf", "f8"), ("path_viol_inf", "f8"),
                       ("term_err_inf", "f8"), ("last_alpha", "f8"), ("qp_iters_total", "i4"),
                       ("sqp_iters", "i4"), ("status", "i4"), ("qp_capped", "i4")])


def build(force=False):
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("rbd.c", "ocp.c", "jerk.c", "oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_num_nodes.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def default_model():
    m = Model(); lib().orc_default_model(C.byref(m)); return m


def default_limits():
    out = [np.zeros(7) for _ in range(6)]
    lib().orc_default_limits(*[_p(o) for o in out])
    return dict(zip(["qmin", "qmax", "vmax", "amax", "jmax", "taumax"], out))


def default_config(num_seg=4, sqp_iters=20, margins=None, **kw):
    c = Config(); lib().orc_default_config(C.byref(c), num_seg, sqp_iters)
    if margins is not None:
        lib().orc_set_margins(C.byref(c), *[C.c_double(x) for x in margins[:4]])
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def rnea(q, v, a, model=None):
    model = model or default_model(); q, v, a = f64(q), f64(v), f64(a); tau = np.zeros(7)
    lib().orc_rnea(C.byref(model), _p(q), _p(v), _p(a), _p(tau)); return tau


def rnea_derivatives(q, v, a, model=None):
    model = model or default_model(); q, v, a = f64(q), f64(v), f64(a)
    tau, dq, dv, M = np.zeros(7), np.zeros((7, 7)), np.zeros((7, 7)), np.zeros((7, 7))
    lib().orc_rnea_derivatives(C.byref(model), _p(q), _p(v), _p(a), _p(tau), _p(dq), _p(dv), _p(M))
    return tau, dq, dv, M


def rnea_derivatives_analytic(q, v, a, model=None):
    """closed-form partials in world-frame spatial algebra (the formulation of pinocchio::computeRNEADerivatives / crba): independent of rnea_derivatives"""
    model = model or default_model(); q, v, a = f64(q), f64(v), f64(a)
    tau, dq, dv, M = np.zeros(7), np.zeros((7, 7)), np.zeros((7, 7)), np.zeros((7, 7))
    lib().orc_rnea_derivatives_analytic(C.byref(model), _p(q), _p(v), _p(a), _p(tau), _p(dq), _p(dv), _p(M))
    return tau, dq, dv, M


def fk(q, model=None):
    model = model or default_model(); q = f64(q)
    p7, R7, p8, pt = np.zeros(3), np.zeros((3, 3)), np.zeros(3), np.zeros(3)
    lib().orc_fk(C.byref(model), _p(q), _p(p7), _p(R7), _p(p8), _p(pt))
    return p7, R7, p8, pt


def frame_jacobian(q, off, model=None):
    model = model or default_model(); q, off = f64(q), f64(off); J = np.zeros((6, 7))
    lib().orc_frame_jacobian(C.byref(model), _p(q), _p(off), _p(J)); return J


def eval_constraints(x, u, quirk=1, jac=True, model=None):
    model = model or default_model(); x, u = f64(x), f64(u); g = np.zeros(8)
    G = np.zeros((8, 22)) if jac else None
    lib().orc_eval_constraints(C.byref(model), int(quirk), _p(x), _p(u), _p(g), _p(G)); return g, G


def time_nodes(num_seg):
    t = np.zeros(3 * num_seg + 1); lib().orc_time_nodes(num_seg, _p(t)); return t


def diff_matrix():
    D = np.zeros((4, 4)); lib().orc_diff_matrix(_p(D)); return D


def warm_start(cfg, x0, xf, amax_used=None):
    N = 3 * cfg.num_seg + 1
    x0, xf = f64(x0), f64(xf)
    amax_used = f64(amax_used if amax_used is not None else np.array(cfg.ubu[:]))
    xg, ug, T = np.zeros((N, 14)), np.zeros((N, 7)), C.c_double(0)
    lib().orc_warm_start(C.byref(cfg), _p(amax_used), _p(x0), _p(xf), _p(xg), _p(ug), C.byref(T))
    return xg, ug, T.value


def warm_start_jerk(num_seg, vmax, amax, jmax, x0, xf, acc0=None, accT=None):
    """jerk-limited time-synchronised warm start (oracle/jerk.c): node states, node controls, duration; acc0 / accT: boundary accelerations [7]"""
    N = 3 * num_seg + 1
    vmax, amax, jmax, x0, xf = f64(vmax), f64(amax), f64(jmax), f64(x0), f64(xf)
    a0 = f64(acc0) if acc0 is not None else None; aT = f64(accT) if accT is not None else None
    xg, ug, T = np.zeros((N, 14)), np.zeros((N, 7)), C.c_double(0)
    lib().orc_warm_start_jerk_acc(int(num_seg), _p(vmax), _p(amax), _p(jmax), _p(x0), _p(xf), _p(a0) if a0 is not None else None,
                                  _p(aT) if aT is not None else None, _p(xg), _p(ug), C.byref(T))
    return xg, ug, T.value


def jerk_trajectory(vmax, amax, jmax, x0, xf, n_pts=200, acc0=None, accT=None):
    """uniform samples (n_pts+1) x 22 = t, q, v, a of the same trajectory, and its duration"""
    vmax, amax, jmax, x0, xf = f64(vmax), f64(amax), f64(jmax), f64(x0), f64(xf)
    a0 = f64(acc0) if acc0 is not None else None; aT = f64(accT) if accT is not None else None
    out, T = np.zeros((n_pts + 1, 22)), C.c_double(0)
    lib().orc_jerk_trajectory_acc(_p(vmax), _p(amax), _p(jmax), _p(x0), _p(xf), _p(a0) if a0 is not None else None,
                                  _p(aT) if aT is not None else None, int(n_pts), _p(out), C.byref(T))
    return out, T.value


def solve(cfg, x0, xf, xg, ug, Tg, model=None):
    model = model or default_model(); N = 3 * cfg.num_seg + 1
    x0, xf, xg, ug = f64(x0), f64(xf), f64(xg), f64(ug)
    xs, us, T, info = np.zeros((N, 14)), np.zeros((N, 7)), C.c_double(0), Info()
    lib().orc_solve(C.byref(model), C.byref(cfg), _p(x0), _p(xf), _p(xg), _p(ug), C.c_double(Tg),
                    _p(xs), _p(us), C.byref(T), C.byref(info))
    return xs, us, T.value, info


def solve_carry(cfg, x0, xf, xg, ug, Tg, lam=None, model=None):
    """a re-solve on one planner object: lam [m + n] in / out (None: zeros); the start only when cfg.carry_multipliers is set.  Returns xs, us, T, info, lam"""
    model = model or default_model(); N = 3 * cfg.num_seg + 1
    lib().orc_num_multipliers.restype = C.c_int
    mn = lib().orc_num_multipliers(C.byref(cfg), 1)
    lam = np.zeros(mn) if lam is None else f64(lam).copy()
    assert lam.shape == (mn,)
    x0, xf, xg, ug = f64(x0), f64(xf), f64(xg), f64(ug)
    xs, us, T, info = np.zeros((N, 14)), np.zeros((N, 7)), C.c_double(0), Info()
    lib().orc_solve_carry(C.byref(model), 1, C.byref(cfg), _p(x0), _p(xf), _p(xg), _p(ug), C.c_double(Tg), _p(lam), _p(xs), _p(us), C.byref(T), C.byref(info))
    return xs, us, T.value, info, lam


def solve_batch(cfg, x0, xf, xg, ug, Tg, threads=1, model=None):
    model = model or default_model(); N = 3 * cfg.num_seg + 1
    x0, xf, xg, ug, Tg = f64(x0), f64(xf), f64(xg), f64(ug), f64(Tg)
    B = x0.shape[0]
    xs, us, T = np.zeros((B, N, 14)), np.zeros((B, N, 7)), np.zeros(B)
    info = np.zeros(B, dtype=INFO_DTYPE)
    lib().orc_solve_batch(C.byref(model), C.byref(cfg), B, _p(x0), _p(xf), _p(xg), _p(ug), _p(Tg),
                          _p(xs), _p(us), _p(T), info.ctypes.data_as(C.c_void_p), int(threads))
    return xs, us, T, info


# ---- multi-arm form (BASELINE.json configs[3]: 14-DoF dual Panda, the same OCP with doubled sizes) ----
def arm_models(bases):
    """ctypes array of Models: the compiled-in Panda mounted at each base = (yaw about world z, [x, y, z]); the base placement
    is folded into the first joint placement (R0[0] <- Rz(yaw) R0[0], p[0] <- base + Rz(yaw) p[0])."""
    arr = (Model * len(bases))()
    for a, (yaw, xyz) in enumerate(bases):
        m = default_model()
        c, s = np.cos(yaw), np.sin(yaw)
        Rz = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        R0 = np.array(m.R0).reshape(7, 3, 3); p = np.array(m.p).reshape(7, 3)
        R0[0] = Rz @ R0[0]; p[0] = np.asarray(xyz, dtype=np.float64) + Rz @ p[0]
        for i in range(9):
            m.R0[0][i] = R0[0].ravel()[i]
        for i in range(3):
            m.p[0][i] = p[0][i]
        arr[a] = m
    return arr


DUAL_BASES = ((0.0, (0.0, 0.0, 0.0)), (np.pi, (1.0, 0.0, 0.0)))      # two Pandas facing each other, 1 m apart


def solve_multi(models, cfg, x0, xf, xg, ug, Tg):
    narm = len(models); N = 3 * cfg.num_seg + 1
    x0, xf, xg, ug = f64(x0), f64(xf), f64(xg), f64(ug)
    assert x0.shape == (14 * narm,) and xg.shape == (N, 14 * narm) and ug.shape == (N, 7 * narm)
    xs, us, T, info = np.zeros((N, 14 * narm)), np.zeros((N, 7 * narm)), C.c_double(0), Info()
    lib().orc_solve_multi(models, narm, C.byref(cfg), _p(x0), _p(xf), _p(xg), _p(ug), C.c_double(Tg), _p(xs), _p(us), C.byref(T), C.byref(info))
    return xs, us, T.value, info


def solve_carry_multi(models, cfg, x0, xf, xg, ug, Tg, lam=None):
    """multi-arm form of solve_carry: lam [m + n] in / out (None: zeros).  Returns xs, us, T, info, lam"""
    narm = len(models); N = 3 * cfg.num_seg + 1
    lib().orc_num_multipliers.restype = C.c_int
    mn = lib().orc_num_multipliers(C.byref(cfg), narm)
    lam = np.zeros(mn) if lam is None else f64(lam).copy()
    x0, xf, xg, ug = f64(x0), f64(xf), f64(xg), f64(ug)
    xs, us, T, info = np.zeros((N, 14 * narm)), np.zeros((N, 7 * narm)), C.c_double(0), Info()
    lib().orc_solve_carry(models, narm, C.byref(cfg), _p(x0), _p(xf), _p(xg), _p(ug), C.c_double(Tg), _p(lam), _p(xs), _p(us), C.byref(T), C.byref(info))
    return xs, us, T.value, info, lam


def solve_batch_multi(models, cfg, x0, xf, xg, ug, Tg, threads=1):
    narm = len(models); N = 3 * cfg.num_seg + 1
    x0, xf, xg, ug, Tg = f64(x0), f64(xf), f64(xg), f64(ug), f64(Tg)
    B = x0.shape[0]
    xs, us, T = np.zeros((B, N, 14 * narm)), np.zeros((B, N, 7 * narm)), np.zeros(B)
    info = np.zeros(B, dtype=INFO_DTYPE)
    lib().orc_solve_batch_multi(models, narm, C.byref(cfg), B, _p(x0), _p(xf), _p(xg), _p(ug), _p(Tg), _p(xs), _p(us), _p(T),
                                info.ctypes.data_as(C.c_void_p), int(threads))
    return xs, us, T, info


def debug_qp_multi(models, cfg, x0, xf, xs, us, T, lam=None):
    narm = len(models); N = 3 * cfg.num_seg + 1
    n, m = 21 * N * narm + 1, (14 * (N - 1) + 8 * N) * narm
    x0, xf, xs, us = f64(x0), f64(xf), f64(xs), f64(us)
    lam = f64(lam) if lam is not None else None
    p, y = np.zeros(n), np.zeros(m + n)
    lib().orc_debug_qp_multi.restype = C.c_int
    it = lib().orc_debug_qp_multi(models, narm, C.byref(cfg), _p(x0), _p(xf), _p(xs), _p(us), C.c_double(T), _p(lam), _p(p), _p(y))
    return p, y, it


def merge_arm_states(xa):
    """per-arm states [narm][14] = [q(7); qd(7)] -> the multi-arm layout [q(7 narm); qd(7 narm)]"""
    xa = np.asarray(xa, dtype=np.float64)
    return np.concatenate([xa[:, :7].ravel(), xa[:, 7:].ravel()])


def warm_start_jerk_multi(num_seg, vmax, amax, jmax, x0, xf):
    """Warm start of the multi-arm OCP from the single-arm generator: every arm's jerk-limited trajectory is computed on its own,
    the common duration is the slowest arm's, and a faster arm's trajectory is played back uniformly slower (time scaling
    s = T_arm / T <= 1: q(t) = q_arm(s t), qd = s qd_arm, qdd = s^2 qdd_arm — all limits still hold)."""
    x0, xf = f64(x0), f64(xf); narm = x0.shape[0] // 14; nq = 7 * narm; N = 3 * num_seg + 1
    per = []
    for a in range(narm):
        xa0 = np.concatenate([x0[7 * a:7 * a + 7], x0[nq + 7 * a:nq + 7 * a + 7]])
        xaf = np.concatenate([xf[7 * a:7 * a + 7], xf[nq + 7 * a:nq + 7 * a + 7]])
        per.append(warm_start_jerk(num_seg, vmax, amax, jmax, xa0, xaf))
    T = max(p[2] for p in per)
    xg, ug = np.zeros((N, 14 * narm)), np.zeros((N, 7 * narm))
    for a, (xa, ua, Ta) in enumerate(per):
        s = Ta / T
        xg[:, 7 * a:7 * a + 7] = xa[:, :7]; xg[:, nq + 7 * a:nq + 7 * a + 7] = s * xa[:, 7:]
        ug[:, 7 * a:7 * a + 7] = s * s * ua
    xg[0] = x0; xg[-1] = xf                     # exact end states (motionPlanner.cpp:202-203)
    return xg, ug, T


def sample(num_seg, xs, us, T, n_pts=200, model=None):
    model = model or default_model(); xs, us = f64(xs), f64(us)
    out = np.zeros((n_pts + 1, 29))
    lib().orc_sample(C.byref(model), num_seg, _p(xs), _p(us), C.c_double(T), n_pts, _p(out)); return out


def collocation_defects(num_seg, xs, us, T):
    """defects at all four local nodes of every segment: [num_seg][4][14]"""
    xs, us = f64(xs), f64(us); out = np.zeros((num_seg, 4, 14))
    lib().orc_collocation_defects(int(num_seg), _p(xs), _p(us), C.c_double(T), _p(out)); return out


def debug_qp(cfg, x0, xf, xs, us, T, lam=None, model=None):
    model = model or default_model(); N = 3 * cfg.num_seg + 1
    n, m = 21 * N + 1, 14 * (N - 1) + 8 * N
    x0, xf, xs, us = f64(x0), f64(xf), f64(xs), f64(us)
    lam = f64(lam) if lam is not None else None
    p, y = np.zeros(n), np.zeros(m + n)
    lib().orc_debug_qp.restype = C.c_int
    it = lib().orc_debug_qp(C.byref(model), C.byref(cfg), _p(x0), _p(xf), _p(xs), _p(us), C.c_double(T),
                            _p(lam), _p(p), _p(y))
    return p, y, it


def mpc_point(num_seg, xs, us, T, time, model=None):
    model = model or default_model(); xs, us = f64(xs), f64(us)
    out = np.zeros(28)
    lib().orc_mpc_point(C.byref(model), num_seg, _p(xs), _p(us), C.c_double(T), C.c_double(time), _p(out)); return out


def rh_start_guess(cfg, x, xf):
    """the receding-horizon driver's start guess (mpcmp_rh_run: first solve, and the restart of an instance whose previous solve is no guess): the
    jerk-limited time-synchronised trajectory from the current state, velocity / acceleration limits of the configuration, jerk margin 0.1
    (examples/offline_trajectory.cpp:9)"""
    return warm_start_jerk(cfg.num_seg, np.array(cfg.ubx[7:14]), np.array(cfg.ubu[:]), 0.1 * default_limits()["jmax"], x, xf)


def rh_advance(cfg, xs, us, T, status, dt, xf, x_now):
    """state advance + arrival rule of the receding-horizon driver (orc_rh_advance): returns (new state [14], retired)"""
    xs, us, xf = f64(xs), f64(us), f64(xf); x = f64(x_now).copy()
    lib().orc_rh_advance.restype = C.c_int
    r = lib().orc_rh_advance(C.byref(cfg), _p(xs), _p(us), C.c_double(T), int(status), C.c_double(dt), _p(xf), _p(x))
    return x, bool(r)


def traj_stats(num_seg, xs, us, T, xf, n_pts=200, model=None):
    model = model or default_model(); xs, us, xf = f64(xs), f64(us), f64(xf)
    out = np.zeros(74)
    lib().orc_traj_stats(C.byref(model), num_seg, _p(xs), _p(us), C.c_double(T), _p(xf), int(n_pts), _p(out)); return out
