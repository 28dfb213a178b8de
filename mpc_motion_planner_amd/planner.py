"""Host-side mirror of the reference's planner façade for the batched HIP path.

`Solver` is the raw context (one per GPU); `BatchMotionPlanner` keeps the member names and argument meaning
of `MotionPlanner` (mpc_solver/motionPlanner.hpp:16-176, motionPlanner.cpp) with a leading batch dimension.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import check, dp, f64, lib


class Solver:
    """mpcmp_ctx wrapper. Host-buffer calls take numpy arrays; *_device calls take raw device pointers."""

    def __init__(self, cfg, max_batch, device=0, model=None, models=None):
        """models: a ctypes array of `narm` Models (multi-arm robot, mpcmp_create_multi); model: one Model; neither: the Panda"""
        self.cfg = cfg
        self.narm = len(models) if models is not None else 1
        self.nx, self.nu = 14 * self.narm, 7 * self.narm
        self.N = 3 * cfg.num_seg + 1
        self.n = 21 * self.N * self.narm + 1
        self.m = (14 * (self.N - 1) + 8 * self.N) * self.narm
        self.max_batch = int(max_batch)
        self._ctx = C.c_void_p()
        if models is not None:
            rc = lib().mpcmp_create_multi(C.byref(cfg), models, self.narm, int(device), int(max_batch), C.byref(self._ctx))
        else:
            rc = lib().mpcmp_create(C.byref(cfg), C.byref(model) if model is not None else None, int(device),
                                    int(max_batch), C.byref(self._ctx))
        check(rc)

    def close(self):
        if self._ctx:
            lib().mpcmp_destroy(self._ctx); self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_config(self, cfg):
        check(lib().mpcmp_set_config(self._ctx, C.byref(cfg)), self._ctx); self.cfg = cfg

    def _states(self, x0, xf):
        """(x0, xf) as contiguous [B][14] arrays; a shape mismatch raises instead of letting the C side read out of bounds"""
        x0, xf = f64(x0), f64(xf)
        if x0.ndim != 2 or x0.shape[1] != self.nx or xf.shape != x0.shape:
            raise ValueError("x0 and xf must both be [B][%d] arrays, got %s and %s" % (self.nx, x0.shape, xf.shape))
        return x0, xf

    def _traj(self, sx, su, sT):
        sx, su, sT = f64(sx), f64(su), f64(sT).reshape(-1)
        B = sT.shape[0]
        if sx.shape != (B, self.N, self.nx) or su.shape != (B, self.N, self.nu):
            raise ValueError("trajectory arrays must be [B][%d][%d], [B][%d][%d], [B]; got %s, %s, %s"
                             % (self.N, self.nx, self.N, self.nu, sx.shape, su.shape, sT.shape))
        return sx, su, sT

    # -- hot path, host buffers
    def solve(self, x0, xf, warm=None):
        x0, xf = self._states(x0, xf); B = x0.shape[0]
        wx = wu = wT = None
        if warm is not None:
            wx, wu, wT = self._traj(*warm)
            if wT.shape[0] != B:
                raise ValueError("warm start is for %d problems, states for %d" % (wT.shape[0], B))
        sx, su, sT = np.zeros((B, self.N, self.nx)), np.zeros((B, self.N, self.nu)), np.zeros(B)
        info = np.zeros(B, dtype=capi.INFO_DTYPE)
        check(lib().mpcmp_solve_batch(self._ctx, B, dp(x0), dp(xf), dp(wx), dp(wu), dp(wT), dp(sx), dp(su), dp(sT),
                                      info.ctypes.data_as(C.c_void_p)), self._ctx)
        return sx, su, sT, info

    # -- hot path, device-resident buffers (raw pointers, e.g. torch tensors' data_ptr())
    def solve_device(self, B, x0, xf, sol_x, sol_u, sol_T, info=0, warm=(0, 0, 0), stream=0):
        vp = C.c_void_p
        check(lib().mpcmp_solve_batch_device(self._ctx, int(B), vp(x0), vp(xf), vp(warm[0] or None), vp(warm[1] or None),
                                             vp(warm[2] or None), vp(sol_x), vp(sol_u), vp(sol_T), vp(info or None),
                                             vp(stream or None)), self._ctx)

    def warm_start(self, x0, xf):
        x0, xf = self._states(x0, xf); B = x0.shape[0]
        wx, wu, wT = np.zeros((B, self.N, 14)), np.zeros((B, self.N, 7)), np.zeros(B)
        check(lib().mpcmp_warm_start_batch(self._ctx, B, dp(x0), dp(xf), dp(wx), dp(wu), dp(wT)), self._ctx)
        return wx, wu, wT

    def _acc(self, a, B):
        return None if a is None else f64(np.broadcast_to(np.asarray(a, dtype=np.float64), (B, 7)))

    @staticmethod
    def _lim(v):
        return None if v is None else f64(v).reshape(7)

    def warm_start_jerk(self, x0, xf, jmax, acc0=None, accT=None, vmax=None, amax=None):
        """Jerk-limited, time-synchronised warm start (stands in for Ruckig): (warm_x [B][N][14], warm_u [B][N][7], warm_T [B]).
        acc0 / accT: boundary accelerations [B][7] (None = zero), as set_current_state / set_target_state forward them (motionPlanner.cpp:36-38,50-52).
        vmax / amax [7]: the generator's velocity / acceleration limits (None = the context's margin-applied bounds): ruckig's input.max_velocity / max_acceleration."""
        (x0, xf), jmax = self._states(x0, xf), f64(jmax).reshape(7); B = x0.shape[0]
        wx, wu, wT = np.zeros((B, self.N, self.nx)), np.zeros((B, self.N, self.nu)), np.zeros(B)
        a0, aT = self._acc(acc0, B), self._acc(accT, B)
        check(lib().mpcmp_warm_start_jerk_lim_batch(self._ctx, B, dp(x0), dp(xf), dp(a0), dp(aT), dp(self._lim(vmax)), dp(self._lim(amax)), dp(jmax),
                                                    dp(wx), dp(wu), dp(wT)), self._ctx)
        return wx, wu, wT

    def warm_start_jerk_device(self, B, x0, xf, jmax, warm_x, warm_u, warm_T, stream=0):
        """Device-pointer form: writes the warm start where solve_device(..., warm=(warm_x, warm_u, warm_T)) reads it."""
        vp = C.c_void_p
        jmax = f64(jmax)
        check(lib().mpcmp_warm_start_jerk_batch_device(self._ctx, int(B), vp(x0), vp(xf), dp(jmax), vp(warm_x), vp(warm_u), vp(warm_T),
                                                       vp(stream or None)), self._ctx)

    def jerk_trajectory(self, x0, xf, jmax, n_pts=200, acc0=None, accT=None, vmax=None, amax=None):
        """The same trajectory sampled uniformly: out [B][n_pts+1][22] = t, q, qd, qdd, and the durations [B]."""
        (x0, xf), jmax = self._states(x0, xf), f64(jmax).reshape(7); B = x0.shape[0]
        out, T = np.zeros((B, n_pts + 1, 22)), np.zeros(B)
        a0, aT = self._acc(acc0, B), self._acc(accT, B)
        check(lib().mpcmp_jerk_trajectory_lim_batch(self._ctx, B, dp(x0), dp(xf), dp(a0), dp(aT), dp(self._lim(vmax)), dp(self._lim(amax)), dp(jmax),
                                                    int(n_pts), dp(out), dp(T)), self._ctx)
        return out, T

    def jerk_point(self, x0, xf, jmax, time, acc0=None, accT=None, vmax=None, amax=None):
        """MotionPlanner::get_RK_point: [B][28] = q, qd, qdd, tau of the jerk-limited trajectory at min(time, duration), durations [B]."""
        (x0, xf), jmax = self._states(x0, xf), f64(jmax).reshape(7); B = x0.shape[0]
        time = f64(np.broadcast_to(np.asarray(time, dtype=np.float64), (B,)))
        out, T = np.zeros((B, 28)), np.zeros(B)
        a0, aT = self._acc(acc0, B), self._acc(accT, B)
        check(lib().mpcmp_jerk_point_lim_batch(self._ctx, B, dp(x0), dp(xf), dp(a0), dp(aT), dp(self._lim(vmax)), dp(self._lim(amax)), dp(jmax),
                                               dp(time), dp(out), dp(T)), self._ctx)
        return out, T

    def mpc_point(self, sx, su, sT, time):
        """MotionPlanner::get_MPC_point (with its clamp: time >= T -> normalised time T): [B][28] = q, qd, qdd, tau."""
        sx, su, sT = self._traj(sx, su, sT); B = sT.shape[0]
        time = f64(np.broadcast_to(np.asarray(time, dtype=np.float64), (B,)))
        out = np.zeros((B, 28))
        check(lib().mpcmp_mpc_point_batch(self._ctx, B, dp(sx), dp(su), dp(sT), dp(time), dp(out)), self._ctx)
        return out

    def rnea(self, q, qd, qdd):
        q, qd, qdd = f64(q), f64(qd), f64(qdd); tau = np.zeros_like(q)
        check(lib().mpcmp_rnea_batch(self._ctx, q.shape[0], dp(q), dp(qd), dp(qdd), dp(tau)), self._ctx)
        return tau

    def eval_constraints(self, x, u):
        x, u = f64(x), f64(u); nn = x.shape[0]
        g, G = np.zeros((nn, 8)), np.zeros((nn, 8, 22))
        check(lib().mpcmp_eval_constraints_batch(self._ctx, nn, dp(x), dp(u), dp(g), dp(G)), self._ctx)
        return g, G

    def qp(self, x0, xf, xs, us, T):
        x0, xf = self._states(x0, xf); B = x0.shape[0]
        xs, us, T = self._traj(xs, us, T)
        if T.shape[0] != B:
            raise ValueError("linearisation point is for %d problems, states for %d" % (T.shape[0], B))
        p, y = np.zeros((B, self.n)), np.zeros((B, self.m + self.n)); it = np.zeros(B, dtype=np.int32)
        check(lib().mpcmp_qp_batch(self._ctx, B, dp(x0), dp(xf), dp(xs), dp(us), dp(T), dp(p), dp(y),
                                   it.ctypes.data_as(C.c_void_p)), self._ctx)
        return p, y, it

    def sample(self, sx, su, sT, n_pts=200):
        sx, su, sT = self._traj(sx, su, sT); B = sx.shape[0]
        out = np.zeros((B, n_pts + 1, 29))
        check(lib().mpcmp_sample_batch(self._ctx, B, dp(sx), dp(su), dp(sT), int(n_pts), dp(out)), self._ctx)
        return out

    def sample_device(self, B, sol_x, sol_u, sol_T, n_pts, out, stream=0):
        vp = C.c_void_p
        check(lib().mpcmp_sample_batch_device(self._ctx, int(B), vp(sol_x), vp(sol_u), vp(sol_T), int(n_pts), vp(out),
                                              vp(stream or None)), self._ctx)

    def traj_stats(self, sx, su, sT, xf, n_pts=200):
        (sx, su, sT), xf = self._traj(sx, su, sT), f64(xf); B = sx.shape[0]
        if xf.shape != (B, 14):
            raise ValueError("xf must be [B][14]")
        out = np.zeros((B, 74))
        check(lib().mpcmp_traj_stats_batch(self._ctx, B, dp(sx), dp(su), dp(sT), dp(xf), int(n_pts), dp(out)), self._ctx)
        return out

    def reset_multipliers(self):
        """zero the multipliers every problem slot carries to its next solve (mpcmp_config.carry_multipliers)"""
        check(lib().mpcmp_reset_multipliers(self._ctx), self._ctx)

    # -- receding horizon (BASELINE config #5)
    def rh_init(self, x0, xf):
        x0, xf = self._states(x0, xf)
        self._rh_B = x0.shape[0]
        check(lib().mpcmp_rh_init(self._ctx, self._rh_B, dp(x0), dp(xf)), self._ctx)

    def rh_run(self, steps, dt, use_graph=True):
        check(lib().mpcmp_rh_run(self._ctx, int(steps), C.c_double(dt), int(bool(use_graph))), self._ctx)

    def rh_get(self):
        B = self._rh_B
        x0 = np.zeros((B, self.nx)); sx, su, sT = np.zeros((B, self.N, self.nx)), np.zeros((B, self.N, self.nu)), np.zeros(B)
        info = np.zeros(B, dtype=capi.INFO_DTYPE)
        check(lib().mpcmp_rh_get(self._ctx, dp(x0), dp(sx), dp(su), dp(sT), info.ctypes.data_as(C.c_void_p)), self._ctx)
        return x0, sx, su, sT, info

    def rh_stats(self):
        """(re-solves executed on live instances, instances retired) since rh_init (mpcmp_rh_stats)"""
        done, arr = C.c_longlong(0), C.c_longlong(0)
        check(lib().mpcmp_rh_stats(self._ctx, C.byref(done), C.byref(arr)), self._ctx)
        return int(done.value), int(arr.value)

    def debug_fetch(self, which, count):
        """diagnostics: first `count` doubles of a workspace array (0 z, 1 lambda, 2 c_eq, 3 g, 4 p, 5 y), device layout"""
        out = np.zeros(int(count))
        check(lib().mpcmp_debug_fetch(self._ctx, int(which), dp(out), C.c_long(int(count))), self._ctx)
        return out

    def kernel_timing(self, reset=False):
        name = C.c_char_p(); ms = C.c_double(); nl = C.c_int()
        check(lib().mpcmp_kernel_timing(self._ctx, int(reset), C.byref(name), C.byref(ms), C.byref(nl)), self._ctx)
        return name.value.decode(), ms.value, nl.value


class BatchMotionPlanner:
    """Batched `MotionPlanner`: same member names / argument meaning, arrays carry a leading batch axis.

    Reference call sequence (examples/benchmark.cpp:6-53): MotionPlanner(urdf); set_constraint_margins(...);
    per problem set_current_state / set_target_state; solve_trajectory(true); get_MPC_trajectory<200>.
    """

    eps = 1e-2  # motionPlanner.hpp:44

    def __init__(self, urdf_path=None, max_batch=1024, num_seg=4, sqp_iters=20, device=0):
        self.model = capi.model_from_urdf(urdf_path) if urdf_path else capi.default_model()
        self.limits = capi.default_limits()
        self.cfg = capi.default_config(num_seg, sqp_iters)         # motionPlanner.cpp:15-24
        self._solver = Solver(self.cfg, max_batch, device, self.model)
        mid = 0.5 * (self.limits["qmin"] + self.limits["qmax"])    # motionPlanner.cpp:5-8
        self.current_state = np.concatenate([mid, np.zeros(7)])[None, :]
        self.target_state = self.current_state.copy()
        self.margin_position_ = self.margin_velocity_ = self.margin_acceleration_ = 1.0
        self.margin_torque_ = self.margin_jerk_ = 1.0
        self._warm = None
        self._sol = None
        self._rk = None

    # motionPlanner.cpp:56-90
    def set_constraint_margins(self, margin_position, margin_velocity, margin_acceleration, margin_torque, margin_jerk):
        self.margin_position_, self.margin_velocity_ = margin_position, margin_velocity
        self.margin_acceleration_, self.margin_torque_, self.margin_jerk_ = margin_acceleration, margin_torque, margin_jerk
        check(lib().mpcmp_set_margins(C.byref(self.cfg), C.c_double(margin_position), C.c_double(margin_velocity),
                                      C.c_double(margin_acceleration), C.c_double(margin_torque)))
        self._solver.set_config(self.cfg)

    # motionPlanner.cpp:92-100
    def set_min_height(self, min_height):
        check(lib().mpcmp_set_min_height(C.byref(self.cfg), C.c_double(min_height)))
        self._solver.set_config(self.cfg)

    # motionPlanner.cpp:27-39 / 41-54
    def set_target_state(self, target_position, target_velocity):
        self.target_state = np.concatenate([f64(target_position), f64(target_velocity)], axis=-1).reshape(-1, 14)

    def set_current_state(self, current_position, current_velocity):
        self.current_state = np.concatenate([f64(current_position), f64(current_velocity)], axis=-1).reshape(-1, 14)

    # motionPlanner.cpp:116-144
    def check_state_in_bounds(self, position, velocity, acceleration=None):
        position, velocity = f64(position), f64(velocity)
        L = self.limits
        s = (1 - self.margin_position_) * (L["qmax"] - L["qmin"]) / 2
        pc = np.any(position > L["qmax"] - s, axis=-1) | np.any(position < L["qmin"] + s, axis=-1)
        vc = np.any(np.abs(velocity) > self.margin_velocity_ * L["vmax"], axis=-1)
        flag = np.where(pc & ~vc, 1, 0) + np.where(~pc & vc, 2, 0) + np.where(pc & vc, 3, 0)
        if acceleration is not None:
            flag = flag + 10 * np.any(np.abs(f64(acceleration)) > self.margin_acceleration_ * L["amax"], axis=-1)
        return flag

    # motionPlanner.hpp:145-172 (generic warm start from a regularly time-spaced trajectory [B, nPoint, 7])
    def warm_start(self, final_time, position_trajectory, velocity_trajectory, acceleration_trajectory):
        q, v, a = f64(position_trajectory), f64(velocity_trajectory), f64(acceleration_trajectory)
        nP = q.shape[1]
        idx = np.floor(capi.time_nodes(self.cfg.num_seg) * (nP - 1) + 0.5).astype(int)     # std::round, motionPlanner.hpp:157
        self._warm = (np.concatenate([q[:, idx], v[:, idx]], axis=-1), a[:, idx], f64(final_time).reshape(-1))

    # motionPlanner.cpp:177-208
    def solve_trajectory(self, use_builtin_warm_start=True):
        B = max(self.current_state.shape[0], self.target_state.shape[0])
        x0 = np.broadcast_to(self.current_state, (B, 14)); xf = np.broadcast_to(self.target_state, (B, 14))
        if use_builtin_warm_start:      # what the reference gets from Ruckig (warm_start_RK, motionPlanner.cpp:146-175)
            self._rk = (np.array(x0), np.array(xf))
            warm = self._solver.warm_start_jerk(x0, xf, self.margin_jerk_ * self.limits["jmax"])     # motionPlanner.cpp:86-88
        else:
            warm = self._warm
        sx, su, sT, info = self._solver.solve(x0, xf, warm)
        self._sol = (sx, su, sT)
        # re-guess with exact end states (motionPlanner.cpp:199-207)
        gx = sx.copy(); gx[:, 0] = x0; gx[:, -1] = xf
        self._warm = (gx, su.copy(), sT.copy())
        self.info = info
        return info

    def solution(self):
        return self._sol

    # motionPlanner.hpp:118-128 / 130-142
    def get_MPC_point(self, time):
        o = self._solver.mpc_point(*self._sol, time=time)
        return o[:, :7], o[:, 7:14], o[:, 14:21], o[:, 21:]

    def get_RK_point(self, time):
        o, _ = self._solver.jerk_point(self._rk[0], self._rk[1], self.margin_jerk_ * self.limits["jmax"], time)
        return o[:, :7], o[:, 7:14], o[:, 14:21], o[:, 21:]

    # motionPlanner.hpp:73-96
    def get_ruckig_trajectory(self, n_pts=200):
        out, _ = self._solver.jerk_trajectory(self._rk[0], self._rk[1], self.margin_jerk_ * self.limits["jmax"], n_pts)
        q, v, a = out[..., 1:8], out[..., 8:15], out[..., 15:22]
        tau = self._solver.rnea(q.reshape(-1, 7), v.reshape(-1, 7), a.reshape(-1, 7)).reshape(q.shape)
        return out[..., 0], q, v, a, tau

    # motionPlanner.hpp:99-116
    def get_MPC_trajectory(self, n_pts=200):
        out = self._solver.sample(*self._sol, n_pts=n_pts)
        return out[..., 0], out[..., 1:8], out[..., 8:15], out[..., 15:22], out[..., 22:29]
