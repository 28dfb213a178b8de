"""ctypes declarations for include/mpcmp.h."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmpcmp.so")


class MpcmpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mpcmp error %d: %s" % (code, msg))
        self.code = code


class Model(C.Structure):
    _fields_ = [("R0", C.c_double * 9 * 7), ("p", C.c_double * 3 * 7), ("mass", C.c_double * 7),
                ("com", C.c_double * 3 * 7), ("I", C.c_double * 9 * 7), ("tool", C.c_double * 3),
                ("link8", C.c_double * 3), ("gravity", C.c_double * 3)]


class Config(C.Structure):
    _fields_ = [("num_seg", C.c_int), ("sqp_iters", C.c_int), ("qp_iters", C.c_int), ("ls_iters", C.c_int),
                ("check_every", C.c_int), ("quirk_dtau_dT", C.c_int),
                ("eps_abs", C.c_double), ("eps_rel", C.c_double),
                ("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double), ("rho_eq_scale", C.c_double),
                ("ls_eta", C.c_double), ("ls_tau", C.c_double), ("hess_reg", C.c_double), ("eps_target", C.c_double),
                ("lbx", C.c_double * 14), ("ubx", C.c_double * 14), ("lbu", C.c_double * 7), ("ubu", C.c_double * 7),
                ("lbg", C.c_double * 8), ("ubg", C.c_double * 8), ("lbT", C.c_double), ("ubT", C.c_double),
                ("qp_warm_start", C.c_int), ("carry_multipliers", C.c_int)]


class Info(C.Structure):
    _fields_ = [("T", C.c_double), ("viol_l1", C.c_double), ("defect_inf", C.c_double),
                ("path_viol_inf", C.c_double), ("term_err_inf", C.c_double), ("last_alpha", C.c_double),
                ("qp_iters_total", C.c_int), ("sqp_iters", C.c_int), ("status", C.c_int), ("qp_capped", C.c_int)]


# status bits of mpcmp_info.status (include/mpcmp.h)
STATUS_NAN, STATUS_NOT_PD, STATUS_XCH_DEAD, STATUS_QP_CAPPED, STATUS_OUTSIDE_TOL, STATUS_T_OUT_OF_BOX, STATUS_ARRIVED = 1, 2, 4, 8, 16, 32, 64
STATUS_HARD = STATUS_NAN | STATUS_NOT_PD | STATUS_XCH_DEAD        # the solve itself failed (the other bits grade the returned iterate)

INFO_DTYPE = np.dtype([("T", "f8"), ("viol_l1", "f8"), ("defect_inf", "f8"), ("path_viol_inf", "f8"),
                       ("term_err_inf", "f8"), ("last_alpha", "f8"), ("qp_iters_total", "i4"),
                       ("sqp_iters", "i4"), ("status", "i4"), ("qp_capped", "i4")])

# every symbol include/mpcmp.h declares (the CPU test suite checks the library exports all of them)
SYMBOLS = ["mpcmp_default_model", "mpcmp_model_from_urdf", "mpcmp_models_from_urdf", "mpcmp_default_limits", "mpcmp_default_config",
           "mpcmp_set_margins", "mpcmp_set_min_height", "mpcmp_num_nodes", "mpcmp_time_nodes", "mpcmp_version",
           "mpcmp_create", "mpcmp_create_multi", "mpcmp_destroy", "mpcmp_set_config", "mpcmp_last_error", "mpcmp_solve_batch",
           "mpcmp_solve_batch_device", "mpcmp_warm_start_batch", "mpcmp_rnea_batch",
           "mpcmp_eval_constraints_batch", "mpcmp_qp_batch", "mpcmp_sample_batch", "mpcmp_sample_batch_device",
           "mpcmp_kernel_timing", "mpcmp_debug_stamps", "mpcmp_rh_init", "mpcmp_rh_run", "mpcmp_rh_get", "mpcmp_traj_stats_batch",
           "mpcmp_tool_jacobian", "mpcmp_forward_velocities", "mpcmp_inverse_velocities", "mpcmp_inverse_kinematics",
           "mpcmp_warm_start_jerk_batch", "mpcmp_warm_start_jerk_batch_device", "mpcmp_jerk_trajectory_batch",
           "mpcmp_jerk_point_batch", "mpcmp_mpc_point_batch", "mpcmp_debug_fetch",
           "mpcmp_warm_start_jerk_acc_batch", "mpcmp_warm_start_jerk_acc_batch_device", "mpcmp_jerk_trajectory_acc_batch", "mpcmp_jerk_point_acc_batch",
           "mpcmp_warm_start_jerk_lim_batch", "mpcmp_jerk_trajectory_lim_batch", "mpcmp_jerk_point_lim_batch", "mpcmp_reset_multipliers", "mpcmp_rh_stats"]


def library_path():
    return _SO


def build_library(force=False):
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    deps = [os.path.join(src, f) for f in ("mpcmp.hip", "solver_kernels.hpp", "qp_kernel_v2.hpp", "qp_kernel_v3.hpp", "qp_kernel_v5.hpp", "structure3.hpp", "rbd_device.hpp", "structure.hpp", "multi_kernels.hpp",
                                           "kinematics_host.hpp", "jerk_device.hpp")]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "mpcmp.h"))
    if force or not os.path.exists(_SO) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps):
        subprocess.check_call(["make", "-C", src, "-B"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def _torch_runtime_first():
    """torch bundles its own libamdhip64.so; libmpcmp.so links the system one (/opt/rocm), same soname.  Whichever is mapped first serves
    both: with torch's first everything works (the order bench.py and the tests have); with the system runtime first, a LATER `import torch`
    maps a second HIP runtime and reports "No HIP GPUs are available".  So: if torch is installed and not yet imported, import it before the
    library is loaded (MPCMP_NO_TORCH_PRELOAD=1 skips this, e.g. for torch-free deployments that want the shorter start-up)."""
    import sys
    if "torch" in sys.modules or os.environ.get("MPCMP_NO_TORCH_PRELOAD"):
        return
    import importlib.util
    if importlib.util.find_spec("torch") is not None:
        try:
            import torch  # noqa: F401
        except Exception as e:      # (say so — the system HIP runtime is mapped first now, and a later `import torch` would find no GPU)
            import warnings
            warnings.warn("mpc_motion_planner_amd: importing torch before loading libmpcmp.so failed (%r); a later `import torch` in this process "
                          "may report no GPU.  Set MPCMP_NO_TORCH_PRELOAD=1 to skip the preload." % (e,))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise MpcmpError(-2, "libmpcmp.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                                 "there is no CPU fallback" % _SO)
        _torch_runtime_first()
        L = C.CDLL(_SO)
        L.mpcmp_version.restype = C.c_char_p
        L.mpcmp_last_error.restype = C.c_char_p
        L.mpcmp_last_error.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def check(rc, ctx=None):
    if rc != 0:
        msg = lib().mpcmp_last_error(ctx)
        raise MpcmpError(rc, (msg or b"").decode())


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def default_model():
    m = Model(); check(lib().mpcmp_default_model(C.byref(m))); return m


def arm_models(bases, model=None):
    """ctypes array of Models for a multi-arm robot: `model` (default: the compiled-in Panda) mounted at every base =
    (yaw about world z, [x, y, z]); the base placement is folded into the first joint placement."""
    arr = (Model * len(bases))()
    for a, (yaw, xyz) in enumerate(bases):
        m = Model.from_buffer_copy(model if model is not None else default_model())
        c, s = np.cos(yaw), np.sin(yaw)
        Rz = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        R0 = Rz @ np.array(m.R0[0][:]).reshape(3, 3); p0 = np.asarray(xyz, dtype=np.float64) + Rz @ np.array(m.p[0][:])
        for i in range(9):
            m.R0[0][i] = R0.ravel()[i]
        for i in range(3):
            m.p[0][i] = p0[i]
        arr[a] = m
    return arr


DUAL_BASES = ((0.0, (0.0, 0.0, 0.0)), (np.pi, (1.0, 0.0, 0.0)))      # two Pandas facing each other, 1 m apart (configs[3] stand-in)


def model_from_urdf(path):
    m = Model(); check(lib().mpcmp_model_from_urdf(path.encode(), C.byref(m))); return m


def models_from_urdf(path, max_chains=8):
    """Every serial 7-joint chain of the URDF (file order) as a ctypes array of Models, ready for Solver(models=...):
    fixed joints, base placements, rotated inertial frames and joint axes other than +z are folded in (mpcmp.h)."""
    arr = (Model * max_chains)(); n = C.c_int(0)
    check(lib().mpcmp_models_from_urdf(path.encode(), max_chains, arr, C.byref(n)))
    out = (Model * n.value)()
    for i in range(n.value):
        out[i] = arr[i]
    return out


# ---- scenario helpers of the robot wrapper (host side; robot_utils/pandaWrapper.cpp:14-107) ----
def tool_jacobian(model, q):
    """World-aligned 6x7 tool-frame Jacobian [linear; angular], tool position and rotation."""
    q = f64(q); J = np.zeros((6, 7)); p = np.zeros(3); R = np.zeros((3, 3))
    check(lib().mpcmp_tool_jacobian(C.byref(model), dp(q), dp(J), dp(p), dp(R)))
    return J, p, R


def forward_velocities(model, q, qd):
    """PandaWrapper::forward_velocities: task velocity [linear(3); angular(3)] of the tool frame."""
    q, qd = f64(q), f64(qd); out = np.zeros(6)
    check(lib().mpcmp_forward_velocities(C.byref(model), dp(q), dp(qd), dp(out)))
    return out


def inverse_velocities(model, q, linear_velocity, angular_velocity):
    """PandaWrapper::inverse_velocities: damped (1e-5) pseudo-inverse map from task to joint velocity."""
    q, lin, ang = f64(q), f64(linear_velocity), f64(angular_velocity); out = np.zeros(7)
    check(lib().mpcmp_inverse_velocities(C.byref(model), dp(q), dp(lin), dp(ang), dp(out)))
    return out


def inverse_kinematic(model, orientation, position, q_init=None):
    """PandaWrapper::inverse_kinematic: returns (q, converged, iterations); the start configuration is explicit."""
    R, p = f64(orientation), f64(position); q = np.zeros(7); it = C.c_int(0)
    qi = f64(q_init) if q_init is not None else None
    rc = lib().mpcmp_inverse_kinematics(C.byref(model), dp(R), dp(p), dp(qi), dp(q), C.byref(it))
    if rc not in (0, 1):
        check(rc)
    return q, rc == 0, it.value


def default_limits():
    out = [np.zeros(7) for _ in range(6)]
    check(lib().mpcmp_default_limits(*[dp(o) for o in out]))
    return dict(zip(["qmin", "qmax", "vmax", "amax", "jmax", "taumax"], out))


def default_config(num_seg=4, sqp_iters=20, margins=None, **kw):
    c = Config(); check(lib().mpcmp_default_config(C.byref(c), int(num_seg), int(sqp_iters)))
    if margins is not None:
        check(lib().mpcmp_set_margins(C.byref(c), *[C.c_double(x) for x in margins[:4]]))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def num_nodes(num_seg):
    return lib().mpcmp_num_nodes(int(num_seg))


def time_nodes(num_seg):
    t = np.zeros(3 * num_seg + 1); check(lib().mpcmp_time_nodes(int(num_seg), dp(t))); return t
