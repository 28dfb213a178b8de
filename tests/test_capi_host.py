"""CPU-side checks of the product library: it loads, exports every symbol include/mpcmp.h declares, its host
logic (config, model, URDF reader, structure) agrees with the oracle, and compute calls fail loudly without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_py as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def M():
    import mpc_motion_planner_amd as M
    M.build_library()
    return M


def test_library_exports_every_declared_symbol(M):
    hdr = open(os.path.join(ROOT, "include", "mpcmp.h")).read()
    declared = set(re.findall(r"\b(mpcmp_[a-z_0-9]+)\s*\(", hdr))
    from mpc_motion_planner_amd import capi
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    lib = M.lib()
    for s in declared:
        assert hasattr(lib, s), s
    assert M.lib().mpcmp_version().startswith(b"mpcmp")


def test_struct_layouts_match_header(M, tmp_path):
    assert C.sizeof(M.Info) == 64 and M.INFO_DTYPE.itemsize == 64
    assert C.sizeof(M.Config) == C.sizeof(o.Config) and C.sizeof(M.Model) == C.sizeof(o.Model)
    # the header itself, through a C compiler: sizes and the offsets of the fields appended last
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mpcmp.h"\nint main(void) { printf("%zu %zu %zu %zu %zu\\n", sizeof(mpcmp_config), '
                   'sizeof(mpcmp_model), sizeof(mpcmp_info), offsetof(mpcmp_config, qp_warm_start), offsetof(mpcmp_config, lbT)); return 0; }\n')
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    sz = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert sz == [C.sizeof(M.Config), C.sizeof(M.Model), C.sizeof(M.Info), M.Config.qp_warm_start.offset, M.Config.lbT.offset], sz


def test_config_and_limits_match_oracle(M):
    for margins in [(1, 1, 1, 1), (0.9, 0.9, 0.5, 0.9), (0.8, 0.8, 0.6, 0.9)]:
        a = M.default_config(4, 20, margins=margins); b = o.default_config(4, 20, margins=margins)
        assert bytes(a) == bytes(b)
    la, lb = M.default_limits(), o.default_limits()
    for k in la:
        assert np.array_equal(la[k], lb[k])
    # set_constraint_margins arithmetic (motionPlanner.cpp:66-75)
    c = M.default_config(4, 2, margins=(0.9, 0.9, 0.5, 0.9))
    s = 0.1 * (2.8973 + 2.8973) / 2
    assert abs(c.lbx[0] - (-2.8973 + s)) < 1e-15 and abs(c.ubx[7] - 0.9 * 2.175) < 1e-15
    assert abs(c.ubu[1] - 0.5 * 7.5) < 1e-15 and abs(c.ubg[4] - 0.9 * 12) < 1e-15 and c.lbg[7] == 0.05 and np.isinf(c.ubg[7])


def test_model_matches_oracle_and_urdf(M):
    a, b = M.default_model(), o.default_model()
    for f in ["R0", "p", "mass", "com", "I", "tool", "link8", "gravity"]:
        assert np.array_equal(np.array(getattr(a, f)), np.array(getattr(b, f))), f
    urdf = "/root/reference/robot_utils/panda-model/panda_arm.urdf"
    if not os.path.exists(urdf):
        pytest.skip("reference URDF not present on this machine")
    u = M.model_from_urdf(urdf)
    for f in ["R0", "p", "mass", "com", "I", "tool", "link8", "gravity"]:
        assert np.array_equal(np.array(getattr(a, f)), np.array(getattr(u, f))), f
    assert abs(u.mass[6] - 1.735522) < 1e-12      # link7 + link8 (m=0) + tool (m=1) lumped


def test_urdf_reader_rejects_bad_input(M, tmp_path):
    with pytest.raises(M.MpcmpError):
        M.model_from_urdf(str(tmp_path / "missing.urdf"))
    p = tmp_path / "two_joints.urdf"
    p.write_text('<robot name="x"><link name="a"/><link name="b"/><joint name="j" type="revolute"><parent link="a"/>'
                 '<child link="b"/><axis xyz="0 0 1"/></joint></robot>')
    with pytest.raises(M.MpcmpError):
        M.model_from_urdf(str(p))


def test_time_nodes_and_dims(M):
    for ns in (1, 2, 4, 6, 8):
        assert M.num_nodes(ns) == 3 * ns + 1
        assert np.array_equal(M.time_nodes(ns), o.time_nodes(ns))


def test_scenarios_are_seeded_and_feasible(M):
    from mpc_motion_planner_amd import scenarios
    x0, xf = scenarios.make_batch(64)
    x0b, xfb = scenarios.make_batch(64)
    assert np.array_equal(x0, x0b) and np.array_equal(xf, xfb)
    x0c, _ = scenarios.make_batch(32, stream_offset=32)
    assert np.array_equal(x0c, x0[32:])            # shard r of the global batch == its slice (multi-GPU sharding)
    cfg = M.default_config(4, 2, margins=(0.9, 0.9, 0.5, 0.9))
    for x in (x0, xf):
        assert np.all(x[:, :7] >= np.array(cfg.lbx[:7]) - 1e-12) and np.all(x[:, :7] <= np.array(cfg.ubx[:7]) + 1e-12)
        assert np.all(np.abs(x[:, 7:]) <= np.array(cfg.ubx[7:]) + 1e-12)
        # rejection rule of sample_random_state: joint-7 origin above min_height (motionPlanner.cpp:111)
        assert np.all(scenarios.joint7_height(M.default_model(), x[:, :7]) >= 0.05)
        for i in range(8):
            assert abs(scenarios.joint7_height(M.default_model(), x[i:i + 1, :7])[0] - o.fk(x[i, :7])[0][2]) < 1e-12


def test_no_cpu_fallback(M):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = M.default_config(4, 2)
    with pytest.raises(M.MpcmpError) as e:
        M.Solver(cfg, 4)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_bad_config_rejected_before_touching_the_gpu(M):
    ctx = C.c_void_p()
    bad = M.default_config(3, 2)
    assert M.lib().mpcmp_create(C.byref(bad), None, 0, 4, C.byref(ctx)) == -1
    good = M.default_config(4, 2)
    assert M.lib().mpcmp_create(C.byref(good), None, 0, 0, C.byref(ctx)) == -1
    assert M.lib().mpcmp_create(None, None, 0, 4, C.byref(ctx)) == -1


def test_cpp_shim_and_example_compile(M, tmp_path):
    """the header-only MotionPlanner mirror and the offline_trajectory counterpart build with plain g++"""
    exe = str(tmp_path / "offline_trajectory")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "offline_trajectory.cpp"),
                           "-L" + os.path.join(ROOT, "mpc_motion_planner_amd"), "-lmpcmp",
                           "-Wl,-rpath," + os.path.join(ROOT, "mpc_motion_planner_amd"), "-o", exe])
    # the selftest touches every member of the mirror, incl. the write side of `mpc` and the Ruckig look-alikes (-Wall -Werror:
    # the header must stay warning-free for a caller's build)
    exe2 = str(tmp_path / "shim_selftest")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "shim_selftest.cpp"),
                           "-L" + os.path.join(ROOT, "mpc_motion_planner_amd"), "-lmpcmp",
                           "-Wl,-rpath," + os.path.join(ROOT, "mpc_motion_planner_amd"), "-o", exe2])
    import torch
    if not torch.cuda.is_available():
        for e in (exe, exe2):
            r = subprocess.run([e], capture_output=True, text=True)
            assert r.returncode == 1 and "no CPU fallback" in r.stderr


# ---- scenario helpers of the robot wrapper (robot_utils/pandaWrapper.cpp:14-107): host code, no GPU needed ----
def _rand_q(rng, M):
    lim = M.default_limits()
    return lim["qmin"] + (lim["qmax"] - lim["qmin"]) * (0.15 + 0.7 * rng.random(7))


def test_tool_jacobian_and_forward_velocities_match_oracle(M):
    rng = np.random.default_rng(7)
    mdl, omdl = M.default_model(), o.default_model()
    for _ in range(20):
        q, qd = _rand_q(rng, M), rng.normal(size=7)
        J, p, R = M.tool_jacobian(mdl, q)
        Jo = o.frame_jacobian(q, np.array(omdl.tool), omdl)
        p7, R7, p8, pt = o.fk(q, omdl)
        assert np.abs(J - Jo).max() < 1e-12 and np.abs(p - pt).max() < 1e-12 and np.abs(R - R7).max() < 1e-12
        assert np.abs(M.forward_velocities(mdl, q, qd) - Jo @ qd).max() < 1e-12
        # the linear rows are the derivative of the tool position (central differences)
        for j in range(7):
            e = np.zeros(7); e[j] = 1e-6
            dp = (M.tool_jacobian(mdl, q + e)[1] - M.tool_jacobian(mdl, q - e)[1]) / 2e-6
            assert np.abs(dp - J[:3, j]).max() < 1e-8


def test_inverse_velocities_is_the_damped_pseudo_inverse(M):
    rng = np.random.default_rng(8)
    mdl = M.default_model()
    for _ in range(20):
        q = _rand_q(rng, M)
        lin, ang = rng.uniform(-1.7, 1.7, 3), np.zeros(3)          # examples/benchmark.cpp:20: zero angular speed
        J = M.tool_jacobian(mdl, q)[0]
        qd = M.inverse_velocities(mdl, q, lin, ang)
        ref = J.T @ np.linalg.solve(J @ J.T + 1e-5 * np.eye(6), np.concatenate([lin, ang]))
        assert np.abs(qd - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
        back = M.forward_velocities(mdl, q, qd)
        if np.linalg.cond(J @ J.T) < 1e3:                          # away from singularities the damping is negligible
            assert np.abs(back - np.concatenate([lin, ang])).max() < 1e-2


def test_inverse_kinematic_reaches_reachable_poses(M):
    rng = np.random.default_rng(9)
    mdl = M.default_model()
    ok = 0
    for _ in range(10):
        q_true = _rand_q(rng, M)
        _, p, R = M.tool_jacobian(mdl, q_true)
        q, conv, it = M.inverse_kinematic(mdl, R, p, q_init=q_true + 0.3 * rng.normal(size=7))
        _, p2, R2 = M.tool_jacobian(mdl, q)
        if conv:
            ok += 1
            assert it <= 1000 and np.abs(p2 - p).max() < 2e-4 and np.abs(R2 - R).max() < 2e-4
    assert ok >= 8
    # an unreachable pose hits the iteration cap and says so
    q, conv, it = M.inverse_kinematic(mdl, np.eye(3), np.array([3.0, 0.0, 0.0]))
    assert not conv and it == 1000
