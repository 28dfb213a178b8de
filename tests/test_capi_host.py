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


def test_struct_layouts_match_header(M):
    assert C.sizeof(M.Info) == 64 and M.INFO_DTYPE.itemsize == 64
    assert C.sizeof(M.Config) == C.sizeof(o.Config) and C.sizeof(M.Model) == C.sizeof(o.Model)


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
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr
