"""General URDF reader (mpcmp_models_from_urdf; reference: robot_utils/pandaWrapper.cpp:3-12 hands any URDF to Pinocchio).
CPU only: the reader is host code of libmpcmp.so.  The expectations are computed independently in numpy from the same numbers
the URDF text holds (transform composition, inertia rotation, parallel-axis lumping, chain forward kinematics)."""
import os

import numpy as np
import pytest

import oracle_py as o


@pytest.fixture(scope="module")
def M():
    import mpc_motion_planner_amd as m
    if not os.path.exists(m.library_path()):
        pytest.skip("libmpcmp.so not built")
    return m


def rpy_R(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def xf(xyz=(0, 0, 0), rpy=(0, 0, 0)):
    T = np.eye(4); T[:3, :3] = rpy_R(rpy); T[:3, 3] = xyz
    return T


def axis_R(a):
    a = np.asarray(a, float) / np.linalg.norm(a)
    z = np.array([0.0, 0.0, 1.0])
    if np.allclose(a, z):
        return np.eye(3)
    if np.allclose(a, -z):
        return np.diag([1.0, -1.0, -1.0])
    v = np.cross(z, a); c = a[2]
    V = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    return np.eye(3) + V + V @ V / (1.0 + c)


def rot_axis(a, q):
    a = np.asarray(a, float) / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * K @ K


def fl(seq):
    """plain Python floats (numpy 2 prints np.float64(...) under %r)"""
    return tuple(float(v) for v in seq)


def inertial_xml(b):
    I = b["I"]
    return ('<inertial><origin xyz="%r %r %r" rpy="%r %r %r"/><mass value="%r"/>'
            '<inertia ixx="%r" ixy="%r" ixz="%r" iyy="%r" iyz="%r" izz="%r"/></inertial>'
            % (*fl(b["com"]), *fl(b["irpy"]), float(b["m"]), *fl((I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]))))


def make_spec(rng, n_chains, rotated=True, axes=None):
    """random robot: a base link, per chain a fixed base joint, seven revolute joints (a fixed, possibly rotated, intermediate
    link between joints 3 and 4), and two fixed frames behind joint 7 (the second one rotated)"""
    chains = []
    for c in range(n_chains):
        def body():
            A = rng.normal(size=(3, 3)); I = A @ A.T * 0.01 + np.eye(3) * 0.02
            return {"m": float(rng.uniform(0.5, 4)), "com": rng.normal(size=3) * 0.05,
                    "irpy": rng.normal(size=3) * (0.5 if rotated else 0.0), "I": I}
        ch = {"base": (rng.normal(size=3), (0.0, 0.0, float(rng.uniform(-3, 3))) if rotated else (0.0, 0.0, 0.0)),
              "joints": [], "links": [body() for _ in range(7)], "mid": body(), "mid_T": (rng.normal(size=3) * 0.1, rng.normal(size=3) * (0.4 if rotated else 0.0)),
              "f1": body(), "f1_T": (np.array([0.0, 0.0, 0.1]), (0.0, 0.0, 0.0)), "f2": body(), "f2_T": (rng.normal(size=3) * 0.1, rng.normal(size=3) * (0.6 if rotated else 0.0))}
        for i in range(7):
            ax = (0.0, 0.0, 1.0) if axes is None else axes[i]
            ch["joints"].append({"xyz": rng.normal(size=3) * 0.2, "rpy": rng.choice([-np.pi / 2, 0.0, np.pi / 2], size=3) if not rotated else rng.normal(size=3), "axis": ax})
        chains.append(ch)
    return chains


def write_urdf(path, chains):
    x = ['<?xml version="1.0"?>', '<robot name="t">', '<link name="base"/>']
    for c, ch in enumerate(chains):
        n = "c%d_" % c
        x.append('<link name="%sl0"/>' % n)
        x.append('<joint name="%sbase" type="fixed"><origin xyz="%r %r %r" rpy="%r %r %r"/><parent link="base"/><child link="%sl0"/></joint>'
                 % (n, *fl(ch["base"][0]), *fl(ch["base"][1]), n))
        parent = n + "l0"
        for i in range(7):
            j = ch["joints"][i]
            if i == 3:      # a fixed intermediate link between joints 3 and 4 (hangs on joint frame 3)
                x.append('<link name="%smid">%s</link>' % (n, inertial_xml(ch["mid"])))
                x.append('<joint name="%smidj" type="fixed"><origin xyz="%r %r %r" rpy="%r %r %r"/><parent link="%s"/><child link="%smid"/></joint>'
                         % (n, *fl(ch["mid_T"][0]), *fl(ch["mid_T"][1]), parent, n))
                parent = n + "mid"
            x.append('<link name="%sl%d">%s</link>' % (n, i + 1, inertial_xml(ch["links"][i])))
            x.append('<joint name="%sj%d" type="revolute"><origin xyz="%r %r %r" rpy="%r %r %r"/><parent link="%s"/><child link="%sl%d"/>'
                     '<axis xyz="%r %r %r"/><limit lower="-2" upper="2" effort="10" velocity="2"/></joint>'
                     % (n, i + 1, *fl(j["xyz"]), *fl(j["rpy"]), parent, n, i + 1, *fl(j["axis"])))
            parent = "%sl%d" % (n, i + 1)
        for k, nm in ((1, "f1"), (2, "f2")):
            x.append('<link name="%s%s">%s</link>' % (n, nm, inertial_xml(ch[nm])))
            x.append('<joint name="%s%sj" type="fixed"><origin xyz="%r %r %r" rpy="%r %r %r"/><parent link="%s"/><child link="%s%s"/></joint>'
                     % (n, nm, *fl(ch[nm + "_T"][0]), *fl(ch[nm + "_T"][1]), parent, n, nm))
            parent = n + nm
    x.append("</robot>")
    open(path, "w").write("\n".join(x))


def expected(ch):
    """model fields of one chain, numpy"""
    out = {"R0": [], "p": [], "mass": [], "com": [], "I": []}
    C = xf(*ch["base"])                        # link frame in the current model frame
    bodies = None
    def add(body, C):
        R = C[:3, :3] @ rpy_R(body["irpy"])
        bodies.append((body["m"], C[:3, :3] @ body["com"] + C[:3, 3], R @ body["I"] @ R.T))
    def flush():
        M_ = sum(b[0] for b in bodies); c = sum(b[0] * b[1] for b in bodies) / M_
        I = sum(b[2] + b[0] * (np.dot(b[1] - c, b[1] - c) * np.eye(3) - np.outer(b[1] - c, b[1] - c)) for b in bodies)
        out["mass"].append(M_); out["com"].append(c); out["I"].append(I)
    for i in range(7):
        if i == 3:
            C = C @ xf(*ch["mid_T"]); add(ch["mid"], C)
        j = ch["joints"][i]
        Ra = np.eye(4); Ra[:3, :3] = axis_R(j["axis"])
        P = C @ xf(j["xyz"], j["rpy"]) @ Ra
        out["R0"].append(P[:3, :3]); out["p"].append(P[:3, 3])
        if bodies is not None:
            flush()
        bodies = []
        C = np.linalg.inv(Ra)
        add(ch["links"][i], C)
    C1 = C @ xf(*ch["f1_T"]); add(ch["f1"], C1)
    C2 = C1 @ xf(*ch["f2_T"]); add(ch["f2"], C2)
    flush()
    out["link8"] = C1[:3, 3]; out["tool"] = C2[:3, 3]
    return out


def chain_fk_tool(ch, q):
    T = xf(*ch["base"])
    for i in range(7):
        if i == 3:
            T = T @ xf(*ch["mid_T"])
        j = ch["joints"][i]
        R = np.eye(4); R[:3, :3] = rot_axis(j["axis"], q[i])
        T = T @ xf(j["xyz"], j["rpy"]) @ R
    return (T @ xf(*ch["f1_T"]) @ xf(*ch["f2_T"]))[:3, 3]


def check_models(M, models, chains, tol=1e-12):
    assert len(models) == len(chains)
    for m, ch in zip(models, chains):
        e = expected(ch)
        for i in range(7):
            assert np.abs(np.array(m.R0[i][:]).reshape(3, 3) - e["R0"][i]).max() < tol
            assert np.abs(np.array(m.p[i][:]) - e["p"][i]).max() < tol
            assert abs(m.mass[i] - e["mass"][i]) < tol
            assert np.abs(np.array(m.com[i][:]) - e["com"][i]).max() < tol
            assert np.abs(np.array(m.I[i][:]).reshape(3, 3) - e["I"][i]).max() < tol
        assert np.abs(np.array(m.tool[:]) - e["tool"]).max() < tol and np.abs(np.array(m.link8[:]) - e["link8"]).max() < tol
        assert m.gravity[2] == -9.81


def test_two_arm_urdf_rotated_everything(M, tmp_path):
    """two chains on one base; rotated base placements, fixed joints (one between revolute joints), inertial frames and tool frame"""
    rng = np.random.default_rng(5)
    chains = make_spec(rng, 2, rotated=True)
    p = str(tmp_path / "two_arm.urdf"); write_urdf(p, chains)
    models = M.models_from_urdf(p)
    check_models(M, models, chains)
    # kinematics of the reader's model == kinematics of the URDF (host FK of the product library)
    for m, ch in zip(models, chains):
        for _ in range(5):
            q = rng.uniform(-2, 2, 7)
            _, ptool, _ = M.tool_jacobian(m, q)
            assert np.abs(ptool - chain_fk_tool(ch, q)).max() < 1e-12
    with pytest.raises(M.MpcmpError):          # the single-chain entry point refuses a two-chain robot
        M.model_from_urdf(p)


def test_general_joint_axes(M, tmp_path):
    """axes other than +z (-z, x, y, oblique) are absorbed by a constant rotation of the child frame"""
    rng = np.random.default_rng(6)
    axes = [(0, 0, -1), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0.3, -0.5, 0.8), (0, -1, 0), (-1, 0, 0)]
    chains = make_spec(rng, 1, rotated=True, axes=axes)
    p = str(tmp_path / "axes.urdf"); write_urdf(p, chains)
    models = M.models_from_urdf(p)
    check_models(M, models, chains)
    for _ in range(8):
        q = rng.uniform(-2, 2, 7)
        _, ptool, _ = M.tool_jacobian(models[0], q)
        assert np.abs(ptool - chain_fk_tool(chains[0], q)).max() < 1e-12
    # gravity torques: the product's model through the oracle's RNEA == finite differences of the potential energy of the URDF's bodies
    ch = chains[0]
    def potential(q):
        T = xf(*ch["base"]); U = 0.0
        for i in range(7):
            if i == 3:
                T = T @ xf(*ch["mid_T"]); U += 9.81 * ch["mid"]["m"] * (T[:3, :3] @ ch["mid"]["com"] + T[:3, 3])[2]
            j = ch["joints"][i]
            R = np.eye(4); R[:3, :3] = rot_axis(j["axis"], q[i])
            T = T @ xf(j["xyz"], j["rpy"]) @ R
            U += 9.81 * ch["links"][i]["m"] * (T[:3, :3] @ ch["links"][i]["com"] + T[:3, 3])[2]
        T1 = T @ xf(*ch["f1_T"]); U += 9.81 * ch["f1"]["m"] * (T1[:3, :3] @ ch["f1"]["com"] + T1[:3, 3])[2]
        T2 = T1 @ xf(*ch["f2_T"]); U += 9.81 * ch["f2"]["m"] * (T2[:3, :3] @ ch["f2"]["com"] + T2[:3, 3])[2]
        return U
    om = o.Model.from_buffer_copy(models[0])
    q = rng.uniform(-1.5, 1.5, 7)
    tau = o.rnea(q, np.zeros(7), np.zeros(7), model=om)
    g = np.array([(potential(q + 1e-6 * np.eye(7)[i]) - potential(q - 1e-6 * np.eye(7)[i])) / 2e-6 for i in range(7)])
    assert np.abs(tau - g).max() < 1e-6 * max(1.0, np.abs(g).max())


def test_unrotated_urdf_is_bit_exact(M, tmp_path):
    """without rotated fixed joints / inertial frames the reader hands the numbers of the file through unchanged (identity factors
    are skipped), as the round-1 reader did for panda_arm.urdf"""
    rng = np.random.default_rng(7)
    chains = make_spec(rng, 1, rotated=False)
    chains[0]["base"] = (np.zeros(3), (0.0, 0.0, 0.0))
    chains[0]["mid_T"] = (np.zeros(3), (0.0, 0.0, 0.0)); chains[0]["mid"]["m"] = 0.0; chains[0]["mid"]["I"] = np.zeros((3, 3))
    p = str(tmp_path / "plain.urdf"); write_urdf(p, chains)
    m = M.model_from_urdf(p)
    ch = chains[0]
    for i in (0, 1, 3, 4, 5):       # (the third joint frame carries the massless intermediate link: lumped; the seventh the tool bodies)
        assert np.array_equal(np.array(m.p[i][:]), ch["joints"][i]["xyz"])
        assert np.array_equal(np.array(m.R0[i][:]).reshape(3, 3), rpy_R(ch["joints"][i]["rpy"]))
        assert m.mass[i] == ch["links"][i]["m"] and np.array_equal(np.array(m.com[i][:]), ch["links"][i]["com"])
        I = ch["links"][i]["I"]
        assert np.array_equal(np.array(m.I[i][:]).reshape(3, 3), np.array([[I[0, 0], I[0, 1], I[0, 2]], [I[0, 1], I[1, 1], I[1, 2]], [I[0, 2], I[1, 2], I[2, 2]]]))


def test_reader_rejections(M, tmp_path):
    rng = np.random.default_rng(8)
    chains = make_spec(rng, 1)
    p = str(tmp_path / "ok.urdf"); write_urdf(p, chains)
    txt = open(p).read()
    bad = str(tmp_path / "prismatic.urdf"); open(bad, "w").write(txt.replace('name="c0_j3" type="revolute"', 'name="c0_j3" type="prismatic"'))
    with pytest.raises(M.MpcmpError):
        M.models_from_urdf(bad)
    six = str(tmp_path / "six.urdf"); open(six, "w").write(txt.replace('name="c0_j7" type="revolute"', 'name="c0_j7" type="fixed"'))
    with pytest.raises(M.MpcmpError):
        M.models_from_urdf(six)
    # a second revolute joint hanging on link 2: a tree, not a serial chain
    branch = txt.replace("</robot>", '<link name="extra"/><joint name="jx" type="revolute"><parent link="c0_l2"/><child link="extra"/><axis xyz="0 0 1"/></joint></robot>')
    br = str(tmp_path / "branch.urdf"); open(br, "w").write(branch)
    with pytest.raises(M.MpcmpError):
        M.models_from_urdf(br)
    three = make_spec(rng, 3)
    p3 = str(tmp_path / "three.urdf"); write_urdf(p3, three)
    with pytest.raises(M.MpcmpError):
        M.models_from_urdf(p3, max_chains=2)
    assert len(M.models_from_urdf(p3)) == 3


def test_dual_panda_urdf_matches_arm_models(M, tmp_path):
    """the dual-arm robot of BASELINE.json configs[3] (two compiled-in Pandas on one base, capi.DUAL_BASES) written as ONE URDF
    gives the models the benches and GPU tests build with arm_models()"""
    d = M.default_model()
    x = ['<robot name="dual">', '<link name="world"/>']
    for a, (yaw, xyz) in enumerate(M.DUAL_BASES):
        n = "a%d_" % a
        x.append('<link name="%slink0"/><joint name="%smount" type="fixed"><origin xyz="%r %r %r" rpy="0 0 %r"/><parent link="world"/><child link="%slink0"/></joint>'
                 % (n, n, *fl(xyz), float(yaw), n))
        parent = n + "link0"
        for i in range(7):
            R = np.array(d.R0[i][:]).reshape(3, 3)
            rpy = (float(np.arctan2(R[2, 1], R[2, 2])), float(-np.arcsin(R[2, 0])), float(np.arctan2(R[1, 0], R[0, 0])))
            I = np.array(d.I[i][:]).reshape(3, 3)
            x.append('<link name="%slink%d"><inertial><origin xyz="%r %r %r"/><mass value="%r"/><inertia ixx="%r" ixy="%r" ixz="%r" iyy="%r" iyz="%r" izz="%r"/></inertial></link>'
                     % (n, i + 1, *fl(d.com[i][:]), float(d.mass[i]), *fl((I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]))))
            x.append('<joint name="%sjoint%d" type="revolute"><origin xyz="%r %r %r" rpy="%r %r %r"/><parent link="%s"/><child link="%slink%d"/><axis xyz="0 0 1"/></joint>'
                     % (n, i + 1, *fl(d.p[i][:]), *rpy, parent, n, i + 1))
            parent = "%slink%d" % (n, i + 1)
        x.append('<link name="%slink8"/><joint name="%sj8" type="fixed"><origin xyz="%r %r %r"/><parent link="%s"/><child link="%slink8"/></joint>'
                 % (n, n, *fl(d.link8[:]), parent, n))
        off = np.array(d.tool[:]) - np.array(d.link8[:])
        x.append('<link name="%stool"/><joint name="%sjt" type="fixed"><origin xyz="%r %r %r"/><parent link="%slink8"/><child link="%stool"/></joint>'
                 % (n, n, *fl(off), n, n))
    x.append("</robot>")
    p = str(tmp_path / "dual_panda.urdf"); open(p, "w").write("\n".join(x))
    got, want = M.models_from_urdf(p), M.arm_models(M.DUAL_BASES)
    assert len(got) == 2
    for g, w in zip(got, want):
        for f in ["R0", "p", "mass", "com", "I", "tool", "link8", "gravity"]:
            assert np.abs(np.array(getattr(g, f)) - np.array(getattr(w, f))).max() < 1e-15, f
