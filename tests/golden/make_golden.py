#!/usr/bin/env python3
"""Extract numeric golden vectors from the reference's notebooks (numbers only).

Run ONCE in the build container (where /root/reference exists); the GPU box never
needs it.  Sources (all stored *outputs* of notebooks, i.e. data, not code):

  analysis/data_analysis.ipynb cell 1 (.ipynb:23-25620)  -> kat_rnea.csv, gold_traj.json
      56 plotly traces: trace 8*j+c, j=joint, c in {0:q_mpc,1:q_rk,2:v_mpc,3:v_rk,
      4:a_mpc,5:a_rk,6:tau_mpc,7:tau_rk}; 201 samples each at 6 s.f.
      target q_j = layout.shapes[10*j].y0, target v_j = layout.shapes[10*j+3].y0
  analysis/data_analysis.ipynb cell 3 (.ipynb:25741-27907) -> kat_fk_link8.csv
      FK of frame panda_link8 at the 201 MPC q samples, full double precision
  test_develop/test_rnea_derivatives.ipynb cells 4,5,7      -> kat_jac.json
      (cells 2,3 are stale w.r.t. the shipped URDF and are deliberately excluded)
"""
import json, os, sys
import numpy as np

REF = os.environ.get("MPCMP_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    nb = json.load(open(os.path.join(REF, "analysis/data_analysis.ipynb")))
    fig = nb["cells"][1]["outputs"][0]["data"]["application/vnd.plotly.v1+json"]
    tr = fig["data"]
    assert len(tr) == 56
    def get(c):  # -> (time[201], val[201,7])
        t = np.array(tr[c]["x"], dtype=float)
        v = np.stack([np.array(tr[8 * j + c]["y"], dtype=float) for j in range(7)], axis=1)
        return t, v
    t_mpc, q_mpc = get(0); t_rk, q_rk = get(1)
    _, v_mpc = get(2); _, v_rk = get(3)
    _, a_mpc = get(4); _, a_rk = get(5)
    _, tau_mpc = get(6); _, tau_rk = get(7)
    sh = fig["layout"]["shapes"]
    q_target = [sh[10 * j]["y0"] for j in range(7)]
    v_target = [sh[10 * j + 3]["y0"] for j in range(7)]

    # KAT-RNEA: 402 rows: src(0=mpc,1=rk), q7, v7, a7, tau7
    rows = []
    for src, (q, v, a, tau) in enumerate([(q_mpc, v_mpc, a_mpc, tau_mpc), (q_rk, v_rk, a_rk, tau_rk)]):
        for i in range(201):
            rows.append(np.concatenate([[src], q[i], v[i], a[i], tau[i]]))
    np.savetxt(os.path.join(OUT, "kat_rnea.csv"), np.array(rows), fmt="%.9g", delimiter=",",
               header="src(0=mpc 1=ruckig),q1..q7,v1..v7,a1..a7,tau1..tau7  [pinocchio::rnea outputs stored in data_analysis.ipynb cell 1]")

    gold = {
        "source": "analysis/data_analysis.ipynb cell 1 output; figure title %r" % fig["layout"]["title"]["text"],
        "margins": [0.9, 0.9, 0.5, 0.9, 0.1],
        "q0": q_mpc[0].tolist(), "v0": v_mpc[0].tolist(),
        "qT": q_target, "vT": v_target,
        "T_ruckig": float(t_rk[-1]), "T_mpc": float(t_mpc[-1]),
        "t_mpc": t_mpc.tolist(), "q_mpc": q_mpc.tolist(), "v_mpc": v_mpc.tolist(),
        "a_mpc": a_mpc.tolist(), "tau_mpc": tau_mpc.tolist(),
        "t_rk": t_rk.tolist(), "q_rk": q_rk.tolist(), "v_rk": v_rk.tolist(),
        "a_rk": a_rk.tolist(), "tau_rk": tau_rk.tolist(),
    }
    json.dump(gold, open(os.path.join(OUT, "gold_traj.json"), "w"))

    fig3 = nb["cells"][3]["outputs"][0]["data"]["application/vnd.plotly.v1+json"]
    xyz = np.stack([np.array(fig3["data"][k]["y"], dtype=float) for k in range(3)], axis=1)
    np.savetxt(os.path.join(OUT, "kat_fk_link8.csv"), np.concatenate([q_mpc, xyz], axis=1), fmt="%.17g",
               delimiter=",", header="q1..q7 (6 s.f., from cell 1), x,y,z of frame panda_link8 (full precision, cell 3)")

    jac = {
        "source": "test_develop/test_rnea_derivatives.ipynb stored outputs of cells 4, 5, 7",
        "J1": {"q": [0.0, 0.0, 0.0, -1.5, 0.0, 1.0, 0.0], "qd": [1.0] * 7,
               "world_aligned_joint7_velocity": [0.35818945, 1.12140565, -0.00527273, 0.51806945, -1.0, 1.19315464],
               "digits": 8},
        "J2": {"q": [1.0, 1.0, 1.0, 1.0, -1.0, 1.0, -1.0], "qd": [5.0, 5.0, 5.0, 5.0, -5.0, 5.0, -5.0],
               "world_aligned_joint7_velocity": [1.078, 1.585, -3.51, -1.201, 0.781, 1.85], "digits": 3},
        "FK1": {"q": [0.2, 0.3, -0.1, -1.1, 0.2, 1.4, 0.5],
                "oMi7_p": [0.618923, 0.0804767, 0.757975],
                "oMi7_R": [[0.934675, -0.354976, -0.019329], [-0.347142, -0.923067, 0.165649],
                           [-0.0766435, -0.148118, -0.985995]], "digits": 6},
    }
    # cross-check the literal values above against the notebook text so typos cannot creep in
    nb2 = json.load(open(os.path.join(REF, "test_develop/test_rnea_derivatives.ipynb")))
    txt5 = "".join(nb2["cells"][5]["outputs"][0]["text"])
    assert "0.35818945" in txt5 and "1.19315464" in txt5
    txt4 = "".join(nb2["cells"][4]["outputs"][0]["text"])
    assert "1.078  1.585 -3.51  -1.201  0.781  1.85" in txt4
    txt7 = "".join(nb2["cells"][7]["outputs"][0]["data"]["text/plain"])
    assert "0.618923 0.0804767  0.757975" in txt7 and "-0.985995" in txt7
    json.dump(jac, open(os.path.join(OUT, "kat_jac.json"), "w"), indent=1)
    print("wrote golden vectors to", OUT)


if __name__ == "__main__":
    sys.exit(main())
