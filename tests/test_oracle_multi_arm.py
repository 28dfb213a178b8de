"""Oracle, multi-arm form (BASELINE.json configs[3]: 14-DoF dual Panda, N = 25): the same OCP as robot_ocp.hpp:31-213 with
doubled sizes (NX=28, NU=14, NG=16).  The arms are independent chains that couple only through the final time T."""
import numpy as np

import oracle_py as o

MARGINS = (0.9, 0.9, 0.5, 0.9, 0.1)


def _limits():
    lim = o.default_limits()
    return MARGINS[1] * lim["vmax"], MARGINS[2] * lim["amax"], MARGINS[4] * lim["jmax"]


def _states(B, off=0):
    from mpc_motion_planner_amd import scenarios
    a0, af = scenarios.make_batch(B, MARGINS, stream_offset=off)
    b0, bf = scenarios.make_batch(B, MARGINS, stream_offset=off + 7919)
    x0 = np.stack([o.merge_arm_states([a0[b], b0[b]]) for b in range(B)])
    xf = np.stack([o.merge_arm_states([af[b], bf[b]]) for b in range(B)])
    return x0, xf, (a0, af, b0, bf)


def test_single_arm_through_the_multi_arm_entry_point_is_identical():
    x0, xf, (a0, af, _, _) = _states(2)
    cfg = o.default_config(4, 3, margins=MARGINS)
    for b in range(2):
        xg, ug, Tg = o.warm_start_jerk(4, *_limits(), a0[b], af[b])
        r1 = o.solve(cfg, a0[b], af[b], xg, ug, Tg)
        r2 = o.solve_multi(o.arm_models(o.DUAL_BASES[:1]), cfg, a0[b], af[b], xg, ug, Tg)
        assert np.array_equal(r1[0], r2[0]) and np.array_equal(r1[1], r2[1]) and r1[2] == r2[2]


def test_dual_arm_solution_is_feasible_for_both_arms_and_slower_than_either_alone():
    x0, xf, (a0, af, b0, bf) = _states(2, off=31)
    models = o.arm_models(o.DUAL_BASES)
    cfg = o.default_config(4, 10, margins=MARGINS)
    for b in range(2):
        xg, ug, Tg = o.warm_start_jerk_multi(4, *_limits(), x0[b], xf[b])
        assert np.array_equal(xg[0], x0[b]) and np.array_equal(xg[-1], xf[b])
        xs, us, T, info = o.solve_multi(models, cfg, x0[b], xf[b], xg, ug, Tg)
        assert (info.status & 7) == 0 and info.defect_inf < 2e-2 and info.path_viol_inf < 2e-2 and info.term_err_inf < 2e-2
        singles = []
        for (s0, sf, mdl) in ((a0[b], af[b], o.arm_models(o.DUAL_BASES[:1])), (b0[b], bf[b], o.arm_models(o.DUAL_BASES[1:]))):
            wg = o.warm_start_jerk(4, *_limits(), s0, sf)
            singles.append(o.solve_multi(mdl, cfg, s0, sf, *wg)[2])
        assert T > max(singles) - 2e-2 and T < Tg + 1e-9          # coupled only through T: the slower arm sets it
        # per-arm dynamics: torque rows of arm B evaluated with arm B's own (mounted) model
        for k in (0, 6, 12):
            for a, mdl in enumerate((o.arm_models(o.DUAL_BASES[:1])[0], o.arm_models(o.DUAL_BASES[1:])[0])):
                q, v, acc = xs[k, 7 * a:7 * a + 7], xs[k, 14 + 7 * a:14 + 7 * a + 7], us[k, 7 * a:7 * a + 7]
                tau = o.rnea(q, v, acc, model=mdl)
                assert np.all(np.abs(tau) <= np.array(cfg.ubg[:7]) + 5e-2)


def test_yawed_base_does_not_change_the_torques():
    """gravity is along world z, so mounting the arm rotated about z and shifted leaves RNEA unchanged; the tool height too"""
    rng = np.random.default_rng(0)
    m0 = o.arm_models(o.DUAL_BASES[:1])[0]; m1 = o.arm_models(o.DUAL_BASES[1:])[0]
    for _ in range(5):
        q, v, a = rng.uniform(-1.5, 1.5, 7), rng.uniform(-1, 1, 7), rng.uniform(-3, 3, 7)
        assert np.abs(o.rnea(q, v, a, model=m0) - o.rnea(q, v, a, model=m1)).max() < 1e-11
        p0, p1 = o.fk(q, model=m0)[3], o.fk(q, model=m1)[3]
        assert abs(p0[2] - p1[2]) < 1e-12 and abs((1.0 - p1[0]) - p0[0]) < 1e-12 and abs(p1[1] + p0[1]) < 1e-12
