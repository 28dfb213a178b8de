// Drives the header-only MotionPlanner shim the way a reference caller would and prints every result as one JSON line, so
// that tests/test_gpu_parity.py can compare it with the CPU oracle:
//   warm_start(T, q, v, a)  ->  solve_trajectory(false)  ->  get_MPC_point (below and beyond T: the clamp quirk of
//   motionPlanner.hpp:120-121)  ->  solve_trajectory(true)  ->  get_RK_point (inside and beyond the duration)  ->  the write side
//   of `mpc` (x_guess / u_guess / p_guess / solve, control_bounds)  ->  the Ruckig members (otg.calculate, trajectory.at_time).
//   g++ -O2 -std=c++17 -Iinclude examples/shim_selftest.cpp -Lmpc_motion_planner_amd -lmpcmp -Wl,-rpath,$PWD/mpc_motion_planner_amd
#include <cmath>
#include <cstdio>
#include <iostream>
#include "mpcmp_motion_planner.hpp"

static void dump(const char *name, const MotionPlanner::Vec7 &q, const MotionPlanner::Vec7 &v, const MotionPlanner::Vec7 &a,
                 const MotionPlanner::Vec7 &tau, bool last = false) {
    std::printf("\"%s\": [", name);
    for (int j = 0; j < 7; j++) std::printf("%.17g, ", q(j));
    for (int j = 0; j < 7; j++) std::printf("%.17g, ", v(j));
    for (int j = 0; j < 7; j++) std::printf("%.17g, ", a(j));
    for (int j = 0; j < 7; j++) std::printf(j == 6 ? "%.17g" : "%.17g, ", tau(j));
    std::printf(last ? "]" : "], ");
}

int main() {
    try {
        MotionPlanner planner("", 4, 3);                      // 13 nodes, 3 SQP iterations
        planner.set_constraint_margins(0.9, 0.9, 0.5, 0.9, 0.1);
        planner.seed(7);
        MotionPlanner::Vec7 q0, v0, qT, vT, q, v, a, tau;
        planner.sample_random_state(q0, v0);
        planner.sample_random_state(qT, vT);
        planner.set_current_state(q0, v0);
        planner.set_target_state(qT, vT);
        // a regularly time-spaced guess: straight line in joint space over 41 points, constant velocity, zero acceleration
        const int nP = 41;
        const double Tg = 2.0;
        std::vector<double> pq(7 * nP), pv(7 * nP), pa(7 * nP, 0.0);
        for (int i = 0; i < nP; i++)
            for (int j = 0; j < 7; j++) { pq[7 * i + j] = q0(j) + (qT(j) - q0(j)) * i / (nP - 1.0); pv[7 * i + j] = (qT(j) - q0(j)) / Tg; }
        planner.warm_start(Tg, pq, pv, pa);
        planner.solve_trajectory(false);
        std::printf("{\"x0\": [");
        for (int j = 0; j < 14; j++) std::printf(j == 13 ? "%.17g" : "%.17g, ", planner.current_state(j));
        std::printf("], \"xf\": [");
        for (int j = 0; j < 14; j++) std::printf(j == 13 ? "%.17g" : "%.17g, ", planner.target_state(j));
        std::printf("], \"T_warm\": %.17g, \"iters_warm\": %d, ", planner.solution_T(), planner.last_info.qp_iters_total);
        const double Tw = planner.solution_T();
        planner.get_MPC_point(0.37 * Tw, q, v, a, tau); dump("mpc_point_in", q, v, a, tau);
        planner.get_MPC_point(Tw + 0.3, q, v, a, tau); dump("mpc_point_beyond", q, v, a, tau);
        std::printf("\"mpc_p\": %.17g, \"mpc_iter\": %d, ", planner.mpc.solution_p()[0], planner.mpc.info().iter);
        planner.solve_trajectory(true);                        // jerk-limited (Ruckig stand-in) warm start
        std::printf("\"T_rk_solve\": %.17g, \"iters_rk\": %d, ", planner.solution_T(), planner.last_info.qp_iters_total);
        planner.get_RK_point(0.4, q, v, a, tau); dump("rk_point_in", q, v, a, tau);
        planner.get_RK_point(1e3, q, v, a, tau); dump("rk_point_beyond", q, v, a, tau);
        // write side of `mpc` (motionPlanner.cpp:172-174,184): the straight-line guess again, through x_guess / u_guess / p_guess + solve()
        {
            const std::vector<double> tn = planner.mpc.time_nodes();
            std::vector<double> xg(14 * tn.size()), ug(7 * tn.size(), 0.0);
            for (size_t i = 0; i < tn.size(); i++) {
                const int idx = (int)std::lround(tn[i] * (nP - 1));
                for (int j = 0; j < 7; j++) { xg[14 * i + j] = pq[7 * idx + j]; xg[14 * i + 7 + j] = pv[7 * idx + j]; }
            }
            planner.mpc.x_guess(xg); planner.mpc.u_guess(ug); planner.mpc.p_guess(Tg);
            planner.mpc.solve();
            std::printf("\"T_guess_api\": %.17g, \"iters_guess_api\": %d, ", planner.mpc.solution_p()[0], planner.mpc.info().qp_iters_total);
        }
        // mpc.control_bounds (motionPlanner.cpp:75): 40 % of the acceleration limits, then the margins' bounds again
        {
            MotionPlanner::Vec7 lo, hi;
            for (int j = 0; j < 7; j++) { hi(j) = 0.4 * planner.robot.max_acceleration(j); lo(j) = -hi(j); }
            planner.mpc.control_bounds(lo, hi);
            planner.solve_trajectory(false);                   // from the previous solution, end states re-pinned (motionPlanner.cpp:199-207)
            std::printf("\"T_ctrl_box\": %.17g, \"iters_ctrl_box\": %d, \"u_max_ctrl_box\": [", planner.solution_T(), planner.last_info.qp_iters_total);
            for (int j = 0; j < 7; j++) {
                double m = 0;
                for (int i = 0; i < planner.num_nodes(); i++) m = std::fmax(m, std::fabs(planner.solution_u()[(size_t)7 * i + j]));
                std::printf(j == 6 ? "%.17g], " : "%.17g, ", m / planner.robot.max_acceleration(j));
            }
            planner.set_constraint_margins(0.9, 0.9, 0.5, 0.9, 0.1);
        }
        // the Ruckig members (motionPlanner.hpp:35-37, motionPlanner.cpp:149,160)
        {
            const int res = (int)planner.otg.calculate(planner.input, planner.trajectory);
            std::array<double, 7> p7, v7, a7;
            planner.trajectory.at_time(0.4, p7, v7, a7);
            std::printf("\"otg_result\": %d, \"rk_duration\": %.17g, \"rk_at_time\": [", res, planner.trajectory.get_duration());
            for (int j = 0; j < 7; j++) std::printf("%.17g, ", p7[j]);
            for (int j = 0; j < 7; j++) std::printf("%.17g, ", v7[j]);
            for (int j = 0; j < 7; j++) std::printf(j == 6 ? "%.17g], " : "%.17g, ", a7[j]);
            mpcmp_shim::InputParameter slow = planner.input;   // caller-written limits
            for (int j = 0; j < 7; j++) { slow.max_jerk[j] *= 0.5; slow.max_velocity[j] *= 0.8; }
            mpcmp_shim::Trajectory tr2;
            planner.otg.calculate(slow, tr2);
            std::printf("\"rk_duration_slow\": %.17g, ", tr2.get_duration());
            // an invalid input record is the RETURN VALUE (ruckig::Result::Error*), not an exception; at_time without a trajectory throws
            mpcmp_shim::InputParameter bad = planner.input;
            bad.max_acceleration[3] = 0.0;
            mpcmp_shim::Trajectory tr3;
            const int res_bad = (int)planner.otg.calculate(bad, tr3);
            bool threw = false;
            try { tr3.at_time(0.1, p7, v7, a7); } catch (const std::logic_error &) { threw = true; }
            std::printf("\"otg_result_bad\": %d, \"at_time_threw\": %d, ", res_bad, threw ? 1 : 0);
            // solve_trajectory(true) warm-starts from ALL of `input` (motionPlanner.cpp:146-149): a caller-written boundary acceleration and a tighter
            // velocity limit change the guess's duration exactly as they change otg.calculate's
            mpcmp_shim::InputParameter keep = planner.input;
            for (int j = 0; j < 7; j++) { planner.input.current_acceleration[j] = 0.3 * planner.input.max_acceleration[j]; planner.input.max_velocity[j] *= 0.8; }
            mpcmp_shim::Trajectory tr4;
            planner.otg.calculate(planner.input, tr4);
            planner.solve_trajectory(true);
            planner.get_RK_point(0.0, q, v, a, tau);
            std::printf("\"rk_duration_input\": %.17g, \"guess_T_input\": %.17g, \"rk_a0_ratio\": %.17g, ", tr4.get_duration(), planner.guess_T(), a(2) / planner.input.max_acceleration[2]);
            planner.input = keep;
        }
        // robot.data look-alike (examples/benchmark.cpp:108-110)
        pinocchio::forwardKinematics(planner.robot.model, planner.robot.data, qT);
        pinocchio::updateFramePlacement(planner.robot.model, planner.robot.data, planner.robot.frame_id);
        std::printf("\"tool_z\": %.17g}\n", planner.robot.data.oMf[planner.robot.frame_id].translation()[2]);
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
