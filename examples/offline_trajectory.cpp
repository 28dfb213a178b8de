// Counterpart of the reference's examples/offline_trajectory.cpp on the mpcmp C ABI: one random (start,target)
// pair, margins (0.9,0.9,0.5,0.9,0.1), solve, resample 201 points of the initial guess and of the MPC solution,
// write the 403x29 text file the reference's analysis/data_analysis.ipynb reads.
//   g++ -O2 -std=c++17 -Iinclude examples/offline_trajectory.cpp -Lmpc_motion_planner_amd -lmpcmp
//     -Wl,-rpath,$PWD/mpc_motion_planner_amd -o offline_trajectory
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include "mpcmp_motion_planner.hpp"

int main(int argc, char **argv) {
    const char *urdf = argc > 1 ? argv[1] : "";               // "" -> compiled-in panda_arm model
    const char *out = argc > 2 ? argv[2] : "optimal_solution.txt";
    try {
        MotionPlanner planner(urdf);                           // 19 nodes, 2 SQP iterations: reference as shipped
        planner.set_constraint_margins(0.9, 0.9, 0.5, 0.9, 0.1);
        if (argc > 3) planner.seed(std::strtoull(argv[3], nullptr, 10));
        MotionPlanner::Vec7 q0, v0, qT, vT;
        planner.sample_random_state(q0, v0);
        planner.sample_random_state(qT, vT);
        // target task velocity inside the Cartesian limits, scaled down otherwise (reference examples/offline_trajectory.cpp:26-41)
        auto norm3 = [](const mpcmp_shim::Mat<6, 1> &t, int o) { return std::sqrt(t(o) * t(o) + t(o + 1) * t(o + 1) + t(o + 2) * t(o + 2)); };
        mpcmp_shim::Mat<6, 1> task_velocity = planner.robot.forward_velocities(qT, vT);
        if (norm3(task_velocity, 0) > planner.robot.max_linear_velocity) {
            const double f = 0.9 * planner.robot.max_linear_velocity / norm3(task_velocity, 0);
            for (int j = 0; j < 7; j++) vT(j) *= f;
            task_velocity = planner.robot.forward_velocities(qT, vT);
        }
        if (norm3(task_velocity, 3) > planner.robot.max_angular_velocity) {
            const double f = 0.9 * planner.robot.max_angular_velocity / norm3(task_velocity, 3);
            for (int j = 0; j < 7; j++) vT(j) *= f;
        }
        planner.set_current_state(q0, v0);
        planner.set_target_state(qT, vT);
        if (planner.check_state_in_bounds(q0, v0) != 0 || planner.check_state_in_bounds(qT, vT) != 0)
            throw std::runtime_error("Initial or target state out of bounds");
        planner.solve_trajectory(true);
        planner.write_optimal_solution<200>(out);
        MotionPlanner::Vec7 q, v, a, tau;
        planner.get_MPC_point(0.5 * planner.solution_T(), q, v, a, tau);
        std::printf("T_mpc = %.6f s   defect = %.2e   terminal error = %.2e   tau1(T/2) = %.4f   -> %s\n", planner.solution_T(),
                    planner.last_info.defect_inf, planner.last_info.term_err_inf, tau(0), out);
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
