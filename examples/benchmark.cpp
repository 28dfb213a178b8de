// Counterpart of the reference's examples/benchmark.cpp: 1000 targets from the default start state, margins
// (0.8, 0.8, 0.6, 0.9, 0.1), solved as ONE batch on the GPU; appends the 162-column rows that
// analysis/benchmark_analysis.ipynb reads.  Targets are drawn as the reference does (benchmark.cpp:19-42): a random
// configuration, a joint velocity realising a random linear tool velocity with zero angular speed
// (PandaWrapper::inverse_velocities), scaled back inside the task- and joint-velocity limits.
//   g++ -O2 -std=c++17 -Iinclude examples/benchmark.cpp -Lmpc_motion_planner_amd -lmpcmp -Wl,-rpath,$PWD/mpc_motion_planner_amd -o mpc_benchmark
#include <cmath>
#include <cstdio>
#include <iostream>
#include <vector>
#include "mpcmp_motion_planner.hpp"

int main(int argc, char **argv) {
    const char *urdf = argc > 1 ? argv[1] : "";
    const char *out = argc > 2 ? argv[2] : "benchmark_data.txt";
    const int B = argc > 3 ? std::atoi(argv[3]) : 1000;            // benchmark.cpp:16
    try {
        MotionPlanner planner(urdf, 6, 2, B);
        planner.set_constraint_margins(0.8, 0.8, 0.6, 0.9, 0.1);     // benchmark.cpp:9
        std::vector<double> xf((size_t)B * 14);
        MotionPlanner::Vec7 q, v;
        for (int b = 0; b < B; b++) {
            planner.sample_random_state(q, v);
            mpcmp_shim::Mat<3, 1> lin, ang;
            for (int r = 0; r < 3; r++) { lin(r) = planner.random_unit() * planner.robot.max_linear_velocity; ang(r) = 0.0; }
            v = planner.robot.inverse_velocities(q, lin, ang);                                      // benchmark.cpp:20
            auto norm3 = [](const mpcmp_shim::Mat<6, 1> &t, int o) { return std::sqrt(t(o) * t(o) + t(o + 1) * t(o + 1) + t(o + 2) * t(o + 2)); };
            mpcmp_shim::Mat<6, 1> task = planner.robot.forward_velocities(q, v);
            if (norm3(task, 0) > planner.robot.max_linear_velocity) {                              // benchmark.cpp:25-31
                const double f = 0.9 * planner.robot.max_linear_velocity / norm3(task, 0);
                for (int j = 0; j < 7; j++) v(j) *= f;
                task = planner.robot.forward_velocities(q, v);
            }
            if (norm3(task, 3) > planner.robot.max_angular_velocity) {                             // benchmark.cpp:32-38
                const double f = 0.9 * planner.robot.max_angular_velocity / norm3(task, 3);
                for (int j = 0; j < 7; j++) v(j) *= f;
            }
            double worst = 0.0;                                                                     // benchmark.cpp:40-42
            for (int j = 0; j < 7; j++) worst = std::fmax(worst, std::fabs(v(j)) / (planner.margin_velocity_ * planner.robot.max_velocity(j)));
            if (worst > 1.0) for (int j = 0; j < 7; j++) v(j) /= 1.1 * worst;
            for (int j = 0; j < 7; j++) { xf[(size_t)b * 14 + j] = q(j); xf[(size_t)b * 14 + 7 + j] = v(j); }
        }
        planner.benchmark_batch(B, xf.data(), out);
        std::printf("%d trajectories -> %s\n", B, out);
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
