// Counterpart of the reference's examples/benchmark.cpp: 1000 targets from the default start state, margins
// (0.8, 0.8, 0.6, 0.9, 0.1), solved as ONE batch on the GPU; appends the 162-column rows that
// analysis/benchmark_analysis.ipynb reads.  (The reference derives the target velocity from a random task-space
// velocity through PandaWrapper::inverse_velocities; here targets come from sample_random_state with the same
// joint-velocity clamp — scenario generation is outside the hot path)
//   g++ -O2 -std=c++17 -Iinclude examples/benchmark.cpp -Lmpc_motion_planner_amd -lmpcmp -Wl,-rpath,$PWD/mpc_motion_planner_amd -o mpc_benchmark
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
