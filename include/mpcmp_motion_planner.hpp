// mpcmp_motion_planner.hpp — header-only C++ mirror of the reference's planner façade over the mpcmp C ABI.
//
// Same class name, member names, argument meaning and error behaviour as `MotionPlanner`
// (mpc_solver/motionPlanner.hpp:16-176, mpc_solver/motionPlanner.cpp) and the limits table of `PandaWrapper`
// (robot_utils/pandaWrapper.hpp:28-40), so that examples/offline_trajectory.cpp / examples/benchmark.cpp keep
// their call sequence; the solve runs on the GPU through libmpcmp.so.  Additions: `solve_batch` (B problems in
// one call — what examples/benchmark.cpp:16 loops over) and the text writers of the two result formats.
//
// With Eigen on the include path the vector/matrix arguments are Eigen types exactly as in the reference;
// without it (this image has no Eigen) a minimal column-major fixed-size matrix with the same element access
// stands in.  Differences that cannot be hidden: Ruckig is replaced by the library's own jerk-limited, time-synchronised
// generator (same problem statement; it reproduces the reference's stored Ruckig trajectory), `mpc` (the polympc object)
// does not exist.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "mpcmp.h"

#if defined(MPCMP_USE_EIGEN) || (defined(__has_include) && __has_include(<Eigen/Dense>))
#include <Eigen/Dense>
namespace mpcmp_shim {
template <int R, int C> using Mat = Eigen::Matrix<double, R, C>;
}
#else
namespace mpcmp_shim {
template <int R, int C>
struct Mat {                                  // column-major like Eigen's default
    std::array<double, (size_t)R * C> d{};
    double &operator()(int i, int j) { return d[(size_t)j * R + i]; }
    double operator()(int i, int j) const { return d[(size_t)j * R + i]; }
    double &operator()(int i) { return d[i]; }
    double operator()(int i) const { return d[i]; }
    double &operator[](int i) { return d[i]; }
    double operator[](int i) const { return d[i]; }
    double *data() { return d.data(); }
    const double *data() const { return d.data(); }
    static Mat Zero() { return Mat(); }
    static constexpr int rows() { return R; }
    static constexpr int cols() { return C; }
};
}  // namespace mpcmp_shim
#endif

#define NDOF 7   // robot_utils/pandaWrapper.hpp:10

// Minimal stand-ins for the Pinocchio objects a reference caller touches through `planner.robot`
// (examples/benchmark.cpp:108-110,154-156: forwardKinematics(model, data, q); updateFramePlacement(model, data, frame_id);
//  data.oMf[frame_id].translation()[2]).  Only what those call sites use exists.
namespace mpcmp_shim {
struct Placement {
    double t[3] = {0, 0, 0};
    double Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double *translation() const { return t; }
    const double *rotation() const { return Rm; }          // row-major
};
struct FrameData {
    static constexpr int JOINT7 = 7, LINK8 = 8, TOOL = 9;  // oMf indices: joint-7 origin, panda_link8, panda_tool
    Placement oMf[10];
    Placement oMi[8];                                       // oMi[7]: joint-7 placement (motionPlanner.cpp:111)
};
}  // namespace mpcmp_shim
#if !(defined(__has_include) && __has_include(<pinocchio/algorithm/kinematics.hpp>))
namespace pinocchio {
// joint placements of the chain for configuration q[7] (pinocchio::forwardKinematics)
inline void forwardKinematicsArray(const mpcmp_model &model, mpcmp_shim::FrameData &data, const double *q) {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
    for (int i = 0; i < 7; i++) {
        for (int r = 0; r < 3; r++) p[r] += R[3 * r] * model.p[i][0] + R[3 * r + 1] * model.p[i][1] + R[3 * r + 2] * model.p[i][2];
        const double c = std::cos(q[i]), s = std::sin(q[i]);
        double J[9], Rn[9];
        for (int r = 0; r < 3; r++) {
            J[3 * r] = model.R0[i][3 * r] * c + model.R0[i][3 * r + 1] * s;
            J[3 * r + 1] = -model.R0[i][3 * r] * s + model.R0[i][3 * r + 1] * c;
            J[3 * r + 2] = model.R0[i][3 * r + 2];
        }
        for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++)
            Rn[3 * r + k] = R[3 * r] * J[k] + R[3 * r + 1] * J[3 + k] + R[3 * r + 2] * J[6 + k];
        for (int k = 0; k < 9; k++) R[k] = Rn[k];
        for (int k = 0; k < 3; k++) data.oMi[i + 1].t[k] = p[k];
        for (int k = 0; k < 9; k++) data.oMi[i + 1].Rm[k] = R[k];
    }
}
template <class V> inline void forwardKinematics(const mpcmp_model &model, mpcmp_shim::FrameData &data, const V &q) {
    double qq[7];
    for (int j = 0; j < 7; j++) qq[j] = q(j);
    forwardKinematicsArray(model, data, qq);
}
// placement of one operational frame from the joint placements (pinocchio::updateFramePlacement)
inline void updateFramePlacement(const mpcmp_model &model, mpcmp_shim::FrameData &data, int frame_id) {
    const mpcmp_shim::Placement &j7 = data.oMi[7];
    const double *off = frame_id == mpcmp_shim::FrameData::TOOL ? model.tool : (frame_id == mpcmp_shim::FrameData::LINK8 ? model.link8 : nullptr);
    mpcmp_shim::Placement &f = data.oMf[frame_id];
    for (int k = 0; k < 9; k++) f.Rm[k] = j7.Rm[k];
    for (int r = 0; r < 3; r++)
        f.t[r] = j7.t[r] + (off ? j7.Rm[3 * r] * off[0] + j7.Rm[3 * r + 1] * off[1] + j7.Rm[3 * r + 2] * off[2] : 0.0);
}
}  // namespace pinocchio
#endif

// limits table + model of robot_utils/pandaWrapper.hpp (Pinocchio members replaced by mpcmp_model)
class PandaWrapper {
  public:
    using Vec7 = mpcmp_shim::Mat<NDOF, 1>;
    mpcmp_model model;
    mpcmp_shim::FrameData data;                                   // robot_utils/pandaWrapper.hpp:19 (pinocchio::Data)
    int frame_id = mpcmp_shim::FrameData::TOOL;                   // getFrameId("panda_tool"), pandaWrapper.cpp:11
    Vec7 min_position, max_position, max_velocity, max_acceleration, max_jerk, max_torque;
    double max_torqueDot{1000};
    double max_linear_velocity{1.7};
    double max_angular_velocity{2.5};
    double min_height{0.05};
    explicit PandaWrapper(const std::string &urdf_path) {
        const int rc = urdf_path.empty() ? mpcmp_default_model(&model) : mpcmp_model_from_urdf(urdf_path.c_str(), &model);
        if (rc) throw std::runtime_error(std::string("PandaWrapper: ") + mpcmp_last_error(nullptr));
        mpcmp_default_limits(min_position.data(), max_position.data(), max_velocity.data(), max_acceleration.data(),
                             max_jerk.data(), max_torque.data());
    }
    // world position of the joint-7 origin (oMi[7]) and of the tool frame
    void forward_kinematics(const double *q, double *p_joint7, double *p_tool) const {
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
        for (int i = 0; i < 7; i++) {
            for (int r = 0; r < 3; r++) p[r] += R[3 * r] * model.p[i][0] + R[3 * r + 1] * model.p[i][1] + R[3 * r + 2] * model.p[i][2];
            const double c = std::cos(q[i]), s = std::sin(q[i]);
            double J[9], Rn[9];
            for (int r = 0; r < 3; r++) {
                J[3 * r] = model.R0[i][3 * r] * c + model.R0[i][3 * r + 1] * s;
                J[3 * r + 1] = -model.R0[i][3 * r] * s + model.R0[i][3 * r + 1] * c;
                J[3 * r + 2] = model.R0[i][3 * r + 2];
            }
            for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++)
                Rn[3 * r + k] = R[3 * r] * J[k] + R[3 * r + 1] * J[3 + k] + R[3 * r + 2] * J[6 + k];
            for (int k = 0; k < 9; k++) R[k] = Rn[k];
        }
        if (p_joint7) for (int r = 0; r < 3; r++) p_joint7[r] = p[r];
        if (p_tool) for (int r = 0; r < 3; r++)
            p_tool[r] = p[r] + R[3 * r] * model.tool[0] + R[3 * r + 1] * model.tool[1] + R[3 * r + 2] * model.tool[2];
    }
    // robot_utils/pandaWrapper.cpp:62-88: joint velocity realising a task velocity (damped pseudo-inverse)
    Vec7 inverse_velocities(const Vec7 &q, const mpcmp_shim::Mat<3, 1> &linear_velocity, const mpcmp_shim::Mat<3, 1> &angular_velocity) const {
        Vec7 qd;
        if (mpcmp_inverse_velocities(&model, q.data(), linear_velocity.data(), angular_velocity.data(), qd.data()))
            throw std::runtime_error("PandaWrapper::inverse_velocities failed");
        return qd;
    }
    // robot_utils/pandaWrapper.cpp:90-107: [linear; angular] velocity of the tool frame, world-aligned
    mpcmp_shim::Mat<6, 1> forward_velocities(const Vec7 &q, const Vec7 &qdot) const {
        mpcmp_shim::Mat<6, 1> out;
        if (mpcmp_forward_velocities(&model, q.data(), qdot.data(), out.data())) throw std::runtime_error("PandaWrapper::forward_velocities failed");
        return out;
    }
    // robot_utils/pandaWrapper.cpp:14-60.  The reference starts from a random configuration and drops the success flag;
    // here the start is explicit (default: zeros) and `converged`, when given, receives the flag.
    Vec7 inverse_kinematic(const mpcmp_shim::Mat<3, 3> &orientation, const mpcmp_shim::Mat<3, 1> &position, const Vec7 *q_init = nullptr,
                           bool *converged = nullptr) const {
        double R[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[3 * r + c] = orientation(r, c);
        Vec7 q;
        const int rc = mpcmp_inverse_kinematics(&model, R, position.data(), q_init ? q_init->data() : nullptr, q.data(), nullptr);
        if (rc != 0 && rc != 1) throw std::runtime_error("PandaWrapper::inverse_kinematic failed");
        if (converged) *converged = rc == 0;
        return q;
    }
};

class MotionPlanner;
namespace mpcmp_shim {
// What a reference caller can reach through the public member `mpc` (motionPlanner.hpp:28-29: polympc's MPC<> object),
// reduced to the calls the facade itself makes (motionPlanner.cpp:15-20,33,47,72-97,172-174,184-207; motionPlanner.hpp:106-125):
// settings, guesses, bounds, solve, the solution and its interpolation, and the solver info.  The polympc object itself does
// not exist here: the setters write the planner's configuration record and guess, which the next solve hands to the library.
struct MpcView {
    struct Settings { int &max_iter; int &line_search_max_iter; };
    struct QpSettings { int &max_iter; double &eps_rel; double &eps_abs; };
    struct Status { int value; };
    struct InfoView { Status status; int iter; int qp_iters_total; };
    explicit MpcView(MotionPlanner *o) : owner(o) {}
    Settings settings();
    QpSettings qp_settings();
    std::array<double, 1> solution_p() const;
    const std::vector<double> &solution_x() const;
    const std::vector<double> &solution_u() const;
    Mat<2 * NDOF, 1> solution_x_at(double t) const;       // normalised time in [0,1]
    Mat<NDOF, 1> solution_u_at(double t) const;
    InfoView info() const;
    std::vector<double> time_nodes() const;                // mpc.ocp().time_nodes
    // ---- write side ----
    // guesses in the layout of solution_x() / solution_u(): node-major, node 0 = the current state (motionPlanner.cpp:172-174,205-207)
    void x_guess(const std::vector<double> &x);            // [14 N]
    void u_guess(const std::vector<double> &u);            // [7 N]
    void p_guess(const std::array<double, 1> &p);          // final time
    void p_guess(double T) { p_guess(std::array<double, 1>{T}); }
    // bounds (motionPlanner.cpp:72,75,80,97); they are overwritten by the next set_constraint_margins(), as in the reference
    void state_bounds(const Mat<2 * NDOF, 1> &lb, const Mat<2 * NDOF, 1> &ub);
    void control_bounds(const Mat<NDOF, 1> &lb, const Mat<NDOF, 1> &ub);
    void parameters_bounds(const std::array<double, 1> &lb, const std::array<double, 1> &ub);
    void constraints_bounds(const Mat<NDOF + 1, 1> &lb, const Mat<NDOF + 1, 1> &ub);      // 7 torques, tool height
    // motionPlanner.cpp:47 / :33: the solver pins the first node to the centre of the box and keeps the last node within its
    // half-width (one half-width for all 14 entries: the largest given)
    void initial_state_bounds(const Mat<2 * NDOF, 1> &lb, const Mat<2 * NDOF, 1> &ub);
    void final_state_bounds(const Mat<2 * NDOF, 1> &lb, const Mat<2 * NDOF, 1> &ub);
    void solve();                                          // motionPlanner.cpp:184, from the guess set above
  private:
    MotionPlanner *owner;
};

// Look-alikes of the three Ruckig members (motionPlanner.hpp:34-37) for callers that touch them directly: the input record
// (ruckig::InputParameter's field names), the trajectory (get_duration / at_time) and `otg.calculate(input, trajectory)`.
// Behind them is the library's jerk-limited, time-synchronised generator (mpcmp_jerk_point_lim_batch); every field of the input record is used.
enum Result { Working = 0, Finished = 1, Error = -1 };     // ruckig::Result, the values motionPlanner.cpp:149 can see
struct InputParameter {
    std::array<double, NDOF> current_position{}, current_velocity{}, current_acceleration{};
    std::array<double, NDOF> target_position{}, target_velocity{}, target_acceleration{};
    std::array<double, NDOF> max_velocity{}, max_acceleration{}, max_jerk{};
};
struct Trajectory {
    double get_duration() const { return duration; }
    // position, velocity, acceleration at min(time, duration) (ruckig::Trajectory::at_time, motionPlanner.cpp:160)
    void at_time(double time, std::array<double, NDOF> &position, std::array<double, NDOF> &velocity, std::array<double, NDOF> &acceleration) const;
  private:
    friend struct Otg;
    MotionPlanner *owner = nullptr;
    InputParameter in;
    double duration = -1.0;
};
struct Otg {
    explicit Otg(MotionPlanner *o) : owner(o) {}
    Result calculate(const InputParameter &input, Trajectory &trajectory);    // motionPlanner.cpp:149
  private:
    MotionPlanner *owner;
};
}  // namespace mpcmp_shim

class MotionPlanner {
  public:
    using Vec7 = mpcmp_shim::Mat<NDOF, 1>;
    using Vec14 = mpcmp_shim::Mat<2 * NDOF, 1>;
    using mpc_t = mpcmp_shim::MpcView;
    mpc_t mpc{this};                                                     // motionPlanner.hpp:28-29 (view, see MpcView)

    PandaWrapper robot;
    mpcmp_shim::Otg otg{this};                                           // motionPlanner.hpp:35-37 (look-alikes, see above)
    mpcmp_shim::Trajectory trajectory;
    mpcmp_shim::InputParameter input;
    Vec14 current_state, target_state;
    const double eps = 1e-2;                                            // motionPlanner.hpp:44
    const double inf = std::numeric_limits<double>::infinity();
    double margin_position_, margin_velocity_, margin_acceleration_, margin_torque_, margin_jerk_;
    mpcmp_config config;                                                 // stands in for mpc.settings()/qp_settings()
    mpcmp_info last_info{};

    // num_seg / sqp_iters default to the reference as shipped (robot_ocp.hpp:32, motionPlanner.cpp:15)
    explicit MotionPlanner(std::string urdf_path, int num_seg = 6, int sqp_iters = 2, int max_batch = 1024, int device = 0)
        : robot(urdf_path), max_batch_(max_batch) {
        mpcmp_default_config(&config, num_seg, sqp_iters);               // motionPlanner.cpp:15-20
        N_ = mpcmp_num_nodes(num_seg);
        const int rc = mpcmp_create(&config, &robot.model, device, max_batch, &ctx_);
        if (rc) throw std::runtime_error(std::string("MotionPlanner: ") + mpcmp_last_error(nullptr));
        Vec7 mid, zero = Vec7::Zero();
        for (int j = 0; j < 7; j++) mid(j) = 0.5 * (robot.max_position(j) + robot.min_position(j));  // motionPlanner.cpp:5
        set_target_state(mid, zero);
        set_current_state(mid, zero);
        set_constraint_margins(1.0, 1.0, 1.0, 1.0, 1.0);                 // motionPlanner.cpp:24
        sol_x_.assign((size_t)14 * N_, 0.0); sol_u_.assign((size_t)7 * N_, 0.0);
    }
    ~MotionPlanner() { mpcmp_destroy(ctx_); }
    MotionPlanner(const MotionPlanner &) = delete;
    MotionPlanner &operator=(const MotionPlanner &) = delete;

    void set_target_state(Vec7 target_position, Vec7 target_velocity, Vec7 target_acceleration = Vec7::Zero()) {     // motionPlanner.cpp:27-39
        for (int j = 0; j < 7; j++) {
            target_state(j) = target_position(j); target_state(7 + j) = target_velocity(j);
            input.target_position[j] = target_position(j); input.target_velocity[j] = target_velocity(j); input.target_acceleration[j] = target_acceleration(j);
        }
    }
    void set_current_state(Vec7 current_position, Vec7 current_velocity, Vec7 current_acceleration = Vec7::Zero()) {  // motionPlanner.cpp:41-54
        for (int j = 0; j < 7; j++) {
            current_state(j) = current_position(j); current_state(7 + j) = current_velocity(j);
            input.current_position[j] = current_position(j); input.current_velocity[j] = current_velocity(j); input.current_acceleration[j] = current_acceleration(j);
        }
    }
    void set_constraint_margins(double margin_position, double margin_velocity, double margin_acceleration,
                                double margin_torque, double margin_jerk) {                          // motionPlanner.cpp:56-90
        margin_position_ = margin_position; margin_velocity_ = margin_velocity; margin_acceleration_ = margin_acceleration;
        margin_torque_ = margin_torque; margin_jerk_ = margin_jerk;
        mpcmp_set_margins(&config, margin_position, margin_velocity, margin_acceleration, margin_torque);
        mpcmp_set_min_height(&config, robot.min_height);
        push_config();
        for (int j = 0; j < 7; j++) {                                    // motionPlanner.cpp:86-88
            input.max_velocity[j] = margin_velocity * robot.max_velocity(j);
            input.max_acceleration[j] = margin_acceleration * robot.max_acceleration(j);
            input.max_jerk[j] = margin_jerk * robot.max_jerk(j);
        }
    }
    void set_min_height(double min_height) { mpcmp_set_min_height(&config, min_height); push_config(); }  // :92-100

    // one draw in [-1,1) of the planner's seeded stream: what `Matrix<...>::Random()` coefficients are in the reference's examples
    double random_unit() { return uniform(); }
    // motionPlanner.cpp:102-114 (Eigen::Random replaced by a SplitMix64 stream owned by the planner)
    void sample_random_state(Vec7 &random_position, Vec7 &random_velocity) {
        double p7[3];
        do {
            for (int j = 0; j < 7; j++) {
                const double s = (1 - margin_position_) * (robot.max_position(j) - robot.min_position(j)) / 2;
                random_position(j) = 0.5 * (uniform() * (robot.max_position(j) - robot.min_position(j) - 2 * s) +
                                            (robot.max_position(j) + robot.min_position(j)));
            }
            robot.forward_kinematics(random_position.data(), p7, nullptr);
        } while (p7[2] < robot.min_height);
        for (int j = 0; j < 7; j++) random_velocity(j) = margin_velocity_ * uniform() * robot.max_velocity(j);
    }
    void seed(uint64_t s) { rng_ = s; }

    // motionPlanner.cpp:116-144
    int check_state_in_bounds(Vec7 &position, Vec7 &velocity, Vec7 acceleration = Vec7::Zero()) {
        bool pc = false, vc = false, ac = false;
        for (int j = 0; j < 7; j++) {
            const double s = (1 - margin_position_) * (robot.max_position(j) - robot.min_position(j)) / 2;
            pc |= position(j) > robot.max_position(j) - s || position(j) < robot.min_position(j) + s;
            vc |= std::fabs(velocity(j)) > margin_velocity_ * robot.max_velocity(j);
            ac |= std::fabs(acceleration(j)) > margin_acceleration_ * robot.max_acceleration(j);
        }
        int flag = 0;
        if (pc && !vc) flag = 1;
        if (!pc && vc) flag = 2;
        if (pc && vc) flag = 3;
        if (ac) flag += 10;
        return flag;
    }

    // motionPlanner.hpp:145-172: regularly time-spaced trajectory, nearest-sample pick round(t*(nPoint-1))
    void warm_start(double final_time, const std::vector<double> &position_trajectory /*7 x nPoint col-major*/,
                    const std::vector<double> &velocity_trajectory, const std::vector<double> &acceleration_trajectory) {
        const int nPoint = (int)(position_trajectory.size() / 7);
        std::vector<double> tau(N_);
        mpcmp_time_nodes(config.num_seg, tau.data());
        warm_x_.assign((size_t)14 * N_, 0.0); warm_u_.assign((size_t)7 * N_, 0.0);
        for (int i = 0; i < N_; i++) {
            const int idx = (int)std::lround(tau[i] * (nPoint - 1));
            for (int j = 0; j < 7; j++) {
                warm_x_[14 * i + j] = position_trajectory[7 * idx + j];
                warm_x_[14 * i + 7 + j] = velocity_trajectory[7 * idx + j];
                warm_u_[7 * i + j] = acceleration_trajectory[7 * idx + j];
            }
        }
        warm_T_ = final_time; have_warm_ = true;
    }

#if defined(MPCMP_USE_EIGEN) || (defined(__has_include) && __has_include(<Eigen/Dense>))
    // the reference's own signature (motionPlanner.hpp:145): 7 x nPoint matrices, regularly time spaced
    void warm_start(double final_time, Eigen::MatrixXd position_trajectory, Eigen::MatrixXd velocity_trajectory,
                    Eigen::MatrixXd acceleration_trajectory) {
        const size_t cnt = (size_t)position_trajectory.size();
        warm_start(final_time, std::vector<double>(position_trajectory.data(), position_trajectory.data() + cnt),
                   std::vector<double>(velocity_trajectory.data(), velocity_trajectory.data() + cnt),
                   std::vector<double>(acceleration_trajectory.data(), acceleration_trajectory.data() + cnt));
    }
#endif

    // motionPlanner.cpp:177-208.  true: jerk-limited, time-synchronised trajectory between the two states (what the
    // reference gets from Ruckig, motionPlanner.cpp:146-175); false: previous solution / warm_start() guess
    void solve_trajectory(bool use_ruckig_as_warm_start) {
        const bool use_guess = !use_ruckig_as_warm_start && have_warm_;
        if (!use_guess) {
            guess_x_.assign((size_t)14 * N_, 0.0); guess_u_.assign((size_t)7 * N_, 0.0);
            // the whole of `input`, as warm_start_RK hands it to otg.calculate (motionPlanner.cpp:146-149): states and boundary accelerations
            // (written by set_current_state / set_target_state, motionPlanner.cpp:36-38,50-52, or by the caller), velocity / acceleration / jerk limits
            // (set_constraint_margins, motionPlanner.cpp:86-88, or the caller)
            input_states(guess_x0_, guess_xf_);
            for (int j = 0; j < 7; j++) { guess_a0_[j] = input.current_acceleration[j]; guess_aT_[j] = input.target_acceleration[j]; }
            guess_in_ = input;
            chk(mpcmp_warm_start_jerk_lim_batch(ctx_, 1, guess_x0_, guess_xf_, guess_a0_, guess_aT_, input.max_velocity.data(), input.max_acceleration.data(),
                                                input.max_jerk.data(), guess_x_.data(), guess_u_.data(), &guess_T_));
            guess_is_profile_ = true;
        } else { guess_x_ = warm_x_; guess_u_ = warm_u_; guess_T_ = warm_T_; guess_is_profile_ = false; }
        solve_from_guess();
    }

    // B independent problems in one call (the loop of examples/benchmark.cpp:16). x0/xf: [B][14]
    void solve_batch(int B, const double *x0, const double *xf, double *sol_x, double *sol_u, double *sol_T, mpcmp_info *info) {
        chk(mpcmp_solve_batch(ctx_, B, x0, xf, nullptr, nullptr, nullptr, sol_x, sol_u, sol_T, info));
    }

    // motionPlanner.hpp:99-116
    template <const int NP>
    void get_MPC_trajectory(mpcmp_shim::Mat<1, NP + 1> &time, mpcmp_shim::Mat<7, NP + 1> &position_trajectory,
                            mpcmp_shim::Mat<7, NP + 1> &velocity_trajectory, mpcmp_shim::Mat<7, NP + 1> &acceleration_trajectory,
                            mpcmp_shim::Mat<7, NP + 1> &torque_trajectory) {
        std::vector<double> out((size_t)(NP + 1) * 29);
        chk(mpcmp_sample_batch(ctx_, 1, sol_x_.data(), sol_u_.data(), &sol_T_, NP, out.data()));
        unpack<NP>(out, time, position_trajectory, velocity_trajectory, acceleration_trajectory, torque_trajectory);
    }
    // motionPlanner.hpp:73-96 — the jerk-limited trajectory of the last solve_trajectory(true) sampled uniformly, torques by
    // RNEA (motionPlanner.hpp:92); after a solve from another guess: that guess, resampled like the MPC solution
    template <const int NP>
    void get_ruckig_trajectory(mpcmp_shim::Mat<1, NP + 1> &time, mpcmp_shim::Mat<7, NP + 1> &position_trajectory,
                               mpcmp_shim::Mat<7, NP + 1> &velocity_trajectory, mpcmp_shim::Mat<7, NP + 1> &acceleration_trajectory,
                               mpcmp_shim::Mat<7, NP + 1> &torque_trajectory) {
        std::vector<double> out((size_t)(NP + 1) * 29);
        if (guess_is_profile_) {
            std::vector<double> tr((size_t)(NP + 1) * 22), q((size_t)(NP + 1) * 7), v(q.size()), a(q.size()), tau(q.size());
            chk(mpcmp_jerk_trajectory_lim_batch(ctx_, 1, guess_x0_, guess_xf_, guess_a0_, guess_aT_, guess_in_.max_velocity.data(), guess_in_.max_acceleration.data(),
                                                guess_in_.max_jerk.data(), NP, tr.data(), nullptr));
            for (int i = 0; i <= NP; i++)
                for (int j = 0; j < 7; j++) { q[(size_t)i * 7 + j] = tr[(size_t)i * 22 + 1 + j]; v[(size_t)i * 7 + j] = tr[(size_t)i * 22 + 8 + j]; a[(size_t)i * 7 + j] = tr[(size_t)i * 22 + 15 + j]; }
            chk(mpcmp_rnea_batch(ctx_, NP + 1, q.data(), v.data(), a.data(), tau.data()));
            for (int i = 0; i <= NP; i++) {
                out[(size_t)i * 29] = tr[(size_t)i * 22];
                for (int j = 0; j < 7; j++) {
                    out[(size_t)i * 29 + 1 + j] = q[(size_t)i * 7 + j]; out[(size_t)i * 29 + 8 + j] = v[(size_t)i * 7 + j];
                    out[(size_t)i * 29 + 15 + j] = a[(size_t)i * 7 + j]; out[(size_t)i * 29 + 22 + j] = tau[(size_t)i * 7 + j];
                }
            }
        } else {
            chk(mpcmp_sample_batch(ctx_, 1, guess_x_.data(), guess_u_.data(), &guess_T_, NP, out.data()));
        }
        unpack<NP>(out, time, position_trajectory, velocity_trajectory, acceleration_trajectory, torque_trajectory);
    }
    // motionPlanner.hpp:118-128, including its clamp: for time >= T the normalised time is set to T (not 1)
    void get_MPC_point(double time, Vec7 &position, Vec7 &velocity, Vec7 &acceleration, Vec7 &torque) {
        double o[28];
        chk(mpcmp_mpc_point_batch(ctx_, 1, sol_x_.data(), sol_u_.data(), &sol_T_, &time, o));       // one launch: clamp, interpolation, RNEA
        for (int j = 0; j < 7; j++) { position(j) = o[j]; velocity(j) = o[7 + j]; acceleration(j) = o[14 + j]; torque(j) = o[21 + j]; }
    }
    // motionPlanner.hpp:130-142: the jerk-limited (Ruckig stand-in) trajectory of the last solve_trajectory(true) at
    // min(time, duration), torque by RNEA
    void get_RK_point(double time, Vec7 &position, Vec7 &velocity, Vec7 &acceleration, Vec7 &torque) {
        double o[28];
        chk(mpcmp_jerk_point_lim_batch(ctx_, 1, guess_x0_, guess_xf_, guess_a0_, guess_aT_, guess_in_.max_velocity.data(), guess_in_.max_acceleration.data(),
                                       guess_in_.max_jerk.data(), &time, o, nullptr));
        for (int j = 0; j < 7; j++) { position(j) = o[j]; velocity(j) = o[7 + j]; acceleration(j) = o[14 + j]; torque(j) = o[21 + j]; }
    }
    double solution_T() const { return sol_T_; }
    double guess_T() const { return guess_T_; }      // duration of the guess the last solve started from (p_guess, motionPlanner.cpp:173)
    const std::vector<double> &solution_x() const { return sol_x_; }
    const std::vector<double> &solution_u() const { return sol_u_; }
    int num_nodes() const { return N_; }
    mpcmp_ctx *context() { return ctx_; }

    // analysis/optimal_solution.txt: row 0 target, rows 1..NP+1 initial guess ("Ruckig"), then MPC; 29 columns
    // (examples/offline_trajectory.cpp:69-105)
    template <const int NP>
    void write_optimal_solution(const std::string &path) {
        mpcmp_shim::Mat<1, NP + 1> t; mpcmp_shim::Mat<7, NP + 1> q, v, a, tau;
        FILE *f = std::fopen(path.c_str(), "w");
        if (!f) throw std::runtime_error("cannot open " + path);
        std::fprintf(f, "0");
        for (int j = 0; j < 14; j++) std::fprintf(f, " %.6g", target_state(j));
        for (int j = 0; j < 14; j++) std::fprintf(f, " 0");
        std::fprintf(f, "\n");
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 0) get_ruckig_trajectory<NP>(t, q, v, a, tau); else get_MPC_trajectory<NP>(t, q, v, a, tau);
            for (int i = 0; i <= NP; i++) {
                std::fprintf(f, "%.6g", t(0, i));
                for (int j = 0; j < 7; j++) std::fprintf(f, " %.6g", q(j, i));
                for (int j = 0; j < 7; j++) std::fprintf(f, " %.6g", v(j, i));
                for (int j = 0; j < 7; j++) std::fprintf(f, " %.6g", a(j, i));
                for (int j = 0; j < 7; j++) std::fprintf(f, " %.6g", tau(j, i));
                std::fprintf(f, "\n");
            }
        }
        std::fclose(f);
    }

    // The 1000-iteration loop of examples/benchmark.cpp as ONE batched call: targets xf [B][14] from the current state,
    // then one 162-number row per problem appended to `path` in the reference's layout (benchmark.cpp:164-194):
    //   guess min(28) max(28) | MPC min(28) max(28) | guess terminal error(14) | MPC terminal error(14) |
    //   guess flags(4: jerk, linear vel, angular vel, collision) | MPC flags(4) | target(14)
    // ("guess" = the jerk-limited trajectory standing in for Ruckig, in the node form handed to the solver).
    void benchmark_batch(int B, const double *xf, const std::string &path, int n_pts = 200) {
        const size_t N = (size_t)N_;
        std::vector<double> x0((size_t)B * 14), gx(B * 14 * N), gu(B * 7 * N), gT(B), sx(B * 14 * N), su(B * 7 * N), sT(B);
        std::vector<double> sg((size_t)B * 74), sm((size_t)B * 74);
        std::vector<mpcmp_info> info(B);
        for (int b = 0; b < B; b++) for (int r = 0; r < 14; r++) x0[(size_t)b * 14 + r] = current_state(r);
        chk(mpcmp_warm_start_jerk_lim_batch(ctx_, B, x0.data(), xf, nullptr, nullptr, input.max_velocity.data(), input.max_acceleration.data(), input.max_jerk.data(),
                                            gx.data(), gu.data(), gT.data()));     // benchmark.cpp:46-47
        chk(mpcmp_solve_batch(ctx_, B, x0.data(), xf, gx.data(), gu.data(), gT.data(), sx.data(), su.data(), sT.data(), info.data()));
        chk(mpcmp_traj_stats_batch(ctx_, B, gx.data(), gu.data(), gT.data(), xf, n_pts, sg.data()));
        chk(mpcmp_traj_stats_batch(ctx_, B, sx.data(), su.data(), sT.data(), xf, n_pts, sm.data()));
        FILE *f = std::fopen(path.c_str(), "a");
        if (!f) throw std::runtime_error("cannot open " + path);
        for (int b = 0; b < B; b++) {
            const double *g = sg.data() + (size_t)b * 74, *m = sm.data() + (size_t)b * 74;
            for (int i = 0; i < 56; i++) std::fprintf(f, "%.6g ", g[i]);
            for (int i = 0; i < 56; i++) std::fprintf(f, "%.6g ", m[i]);
            for (int i = 56; i < 70; i++) std::fprintf(f, "%.6g ", g[i]);
            for (int i = 56; i < 70; i++) std::fprintf(f, "%.6g ", m[i]);
            for (int i = 70; i < 74; i++) std::fprintf(f, "%d ", (int)g[i]);
            for (int i = 70; i < 74; i++) std::fprintf(f, "%d ", (int)m[i]);
            for (int i = 0; i < 14; i++) std::fprintf(f, i == 13 ? "%.6g\n" : "%.6g ", xf[(size_t)b * 14 + i]);
        }
        std::fclose(f);
    }

  private:
    friend struct mpcmp_shim::MpcView;
    friend struct mpcmp_shim::Trajectory;
    friend struct mpcmp_shim::Otg;
    mpcmp_ctx *ctx_ = nullptr;
    int N_ = 0, max_batch_ = 0;
    std::vector<double> sol_x_, sol_u_, warm_x_, warm_u_, guess_x_, guess_u_;
    double sol_T_ = 0, warm_T_ = 0, guess_T_ = 0;
    bool guess_is_profile_ = false;                 // the last guess came from the jerk-limited generator
    double guess_x0_[14] = {0}, guess_xf_[14] = {0}, guess_a0_[7] = {0}, guess_aT_[7] = {0};      // states and boundary accelerations of the last jerk-limited guess
    mpcmp_shim::InputParameter guess_in_;          // `input` as it was when that guess was made (limits of get_ruckig_trajectory / get_RK_point)
    void input_states(double *x0, double *xf) const {
        for (int j = 0; j < 7; j++) { x0[j] = input.current_position[j]; x0[7 + j] = input.current_velocity[j]; xf[j] = input.target_position[j]; xf[7 + j] = input.target_velocity[j]; }
    }
    bool have_warm_ = false;
    uint64_t rng_ = 20240001ull;

    // mpc.solve() from (guess_x_, guess_u_, guess_T_), then "Fix initial and final point at correct place" (motionPlanner.cpp:184-207)
    void solve_from_guess() {
        chk(mpcmp_solve_batch(ctx_, 1, current_state.data(), target_state.data(), guess_x_.data(), guess_u_.data(), &guess_T_,
                              sol_x_.data(), sol_u_.data(), &sol_T_, &last_info));
        warm_x_ = sol_x_; warm_u_ = sol_u_; warm_T_ = sol_T_; have_warm_ = true;
        for (int r = 0; r < 14; r++) { warm_x_[r] = current_state(r); warm_x_[(size_t)14 * (N_ - 1) + r] = target_state(r); }
    }
    void chk(int rc) { if (rc) throw std::runtime_error(std::string("mpcmp: ") + mpcmp_last_error(ctx_)); }
    void push_config() { if (ctx_) chk(mpcmp_set_config(ctx_, &config)); }
    double uniform() {   // [-1,1), SplitMix64
        uint64_t z = (rng_ += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
    void interpolate(double t, Vec7 &q, Vec7 &v, Vec7 &a) const {   // mpc.solution_x_at / solution_u_at
        static const double xi[4] = {-1.0, -0.5, 0.5, 1.0};
        const int ns = config.num_seg;
        int s = (int)std::floor(t * ns); if (s >= ns) s = ns - 1; if (s < 0) s = 0;
        const double x = 2.0 * (t * ns - s) - 1.0;
        double L[4];
        for (int j = 0; j < 4; j++) { double w = 1; for (int k = 0; k < 4; k++) if (k != j) w *= (x - xi[k]) / (xi[j] - xi[k]); L[j] = w; }
        for (int r = 0; r < 7; r++) {
            q(r) = v(r) = a(r) = 0;
            for (int j = 0; j < 4; j++) {
                q(r) += L[j] * sol_x_[(size_t)14 * (3 * s + j) + r]; v(r) += L[j] * sol_x_[(size_t)14 * (3 * s + j) + 7 + r];
                a(r) += L[j] * sol_u_[(size_t)7 * (3 * s + j) + r];
            }
        }
    }
    template <const int NP>
    static void unpack(const std::vector<double> &out, mpcmp_shim::Mat<1, NP + 1> &time, mpcmp_shim::Mat<7, NP + 1> &q,
                       mpcmp_shim::Mat<7, NP + 1> &v, mpcmp_shim::Mat<7, NP + 1> &a, mpcmp_shim::Mat<7, NP + 1> &tau) {
        for (int i = 0; i <= NP; i++) {
            const double *o = out.data() + (size_t)i * 29;
            time(0, i) = o[0];
            for (int j = 0; j < 7; j++) { q(j, i) = o[1 + j]; v(j, i) = o[8 + j]; a(j, i) = o[15 + j]; tau(j, i) = o[22 + j]; }
        }
    }
};

// ---- MpcView members (need the complete MotionPlanner) ----
inline mpcmp_shim::MpcView::Settings mpcmp_shim::MpcView::settings() { return {owner->config.sqp_iters, owner->config.ls_iters}; }
inline mpcmp_shim::MpcView::QpSettings mpcmp_shim::MpcView::qp_settings() { return {owner->config.qp_iters, owner->config.eps_rel, owner->config.eps_abs}; }
inline std::array<double, 1> mpcmp_shim::MpcView::solution_p() const { return {owner->sol_T_}; }
inline const std::vector<double> &mpcmp_shim::MpcView::solution_x() const { return owner->sol_x_; }
inline const std::vector<double> &mpcmp_shim::MpcView::solution_u() const { return owner->sol_u_; }
inline mpcmp_shim::Mat<2 * NDOF, 1> mpcmp_shim::MpcView::solution_x_at(double t) const {
    MotionPlanner::Vec7 q, v, a; owner->interpolate(t, q, v, a);
    Mat<2 * NDOF, 1> x;
    for (int j = 0; j < 7; j++) { x(j) = q(j); x(7 + j) = v(j); }
    return x;
}
inline mpcmp_shim::Mat<NDOF, 1> mpcmp_shim::MpcView::solution_u_at(double t) const {
    MotionPlanner::Vec7 q, v, a; owner->interpolate(t, q, v, a);
    return a;
}
inline mpcmp_shim::MpcView::InfoView mpcmp_shim::MpcView::info() const {
    return {{owner->last_info.status}, owner->last_info.sqp_iters, owner->last_info.qp_iters_total};
}
inline std::vector<double> mpcmp_shim::MpcView::time_nodes() const {
    std::vector<double> t((size_t)owner->N_);
    mpcmp_time_nodes(owner->config.num_seg, t.data());
    return t;
}
// write side
inline void mpcmp_shim::MpcView::x_guess(const std::vector<double> &x) {
    if (x.size() != (size_t)14 * owner->N_) throw std::invalid_argument("mpc.x_guess: expected 14 * N entries");
    owner->warm_x_ = x; owner->have_warm_ = true;
    if (owner->warm_u_.size() != (size_t)7 * owner->N_) owner->warm_u_.assign((size_t)7 * owner->N_, 0.0);
}
inline void mpcmp_shim::MpcView::u_guess(const std::vector<double> &u) {
    if (u.size() != (size_t)7 * owner->N_) throw std::invalid_argument("mpc.u_guess: expected 7 * N entries");
    owner->warm_u_ = u;
    if (owner->warm_x_.size() != (size_t)14 * owner->N_) owner->warm_x_.assign((size_t)14 * owner->N_, 0.0);
}
inline void mpcmp_shim::MpcView::p_guess(const std::array<double, 1> &p) { owner->warm_T_ = p[0]; }
inline void mpcmp_shim::MpcView::state_bounds(const Mat<2 * NDOF, 1> &lb, const Mat<2 * NDOF, 1> &ub) {
    for (int r = 0; r < 14; r++) { owner->config.lbx[r] = lb(r); owner->config.ubx[r] = ub(r); }
    owner->push_config();
}
inline void mpcmp_shim::MpcView::control_bounds(const Mat<NDOF, 1> &lb, const Mat<NDOF, 1> &ub) {
    for (int r = 0; r < 7; r++) { owner->config.lbu[r] = lb(r); owner->config.ubu[r] = ub(r); }
    owner->push_config();
}
inline void mpcmp_shim::MpcView::parameters_bounds(const std::array<double, 1> &lb, const std::array<double, 1> &ub) {
    owner->config.lbT = lb[0]; owner->config.ubT = ub[0];
    owner->push_config();
}
inline void mpcmp_shim::MpcView::constraints_bounds(const Mat<NDOF + 1, 1> &lb, const Mat<NDOF + 1, 1> &ub) {
    for (int r = 0; r < 8; r++) { owner->config.lbg[r] = lb(r); owner->config.ubg[r] = ub(r); }
    owner->push_config();
}
inline void mpcmp_shim::MpcView::initial_state_bounds(const Mat<2 * NDOF, 1> &lb, const Mat<2 * NDOF, 1> &ub) {
    for (int r = 0; r < 14; r++) owner->current_state(r) = 0.5 * (lb(r) + ub(r));
}
inline void mpcmp_shim::MpcView::final_state_bounds(const Mat<2 * NDOF, 1> &lb, const Mat<2 * NDOF, 1> &ub) {
    double hw = 0.0;
    for (int r = 0; r < 14; r++) { owner->target_state(r) = 0.5 * (lb(r) + ub(r)); hw = std::fmax(hw, 0.5 * (ub(r) - lb(r))); }
    owner->config.eps_target = hw;
    owner->push_config();
}
inline void mpcmp_shim::MpcView::solve() {
    if (!owner->have_warm_) throw std::logic_error("mpc.solve: no guess (x_guess / u_guess / p_guess, warm_start or a previous solve)");
    owner->guess_x_ = owner->warm_x_; owner->guess_u_ = owner->warm_u_; owner->guess_T_ = owner->warm_T_; owner->guess_is_profile_ = false;
    owner->solve_from_guess();
}

// Ruckig look-alikes.  All of `input` is honoured (states, boundary accelerations, the three limit vectors); the limits travel as arguments
// (mpcmp_jerk_point_lim_batch), the planner's configuration is neither read nor touched.
inline mpcmp_shim::Result mpcmp_shim::Otg::calculate(const InputParameter &input, Trajectory &trajectory) {
    trajectory.owner = owner; trajectory.in = input;
    trajectory.duration = -1.0;
    double x0[14], xf[14], o[28], T = 0.0, t0 = 0.0;
    for (int j = 0; j < 7; j++) { x0[j] = input.current_position[j]; x0[7 + j] = input.current_velocity[j]; xf[j] = input.target_position[j]; xf[7 + j] = input.target_velocity[j]; }
    // a failure (e.g. a limit that is not positive: ruckig::Result::ErrorInvalidInput) is the return value, not an exception
    const int rc = mpcmp_jerk_point_lim_batch(owner->ctx_, 1, x0, xf, input.current_acceleration.data(), input.target_acceleration.data(), input.max_velocity.data(),
                                              input.max_acceleration.data(), input.max_jerk.data(), &t0, o, &T);
    if (rc != MPCMP_OK || !(T >= 0.0)) return Error;
    trajectory.duration = T;
    return Working;                                         // ruckig's offline calculate() returns Working on success
}
inline void mpcmp_shim::Trajectory::at_time(double time, std::array<double, NDOF> &position, std::array<double, NDOF> &velocity,
                                            std::array<double, NDOF> &acceleration) const {
    if (!owner || duration < 0.0) throw std::logic_error("Trajectory::at_time without a successful otg.calculate");
    double x0[14], xf[14], o[28], T = 0.0;
    for (int j = 0; j < 7; j++) { x0[j] = in.current_position[j]; x0[7 + j] = in.current_velocity[j]; xf[j] = in.target_position[j]; xf[7 + j] = in.target_velocity[j]; }
    owner->chk(mpcmp_jerk_point_lim_batch(owner->ctx_, 1, x0, xf, in.current_acceleration.data(), in.target_acceleration.data(), in.max_velocity.data(),
                                          in.max_acceleration.data(), in.max_jerk.data(), &time, o, &T));
    for (int j = 0; j < 7; j++) { position[j] = o[j]; velocity[j] = o[7 + j]; acceleration[j] = o[14 + j]; }
}
