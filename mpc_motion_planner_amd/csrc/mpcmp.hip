// mpcmp.hip — C-ABI entry points (include/mpcmp.h) of the MI355X-native batched min-time MPC solver.
// Host logic only orchestrates HIP launches; there is NO CPU fallback: without a HIP device every compute
// entry point fails with MPCMP_ENODEVICE.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include <algorithm>
#include <array>
#include <functional>
#include <map>

#include "../../include/mpcmp.h"
#include "solver_kernels.hpp"
#include "qp_kernel_v2.hpp"
#include "qp_kernel_v3.hpp"
#ifdef MPCMP_WITH_QP4      /* the measured two-OCPs-per-CU alternative for N = 13 (does not pay): tools/experiments/, built only on request (-DMPCMP_WITH_QP4 -Itools/experiments) */
#include "qp_kernel_v4.hpp"
#endif
#include "qp_kernel_v5.hpp"
#include "multi_kernels.hpp"
#include "kinematics_host.hpp"
#include "jerk_device.hpp"

#ifdef MPCMP_SPLIT_N25
// the N = 25 kernels live in qp3_n25.hip (built with another scheduler strategy: see there)
namespace mpcmp {
extern template __global__ void k_qp3f<8, 1>(mpcmp_config, WS, const Qp3Pat *, Xch, int, double *);
extern template __global__ void k_qp3f<8, 2>(mpcmp_config, WS, const Qp3Pat *, Xch, int, double *);
extern template __global__ void k_qp3<8, 1>(mpcmp_config, WS, const Qp3Pat *, Xch, int, const double *);
extern template __global__ void k_qp3<8, 2>(mpcmp_config, WS, const Qp3Pat *, Xch, int, const double *);
}  // namespace mpcmp
#endif

#ifdef MPCMP_SPLIT_N19
// the N = 19 one-arm pair lives in qp5_n19.hip (its own scheduler strategy: see there)
namespace mpcmp {
extern template __global__ void k_qp3f<6, 1, 5>(mpcmp_config, WS, const Qp3Pat *, Xch, int, double *);
extern template __global__ void k_qp5<6>(mpcmp_config, WS, const Qp3Pat *, Xch, int, const double *);
}  // namespace mpcmp
#endif

using namespace mpcmp;

struct mpcmp_ctx {
    mpcmp_config cfg;
    int device = 0, max_batch = 0, nseg = 0, narm = 1;
    int N = 0, n = 0, meq = 0, m = 0, mn = 0;      // of the whole OCP (all arms): n = 21 N narm + 1, meq = 14 (N-1) narm, m = meq + 8 N narm
    int nx = 14, nu = 7;                           // state / control entries per node: 14 narm, 7 narm
    std::string err;
    hipStream_t stream = nullptr;
    hipStream_t stream_x[3] = {nullptr, nullptr, nullptr};   // further parts of a large solve (see solve_impl)
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    // device buffers
    mpcmp_model *d_model = nullptr;   // [narm]
    mpcmp_model model;             // host copy: passed by value to the kernels that run the rigid-body recursions
    int *d_ext_of_int = nullptr, *d_entry_ptr = nullptr;
    uint32_t *d_terms = nullptr;
    WS ws{};
    std::vector<void *> allocs;
    // host-API staging (device side)
    double *d_x0 = nullptr, *d_xf = nullptr, *d_wx = nullptr, *d_wu = nullptr, *d_wT = nullptr;
    double *d_sx = nullptr, *d_su = nullptr, *d_sT = nullptr;
    double *d_ax0 = nullptr, *d_axf = nullptr, *d_awx = nullptr, *d_awu = nullptr, *d_awT = nullptr;   // multi-arm warm start staging
    mpcmp_info *d_info = nullptr;
    uint32_t *d_stream = nullptr;
    Qp2Streams streams{};
    Qp3Pat *d_pat = nullptr;       // sparse K_JC pattern of k_qp3 (num_seg 6, 8), device copy
    double *d_fac = nullptr;       // factor workspace of k_qp3f -> k_qp3: [max_batch][narm][Qp3::FAC]
    Xch xch{nullptr};              // arm-to-arm exchange slots (multi-arm contexts)
    // num_seg 4: second set of structure tables (structure3.hpp order, T bordered out) for the E-free QP kernels (k_qp3f + k_qp4 / k_qp3)
    int *d3_ext_of_int = nullptr, *d3_entry_ptr = nullptr;
    uint32_t *d3_terms = nullptr;
    uint32_t *d_lane4 = nullptr;   // lane-constant table of k_qp4 (qp4_build_lanes)
    int qp13 = 2;                  // QP kernel of num_seg 4: 2 = k_qp2, 3 = k_qp3f + k_qp3, 4 = k_qp3f + k_qp4 (env MPCMP_QP13)
    int qp19 = 5;                  // QP kernel of num_seg 6, one arm: 5 = k_qp3f<6, 1, 5> + k_qp5 (factor resident in registers), 3 = k_qp3f + k_qp3 (env MPCMP_QP19)
    // timing of the dominant kernel (k_qp)
    struct EvPair { hipEvent_t e[2]; };
    std::vector<EvPair> ev;          // at most MAX_EV pairs are ever created; launches beyond that are not timed until the
    size_t ev_used = 0;              // pairs have been folded by mpcmp_kernel_timing
    static constexpr size_t MAX_EV = 4096;
    double qp_ms = 0.0;
    int qp_launches = 0;
    bool timing = false;             // off until mpcmp_kernel_timing is first called: a plain solve records no events
    // growable device scratch of the host-buffer leaf entry points (no hipMalloc per call once it is large enough)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    bool capturing = false;
    // receding-horizon state (mpcmp_rh_*)
    int rh_B = 0;
    size_t lam_count = 0;          // doubles in ws.lam (max_batch x (m + n))
    bool rh_first = true;
    hipGraph_t rh_graph = nullptr;
    hipGraphExec_t rh_exec = nullptr;
    double rh_dt = 0.0;
    int *d_retired = nullptr;                    // [max_batch] arrived instances of the receding-horizon loop (k_advance); not re-solved
    unsigned long long *d_rh_count = nullptr;    // [2] re-solves executed, instances retired (since mpcmp_rh_init)
    bool rh_step = false;                        // the solve being enqueued is a step of the receding-horizon driver (defaults of its flags, retired instances)
    // diagnostics read from the environment ONCE PER CONTEXT, in mpcmp_create (never process-global state)
    bool force_v1 = false, single_stream = false, debug_occ = false;
    int parts_env = 0;                           // MPCMP_STREAMS: parts of a large batch (0 = automatic, see solve_impl)
};

// the configuration the kernels of a solve see: the two start flags resolved to 0 / 1.  0 = the library default (off for mpcmp_solve_batch, ON for
// the receding-horizon driver), 1 = on, -1 = off everywhere (include/mpcmp.h)
static mpcmp_config run_config(const mpcmp_ctx *ctx) {
    mpcmp_config c = ctx->cfg;
    c.qp_warm_start = ctx->rh_step ? (c.qp_warm_start >= 0) : (c.qp_warm_start > 0);
    c.carry_multipliers = ctx->rh_step ? (c.carry_multipliers >= 0) : (c.carry_multipliers > 0);
    return c;
}

static thread_local std::string g_err;

#define HIPCHK(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            std::ostringstream os_;                                                           \
            os_ << #call << " failed: " << hipGetErrorString(e_) << " (" << __FILE__ << ":" << __LINE__ << ")"; \
            if (ctx) (ctx)->err = os_.str();                                                  \
            g_err = os_.str();                                                                \
            return MPCMP_ERUNTIME;                                                            \
        }                                                                                     \
    } while (0)

// ------------------------------------------------------------------------------------------------
// configuration helpers (host only)
extern "C" const char *mpcmp_version(void) { return "mpcmp 0.1 (gfx950)"; }

extern "C" int mpcmp_num_nodes(int num_seg) { return 3 * num_seg + 1; }

extern "C" int mpcmp_time_nodes(int num_seg, double *tau) {
    if (num_seg < 1 || !tau) return MPCMP_EINVAL;
    static const double xi[4] = {-1.0, -0.5, 0.5, 1.0};
    for (int s = 0; s < num_seg; s++)
        for (int j = 0; j < 4; j++) tau[3 * s + j] = (s + 0.5 * (xi[j] + 1.0)) / num_seg;
    return MPCMP_OK;
}

extern "C" int mpcmp_default_limits(double *qmin, double *qmax, double *vmax, double *amax, double *jmax,
                                    double *taumax) {
    // Franka Emika limits as tabulated by the reference, robot_utils/pandaWrapper.hpp:29-34
    static const double t[6][7] = {{-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973},
                                   {2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973},
                                   {2.1750, 2.1750, 2.1750, 2.1750, 2.6100, 2.6100, 2.6100},
                                   {15.0, 7.5, 10.0, 12.5, 15.0, 20.0, 20.0},
                                   {7500, 3750, 5000, 6250, 7500, 10000, 10000},
                                   {87, 87, 87, 87, 12, 12, 12}};
    double *o[6] = {qmin, qmax, vmax, amax, jmax, taumax};
    for (int k = 0; k < 6; k++)
        if (o[k]) std::memcpy(o[k], t[k], sizeof t[k]);
    return MPCMP_OK;
}

extern "C" int mpcmp_set_min_height(mpcmp_config *c, double min_height) {
    if (!c) return MPCMP_EINVAL;
    c->lbg[7] = min_height;
    c->ubg[7] = INFINITY;
    return MPCMP_OK;
}

extern "C" int mpcmp_set_margins(mpcmp_config *c, double mp, double mv, double ma, double mt) {
    if (!c) return MPCMP_EINVAL;
    double qmin[7], qmax[7], vmax[7], amax[7], tmax[7];
    mpcmp_default_limits(qmin, qmax, vmax, amax, nullptr, tmax);
    for (int j = 0; j < 7; j++) {
        const double s = (1.0 - mp) * (qmax[j] - qmin[j]) / 2.0;   // motionPlanner.cpp:66
        c->lbx[j] = qmin[j] + s; c->ubx[j] = qmax[j] - s;
        c->lbx[7 + j] = -mv * vmax[j]; c->ubx[7 + j] = mv * vmax[j];
        c->lbu[j] = -ma * amax[j]; c->ubu[j] = ma * amax[j];
        c->lbg[j] = -mt * tmax[j]; c->ubg[j] = mt * tmax[j];
    }
    c->lbT = 0.0; c->ubT = 10.0;                                  // motionPlanner.cpp:77-78
    return mpcmp_set_min_height(c, 0.05);                         // pandaWrapper.hpp:40
}

extern "C" int mpcmp_default_config(mpcmp_config *c, int num_seg, int sqp_iters) {
    if (!c || num_seg < 1) return MPCMP_EINVAL;
    std::memset(c, 0, sizeof *c);
    c->num_seg = num_seg; c->sqp_iters = sqp_iters;
    c->qp_iters = 700; c->ls_iters = 10; c->check_every = 25; c->quirk_dtau_dT = 1;
    c->eps_abs = 1e-3; c->eps_rel = 1e-3;
    c->rho = 0.02; c->sigma = 1e-6; c->alpha = 1.4; c->rho_eq_scale = 1e3;      // fitted to the reference's stored solve: tools/polympc_param_fit.py
    c->ls_eta = 0.25; c->ls_tau = 0.5; c->hess_reg = 1e-3; c->eps_target = 1e-2;
    return mpcmp_set_margins(c, 1.0, 1.0, 1.0, 1.0);              // motionPlanner.cpp:24
}

static void lump_last_body(mpcmp_model *m, const double mk[3], const double ck[3][3], const double Ik[3][9]) {
    double M = mk[0] + mk[1] + mk[2], c[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) for (int d = 0; d < 3; d++) c[d] += mk[k] * ck[k][d];
    for (int d = 0; d < 3; d++) c[d] /= M;
    double It[9] = {0};
    for (int k = 0; k < 3; k++) {
        const double d[3] = {ck[k][0] - c[0], ck[k][1] - c[1], ck[k][2] - c[2]};
        const double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        for (int r = 0; r < 3; r++) for (int s = 0; s < 3; s++)
            It[3 * r + s] += Ik[k][3 * r + s] + mk[k] * ((r == s ? d2 : 0.0) - d[r] * d[s]);
    }
    m->mass[6] = M;
    std::memcpy(m->com[6], c, sizeof c);
    std::memcpy(m->I[6], It, sizeof It);
}

static void rpy_to_R(const double rpy[3], double R[9]) {
    const double cr = std::cos(rpy[0]), sr = std::sin(rpy[0]), cp = std::cos(rpy[1]), sp = std::sin(rpy[1]),
                 cy = std::cos(rpy[2]), sy = std::sin(rpy[2]);
    // R = Rz(yaw) Ry(pitch) Rx(roll)
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

// ---- scenario helpers (host): robot_utils/pandaWrapper.cpp:14-107 ----
extern "C" int mpcmp_tool_jacobian(const mpcmp_model *model, const double *q, double *J, double *p, double *R) {
    if (!model || !q || !J) return MPCMP_EINVAL;
    double pp[3], RR[9];
    mpcmp_host::tool_jacobian(*model, q, J, pp, RR);
    if (p) for (int k = 0; k < 3; k++) p[k] = pp[k];
    if (R) for (int k = 0; k < 9; k++) R[k] = RR[k];
    return MPCMP_OK;
}

extern "C" int mpcmp_forward_velocities(const mpcmp_model *model, const double *q, const double *qd, double *out) {
    if (!model || !q || !qd || !out) return MPCMP_EINVAL;
    double J[42], p[3], R[9];
    mpcmp_host::tool_jacobian(*model, q, J, p, R);
    for (int r = 0; r < 6; r++) { double s = 0; for (int k = 0; k < 7; k++) s += J[r * 7 + k] * qd[k]; out[r] = s; }
    return MPCMP_OK;
}

extern "C" int mpcmp_inverse_velocities(const mpcmp_model *model, const double *q, const double *lin, const double *ang, double *qd) {
    if (!model || !q || !lin || !ang || !qd) return MPCMP_EINVAL;
    double J[42], p[3], R[9], x[6];
    mpcmp_host::tool_jacobian(*model, q, J, p, R);
    const double b[6] = {lin[0], lin[1], lin[2], ang[0], ang[1], ang[2]};
    if (!mpcmp_host::solve_normal(J, 1e-5, b, x)) return MPCMP_ERUNTIME;
    for (int k = 0; k < 7; k++) { double s = 0; for (int r = 0; r < 6; r++) s += J[r * 7 + k] * x[r]; qd[k] = s; }
    return MPCMP_OK;
}

extern "C" int mpcmp_inverse_kinematics(const mpcmp_model *model, const double *Rdes, const double *pdes, const double *q_init,
                                        double *q, int *iters) {
    if (!model || !Rdes || !pdes || !q) return MPCMP_EINVAL;
    const double eps = 1e-4, DT = 1e-1, damp = 1e-2;
    const int IT_MAX = 1000;
    for (int k = 0; k < 7; k++) q[k] = q_init ? q_init[k] : 0.0;
    for (int it = 0;; it++) {
        double J[42], p[3], R[9];
        mpcmp_host::tool_jacobian(*model, q, J, p, R);
        // dMf = oMdes^-1 * oMf
        double dR[9], dp[3], err[6];
        const double t[3] = {p[0] - pdes[0], p[1] - pdes[1], p[2] - pdes[2]};
        for (int r = 0; r < 3; r++) {
            dp[r] = Rdes[0 * 3 + r] * t[0] + Rdes[1 * 3 + r] * t[1] + Rdes[2 * 3 + r] * t[2];
            for (int c = 0; c < 3; c++) dR[3 * r + c] = Rdes[0 * 3 + r] * R[0 * 3 + c] + Rdes[1 * 3 + r] * R[1 * 3 + c] + Rdes[2 * 3 + r] * R[2 * 3 + c];
        }
        mpcmp_host::log6(dR, dp, err);
        double nrm = 0;
        for (int k = 0; k < 6; k++) nrm += err[k] * err[k];
        if (iters) *iters = it;
        if (std::sqrt(nrm) < eps) return MPCMP_OK;
        if (it >= IT_MAX) return 1;
        // LOCAL-frame Jacobian: rotate the world-aligned rows back with R^T
        double Jl[42], x[6];
        for (int i = 0; i < 7; i++)
            for (int r = 0; r < 3; r++) {
                Jl[r * 7 + i] = R[0 * 3 + r] * J[0 * 7 + i] + R[1 * 3 + r] * J[1 * 7 + i] + R[2 * 3 + r] * J[2 * 7 + i];
                Jl[(3 + r) * 7 + i] = R[0 * 3 + r] * J[3 * 7 + i] + R[1 * 3 + r] * J[4 * 7 + i] + R[2 * 3 + r] * J[5 * 7 + i];
            }
        if (!mpcmp_host::solve_normal(Jl, damp, err, x)) return MPCMP_ERUNTIME;
        for (int k = 0; k < 7; k++) { double s = 0; for (int r = 0; r < 6; r++) s += Jl[r * 7 + k] * x[r]; q[k] -= DT * s; }
    }
}

extern "C" int mpcmp_default_model(mpcmp_model *m) {
    if (!m) return MPCMP_EINVAL;
    // Kinematic / inertial parameters of the Panda arm (robot_utils/panda-model/panda_arm.urdf)
    static const double rpy[7][3] = {{0, 0, 0}, {-1.57079632679, 0, 0}, {1.57079632679, 0, 0}, {1.57079632679, 0, 0},
                                     {-1.57079632679, 0, 0}, {1.57079632679, 0, 0}, {1.57079632679, 0, 0}};
    static const double xyz[7][3] = {{0, 0, 0.333}, {0, 0, 0}, {0, -0.316, 0}, {0.0825, 0, 0},
                                     {-0.0825, 0.384, 0}, {0, 0, 0}, {0.088, 0, 0}};
    static const double mass[7] = {4.970684, 0.646926, 3.228604, 3.587895, 1.225946, 1.666555, 0.735522};
    static const double com[7][3] = {{3.875e-03, 2.081e-03, -0.1750}, {-3.141e-03, -2.872e-02, 3.495e-03},
                                     {2.7518e-02, 3.9252e-02, -6.6502e-02}, {-5.317e-02, 1.04419e-01, 2.7454e-02},
                                     {-1.1953e-02, 4.1065e-02, -3.8437e-02}, {6.0149e-02, -1.4117e-02, -1.0517e-02},
                                     {1.0517e-02, -4.252e-03, 6.1597e-02}};
    static const double in6[7][6] = {{7.0337e-01, -1.3900e-04, 6.7720e-03, 7.0661e-01, 1.9169e-02, 9.1170e-03},
                                     {7.9620e-03, -3.9250e-03, 1.0254e-02, 2.8110e-02, 7.0400e-04, 2.5995e-02},
                                     {3.7242e-02, -4.7610e-03, -1.1396e-02, 3.6155e-02, -1.2805e-02, 1.0830e-02},
                                     {2.5853e-02, 7.7960e-03, -1.3320e-03, 1.9552e-02, 8.6410e-03, 2.8323e-02},
                                     {3.5549e-02, -2.1170e-03, -4.0370e-03, 2.9474e-02, 2.2900e-04, 8.6270e-03},
                                     {1.9640e-03, 1.0900e-04, -1.1580e-03, 4.3540e-03, 3.4100e-04, 5.4330e-03},
                                     {1.2516e-02, -4.2800e-04, -1.1960e-03, 1.0027e-02, -7.4100e-04, 4.8150e-03}};
    std::memset(m, 0, sizeof *m);
    for (int i = 0; i < 7; i++) {
        rpy_to_R(rpy[i], m->R0[i]);
        std::memcpy(m->p[i], xyz[i], sizeof xyz[i]);
        m->mass[i] = mass[i];
        std::memcpy(m->com[i], com[i], sizeof com[i]);
        const double *I = in6[i];
        const double F[9] = {I[0], I[1], I[2], I[1], I[3], I[4], I[2], I[4], I[5]};
        std::memcpy(m->I[i], F, sizeof F);
    }
    m->link8[2] = 0.107;
    m->tool[2] = 0.107 + 0.15;
    m->gravity[2] = -9.81;
    // fixed children of link7: link8 (m=0, I=1e-3 Id) and panda_tool (m=1, I=1e-3 Id) are lumped into body 7
    const double mk[3] = {mass[6], 0.0, 1.0};
    const double ck[3][3] = {{com[6][0], com[6][1], com[6][2]}, {0, 0, 0.107}, {0, 0, 0.257}};
    double Ik[3][9];
    std::memcpy(Ik[0], m->I[6], sizeof Ik[0]);
    for (int k = 1; k < 3; k++) { std::memset(Ik[k], 0, sizeof Ik[k]); Ik[k][0] = Ik[k][4] = Ik[k][8] = 1e-3; }
    lump_last_body(m, mk, ck, Ik);
    return MPCMP_OK;
}

// ---- URDF reader: any number of serial 7-joint chains on one base (robot_utils/pandaWrapper.cpp:3-12 takes any URDF) --------
// What the kernels need of a robot is, per chain, seven revolute joints about the z axis of their own frame, the placement of
// each joint in its parent joint's frame, one rigid body per joint frame and the two named frames behind the last joint.  A URDF
// is brought into that form here: fixed joints (before, between and behind the revolute joints, rotated or not) are folded into
// the next joint's placement and their links lumped into the body of the joint frame they hang on; a chain's base placement is
// folded into its first joint; rotated inertial frames rotate the inertia tensor; a joint axis other than +z is absorbed by a
// constant rotation of the child frame.
namespace {
struct UrdfLink { double mass = 0, com[3] = {0, 0, 0}, I[9] = {0}, irpy[3] = {0, 0, 0}; bool has_inertial = false; };
struct UrdfJoint { std::string name, type, parent, child; double xyz[3] = {0, 0, 0}, rpy[3] = {0, 0, 0}, axis[3] = {0, 0, 1}; };
struct Xf { double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0}; };       // rigid transform, parent <- child

static std::string attr(const std::string &tag, const std::string &key) {
    size_t p = 0;
    while ((p = tag.find(key + "=", p)) != std::string::npos) {
        if (p > 0 && (isalnum((unsigned char)tag[p - 1]) || tag[p - 1] == '_')) { p += key.size(); continue; }
        const size_t q0 = p + key.size() + 1;
        const char quote = tag[q0];
        const size_t q1 = tag.find(quote, q0 + 1);
        return tag.substr(q0 + 1, q1 - q0 - 1);
    }
    return "";
}
static void parse3(const std::string &s, double *o) {
    if (s.empty()) return;
    std::istringstream is(s);
    is >> o[0] >> o[1] >> o[2];
}
static bool is_identity(const double *R) {
    return R[0] == 1 && R[4] == 1 && R[8] == 1 && R[1] == 0 && R[2] == 0 && R[3] == 0 && R[5] == 0 && R[6] == 0 && R[7] == 0;
}
static void matmul3(const double *A, const double *B, double *Cm) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Cm[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
// (identity factors are skipped, so that a URDF without rotated fixed joints gives bit-for-bit the numbers it holds)
static Xf compose(const Xf &a, const Xf &b) {
    Xf o;
    if (is_identity(a.R)) { std::memcpy(o.R, b.R, sizeof o.R); for (int d = 0; d < 3; d++) o.t[d] = a.t[d] + b.t[d]; }
    else {
        if (is_identity(b.R)) std::memcpy(o.R, a.R, sizeof o.R); else matmul3(a.R, b.R, o.R);
        for (int d = 0; d < 3; d++) o.t[d] = a.t[d] + (a.R[3 * d] * b.t[0] + a.R[3 * d + 1] * b.t[1] + a.R[3 * d + 2] * b.t[2]);
    }
    return o;
}
static void rotate_inertia(const double *R, const double *I, double *o) {          // R I R^T
    if (is_identity(R)) { std::memcpy(o, I, 9 * sizeof(double)); return; }
    double RI[9], Rt[9];
    matmul3(R, I, RI);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rt[3 * r + c] = R[3 * c + r];
    matmul3(RI, Rt, o);
}
// rotation that takes +z to the (normalised) axis a
static bool axis_rotation(const double *a_in, double *R) {
    const double n = std::sqrt(a_in[0] * a_in[0] + a_in[1] * a_in[1] + a_in[2] * a_in[2]);
    if (!(n > 0)) return false;
    const double a[3] = {a_in[0] / n, a_in[1] / n, a_in[2] / n};
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (a[0] == 0 && a[1] == 0 && a[2] == 1) { std::memcpy(R, I3, sizeof I3); return true; }
    if (a[0] == 0 && a[1] == 0 && a[2] == -1) { const double F[9] = {1, 0, 0, 0, -1, 0, 0, 0, -1}; std::memcpy(R, F, sizeof F); return true; }
    // Rodrigues: axis v = z x a, sin = |v|, cos = a_z
    const double v[3] = {-a[1], a[0], 0.0}, c = a[2], k = 1.0 / (1.0 + c);
    const double V[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
    double V2[9];
    matmul3(V, V, V2);
    for (int i = 0; i < 9; i++) R[i] = I3[i] + V[i] + k * V2[i];
    return true;
}
struct BodyAcc { std::vector<double> m; std::vector<std::array<double, 3>> c; std::vector<std::array<double, 9>> I; };
struct ChainAcc {
    std::string first_joint;
    std::vector<Xf> placement;                      // joint i in joint i - 1 (the first: in the base / world frame)
    std::vector<BodyAcc> bodies;                    // per joint frame
    std::vector<std::vector<std::array<double, 3>>> frames;     // fixed descendants of each joint frame (origins, in order of discovery)
};
// same arithmetic, in the same order, as lump_last_body for three bodies
static void lump(const BodyAcc &b, double &M, double *c, double *It) {
    const size_t n = b.m.size();
    M = b.m[0];
    for (size_t k = 1; k < n; k++) M = M + b.m[k];
    c[0] = c[1] = c[2] = 0;
    for (size_t k = 0; k < n; k++) for (int d = 0; d < 3; d++) c[d] += b.m[k] * b.c[k][d];
    for (int d = 0; d < 3; d++) c[d] /= M;
    for (int i = 0; i < 9; i++) It[i] = 0;
    for (size_t k = 0; k < n; k++) {
        const double d[3] = {b.c[k][0] - c[0], b.c[k][1] - c[1], b.c[k][2] - c[2]};
        const double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        for (int r = 0; r < 3; r++) for (int s2 = 0; s2 < 3; s2++)
            It[3 * r + s2] += b.I[k][3 * r + s2] + b.m[k] * ((r == s2 ? d2 : 0.0) - d[r] * d[s2]);
    }
}
}  // namespace

extern "C" int mpcmp_models_from_urdf(const char *path, int max_chains, mpcmp_model *models, int *n_chains) {
    if (!path || !models || !n_chains || max_chains < 1) return MPCMP_EINVAL;
    *n_chains = 0;
    std::ifstream f(path);
    if (!f) { g_err = std::string("cannot open URDF ") + path; return MPCMP_EINVAL; }
    std::stringstream ss; ss << f.rdbuf();
    const std::string xml = ss.str();
    std::map<std::string, UrdfLink> links;
    std::vector<std::string> link_order;
    std::vector<UrdfJoint> joints;
    size_t pos = 0;
    std::string cur_link; UrdfJoint cur_joint; bool in_link = false, in_joint = false, in_inertial = false;
    while ((pos = xml.find('<', pos)) != std::string::npos) {
        if (xml.compare(pos, 4, "<!--") == 0) { const size_t e = xml.find("-->", pos); if (e == std::string::npos) break; pos = e + 3; continue; }
        const size_t end = xml.find('>', pos);
        if (end == std::string::npos) break;
        const std::string tag = xml.substr(pos + 1, end - pos - 1);
        pos = end + 1;
        if (tag.empty() || tag[0] == '?' || tag[0] == '!') continue;
        std::istringstream ts(tag); std::string name; ts >> name;
        if (!name.empty() && name.back() == '/') name.pop_back();
        if (name == "link") { cur_link = attr(tag, "name"); if (!links.count(cur_link)) link_order.push_back(cur_link); links[cur_link]; in_link = tag.back() != '/'; }
        else if (name == "/link") in_link = false;
        else if (name == "joint" && tag.find("type=") != std::string::npos) {
            cur_joint = UrdfJoint(); cur_joint.type = attr(tag, "type"); cur_joint.name = attr(tag, "name"); in_joint = true;
        } else if (name == "/joint") { if (in_joint) joints.push_back(cur_joint); in_joint = false; }
        else if (name == "inertial") in_inertial = true;
        else if (name == "/inertial") in_inertial = false;
        else if (name == "origin") {
            if (in_joint) { parse3(attr(tag, "xyz"), cur_joint.xyz); parse3(attr(tag, "rpy"), cur_joint.rpy); }
            else if (in_link && in_inertial) { parse3(attr(tag, "xyz"), links[cur_link].com); parse3(attr(tag, "rpy"), links[cur_link].irpy); }
        } else if (name == "mass" && in_link && in_inertial) { links[cur_link].mass = atof(attr(tag, "value").c_str()); links[cur_link].has_inertial = true; }
        else if (name == "inertia" && in_link && in_inertial) {
            UrdfLink &L = links[cur_link];
            const double xx = atof(attr(tag, "ixx").c_str()), xy = atof(attr(tag, "ixy").c_str()), xz = atof(attr(tag, "ixz").c_str()),
                         yy = atof(attr(tag, "iyy").c_str()), yz = atof(attr(tag, "iyz").c_str()), zz = atof(attr(tag, "izz").c_str());
            const double F[9] = {xx, xy, xz, xy, yy, yz, xz, yz, zz};
            std::memcpy(L.I, F, sizeof F);
        } else if (name == "parent" && in_joint) cur_joint.parent = attr(tag, "link");
        else if (name == "child" && in_joint) cur_joint.child = attr(tag, "link");
        else if (name == "axis" && in_joint) parse3(attr(tag, "xyz"), cur_joint.axis);
    }
    if (links.empty()) { g_err = "URDF holds no links"; return MPCMP_EINVAL; }
    // the kinematic tree
    std::map<std::string, std::vector<const UrdfJoint *>> kids;
    std::map<std::string, int> is_child;
    for (auto &j : joints) {
        if (!links.count(j.parent) || !links.count(j.child)) { g_err = "joint " + j.name + " names an unknown link"; return MPCMP_EINVAL; }
        kids[j.parent].push_back(&j);
        if (is_child[j.child]++) { g_err = "link " + j.child + " has two parent joints"; return MPCMP_EINVAL; }
    }
    std::string root;
    for (auto &ln : link_order) if (!is_child.count(ln)) { if (!root.empty()) { g_err = "URDF has more than one root link"; return MPCMP_EINVAL; } root = ln; }
    if (root.empty()) { g_err = "URDF has no root link (cycle)"; return MPCMP_EINVAL; }
    std::vector<ChainAcc> chains;
    std::string err;
    // depth first in file order; C: link frame in the current model frame (the world before a chain starts, then the joint frame)
    std::function<bool(const std::string &, const Xf &, int, int)> visit = [&](const std::string &ln, const Xf &C, int ci, int ji) -> bool {
        const UrdfLink &L = links[ln];
        if (ci >= 0) {
            ChainAcc &ch = chains[ci];
            if (L.has_inertial) {
                double Rin[9], Il[9], If[9];
                rpy_to_R(L.irpy, Rin);
                rotate_inertia(Rin, L.I, Il);          // inertial frame -> link axes
                rotate_inertia(C.R, Il, If);           // link axes -> joint frame axes
                Xf com; std::memcpy(com.t, L.com, sizeof L.com);
                const Xf cj = compose(C, com);
                ch.bodies[ji].m.push_back(L.mass);
                ch.bodies[ji].c.push_back({cj.t[0], cj.t[1], cj.t[2]});
                std::array<double, 9> Ia; std::memcpy(Ia.data(), If, sizeof If);
                ch.bodies[ji].I.push_back(Ia);
            }
        }
        int revolute_kids = 0;
        for (const UrdfJoint *j : kids[ln]) {
            Xf T; rpy_to_R(j->rpy, T.R); std::memcpy(T.t, j->xyz, sizeof j->xyz);
            if (j->type == "fixed") {
                const Xf Cc = compose(C, T);
                if (ci >= 0) chains[ci].frames[ji].push_back({Cc.t[0], Cc.t[1], Cc.t[2]});
                if (!visit(j->child, Cc, ci, ji)) return false;
            } else if (j->type == "revolute" || j->type == "continuous") {
                if (ci >= 0 && ++revolute_kids > 1) { err = "link " + ln + " carries two revolute joints: not a serial chain"; return false; }
                Xf Ra;
                if (!axis_rotation(j->axis, Ra.R)) { err = "joint " + j->name + " has a zero axis"; return false; }
                int c2 = ci;
                if (c2 < 0) { chains.emplace_back(); c2 = (int)chains.size() - 1; chains[c2].first_joint = j->name; }
                else if (ji != (int)chains[c2].placement.size() - 1) { err = "joint " + j->name + " branches off the chain starting at " + chains[c2].first_joint; return false; }
                ChainAcc &ch = chains[c2];
                ch.placement.push_back(compose(compose(C, T), Ra));
                ch.bodies.emplace_back(); ch.frames.emplace_back();
                Xf Cn;                                  // child link frame in the new joint frame: the inverse of the axis rotation
                for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Cn.R[3 * r + c] = Ra.R[3 * c + r];
                if (!visit(j->child, Cn, c2, (int)ch.placement.size() - 1)) return false;
            } else { err = "joint " + j->name + ": type '" + j->type + "' is not supported (fixed, revolute, continuous)"; return false; }
        }
        return true;
    };
    // (a chain must run through consecutive joint frames: the recursion above passes the frame index down, so a revolute joint
    //  found below joint frame ji of chain ci extends that chain only if ji is its last frame)
    if (!visit(root, Xf(), -1, -1)) { g_err = err; return MPCMP_EINVAL; }
    if (chains.empty()) { g_err = "URDF holds no revolute joint"; return MPCMP_EINVAL; }
    if ((int)chains.size() > max_chains) { g_err = "URDF holds more chains than the caller has room for"; return MPCMP_ETOOBIG; }
    for (size_t ci = 0; ci < chains.size(); ci++) {
        const ChainAcc &ch = chains[ci];
        if (ch.placement.size() != 7) {
            g_err = "the chain starting at joint " + ch.first_joint + " has " + std::to_string(ch.placement.size()) + " revolute joints; the kernels are built for 7";
            return MPCMP_EINVAL;
        }
        mpcmp_model *m = models + ci;
        std::memset(m, 0, sizeof *m);
        for (int i = 0; i < 7; i++) {
            std::memcpy(m->R0[i], ch.placement[i].R, sizeof m->R0[i]);
            std::memcpy(m->p[i], ch.placement[i].t, sizeof m->p[i]);
            const BodyAcc &b = ch.bodies[i];
            if (b.m.empty()) continue;                                   // (a massless joint frame)
            if (b.m.size() == 1) { m->mass[i] = b.m[0]; std::memcpy(m->com[i], b.c[0].data(), sizeof m->com[i]); std::memcpy(m->I[i], b.I[0].data(), sizeof m->I[i]); }
            else lump(b, m->mass[i], m->com[i], m->I[i]);
        }
        // the two named frames of the reference behind the last joint (panda_link8, panda_tool: panda_arm.urdf:134-153): the first
        // and the last fixed descendant of joint frame 7
        const auto &fr = ch.frames[6];
        if (!fr.empty()) { std::memcpy(m->link8, fr.front().data(), sizeof m->link8); std::memcpy(m->tool, fr.back().data(), sizeof m->tool); }
        m->gravity[2] = -9.81;
    }
    *n_chains = (int)chains.size();
    return MPCMP_OK;
}

extern "C" int mpcmp_model_from_urdf(const char *path, mpcmp_model *m) {
    if (!path || !m) return MPCMP_EINVAL;
    mpcmp_model tmp[8];
    int n = 0;
    const int rc = mpcmp_models_from_urdf(path, 8, tmp, &n);
    if (rc != MPCMP_OK) return rc;
    if (n != 1) { g_err = "URDF holds " + std::to_string(n) + " chains: use mpcmp_models_from_urdf / mpcmp_create_multi"; return MPCMP_EINVAL; }
    *m = tmp[0];
    return MPCMP_OK;
}

// ------------------------------------------------------------------------------------------------
// context
template <typename T>
static int dalloc(mpcmp_ctx *ctx, T **p, size_t count) {
    void *q = nullptr;
    HIPCHK(ctx, hipMalloc(&q, count * sizeof(T)));
    ctx->allocs.push_back(q);
    *p = (T *)q;
    return MPCMP_OK;
}

static int validate(const mpcmp_config *c, std::string &err) {
    if (c->num_seg != 1 && c->num_seg != 2 && c->num_seg != 4 && c->num_seg != 6 && c->num_seg != 8) { err = "num_seg must be 1, 2, 4, 6 or 8 on the HIP backend"; return MPCMP_EINVAL; }
    if (c->sqp_iters < 1 || c->qp_iters < 1 || c->check_every < 1) { err = "iteration counts must be >= 1"; return MPCMP_EINVAL; }
    if (c->ls_iters < 2 || c->ls_iters > 10) { err = "ls_iters must be in [2,10]"; return MPCMP_EINVAL; }
    if (!(c->rho > 0) || !(c->sigma > 0) || !(c->alpha > 0 && c->alpha < 2)) { err = "rho, sigma > 0 and 0 < alpha < 2 required"; return MPCMP_EINVAL; }
    if (c->qp_warm_start < -1 || c->qp_warm_start > 1 || c->carry_multipliers < -1 || c->carry_multipliers > 1) { err = "qp_warm_start and carry_multipliers are -1, 0 or 1"; return MPCMP_EINVAL; }
    return MPCMP_OK;
}

extern "C" const char *mpcmp_last_error(const mpcmp_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

extern "C" int mpcmp_destroy(mpcmp_ctx *ctx) {
    if (!ctx) return MPCMP_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->rh_exec) (void)hipGraphExecDestroy(ctx->rh_exec);
    if (ctx->rh_graph) (void)hipGraphDestroy(ctx->rh_graph);
    for (void *p : ctx->allocs) (void)hipFree(p);
    for (auto &e : ctx->ev) { (void)hipEventDestroy(e.e[0]); (void)hipEventDestroy(e.e[1]); }
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    for (int k = 0; k < 3; k++) { if (ctx->stream_x[k]) (void)hipStreamDestroy(ctx->stream_x[k]); if (ctx->ev_join[k]) (void)hipEventDestroy(ctx->ev_join[k]); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    delete ctx;
    return MPCMP_OK;
}

static int create_impl(const mpcmp_config *cfg, const mpcmp_model *model, int narm, int device, int max_batch, mpcmp_ctx **out);
static int prepare_kernels(mpcmp_ctx *ctx);

extern "C" int mpcmp_create(const mpcmp_config *cfg, const mpcmp_model *model, int device, int max_batch, mpcmp_ctx **out) {
    return create_impl(cfg, model, 1, device, max_batch, out);
}

// Multi-arm robot: `models` points at narm models (independent 7-joint chains on one base; a chain's base placement is folded
// into its first joint placement).  The OCP is the reference's (robot_ocp.hpp:31-213) with NX = 14 narm, NU = 7 narm,
// NG = 8 narm; the arms couple only through the final time.  BASELINE.json configs[3]: narm = 2, num_seg = 8.
extern "C" int mpcmp_create_multi(const mpcmp_config *cfg, const mpcmp_model *models, int narm, int device, int max_batch, mpcmp_ctx **out) {
    if (narm < 1 || narm > 2 || (narm > 1 && !models)) { g_err = "narm must be 1 or 2 (with narm models)"; return MPCMP_EINVAL; }
    return create_impl(cfg, models, narm, device, max_batch, out);
}

static int create_impl(const mpcmp_config *cfg, const mpcmp_model *model, int narm, int device, int max_batch, mpcmp_ctx **out) {
    if (!cfg || !out || max_batch < 1) return MPCMP_EINVAL;
    *out = nullptr;
    if (int rc = validate(cfg, g_err)) return rc;
    if (narm == 2) {
        if (cfg->num_seg != 6 && cfg->num_seg != 8) { g_err = "multi-arm OCPs need num_seg 6 or 8 (k_qp3)"; return MPCMP_EINVAL; }
        if (8 + cfg->qp_iters + 1 + 8 * (cfg->qp_iters / cfg->check_every + 1) > MPCMP_XCH_STRIDE) { g_err = "multi-arm OCPs: qp_iters too large for the exchange slots"; return MPCMP_EINVAL; }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device >= ndev) {
        g_err = "no HIP device available: the mpcmp product path has no CPU fallback";
        return MPCMP_ENODEVICE;
    }
    mpcmp_ctx *ctx = new mpcmp_ctx();
    ctx->cfg = *cfg; ctx->device = device; ctx->max_batch = max_batch; ctx->nseg = cfg->num_seg; ctx->narm = narm;
    ctx->nx = 14 * narm; ctx->nu = 7 * narm;
    ctx->N = 3 * cfg->num_seg + 1; ctx->n = 21 * ctx->N * narm + 1; ctx->meq = 14 * (ctx->N - 1) * narm;
    ctx->m = ctx->meq + 8 * ctx->N * narm; ctx->mn = ctx->m + ctx->n;
    auto fail = [&](int rc) { g_err = ctx->err; mpcmp_destroy(ctx); return rc; };
#define TRY(x) do { int rc_ = (x); if (rc_) return fail(rc_); } while (0)
#define HIPTRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return fail(MPCMP_ERUNTIME); } } while (0)
    HIPTRY(hipSetDevice(device));
    HIPTRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    for (int k = 0; k < 3; k++) {
        HIPTRY(hipStreamCreateWithFlags(&ctx->stream_x[k], hipStreamNonBlocking));
        HIPTRY(hipEventCreateWithFlags(&ctx->ev_join[k], hipEventDisableTiming));
    }
    HIPTRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    mpcmp_model mdl[2];
    for (int a = 0; a < narm; a++) { if (model) mdl[a] = model[a]; else mpcmp_default_model(&mdl[a]); }
    TRY(dalloc(ctx, &ctx->d_model, narm));
    HIPTRY(hipMemcpy(ctx->d_model, mdl, sizeof(mpcmp_model) * narm, hipMemcpyHostToDevice));
    ctx->model = mdl[0];
    StructureTables tab;
    // k_qp3 / k_qp4: per-arm tables with T bordered out (structure3.hpp), in ELL form
    auto build_v3 = [&](int nseg, StructureTables &tb, Qp3Pat &pat) -> bool {
        Tables3 t3;
        if (!build_tables3(nseg, t3)) return false;
        tb.nseg = t3.nseg; tb.ext_of_int = t3.ext_of_int; pat = t3.pat;
        // term lists in ELL form [term index][entry] (entry stride padded to 64): consecutive threads assemble consecutive entries,
        // so every load of a term word is one coalesced 256-byte request per wave (the per-entry lists of the CSR form cost a
        // cache line per 4-byte word, and every workgroup of the launch reads the same table: 0.47 M cycles per factorisation)
        const int E = (int)t3.entry_ptr.size() - 1, EP = (E + 63) / 64 * 64;
        // (kappa, the (T, T) entry, has one term per row of A: the kernel sums it by a workgroup reduction instead)
        const int e_kap = nseg * 1225 + 28 + nseg * 196 + nseg * 98 + 98 + 21 * (3 * nseg + 1);      // Dim3<nseg>::eKap
        int tmax = 0;
        for (int e = 0; e < E; e++) if (e != e_kap) tmax = std::max(tmax, t3.entry_ptr[e + 1] - t3.entry_ptr[e]);
        tb.entry_ptr.assign(EP, 0);
        tb.terms.assign((size_t)tmax * EP, 0xFFFFFFFFu);
        // Slots instead of entries: inside each of the two assembly calls' ranges ([0, EA): everything but K_II; [EA, E): K_II) the
        // entries are sorted by the length of their term lists, longest first, so that the 64 lanes of a wave (consecutive slots)
        // walk lists of (nearly) equal length.  In entry order a wave executed every step of its longest list for all of its 10 - 13
        // entries per lane: 140 term evaluations per lane for 32 terms.  tb.entry_ptr[slot] = count | entry << 8.
        const int EA = e_kap + 1;
        auto count_of = [&](int e) { return e == e_kap ? 0 : t3.entry_ptr[e + 1] - t3.entry_ptr[e]; };
        for (int part = 0; part < 2; part++) {
            const int lo = part ? EA : 0, hi = part ? E : EA;
            std::vector<int> order(hi - lo);
            for (int e = lo; e < hi; e++) order[e - lo] = e;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return count_of(a) > count_of(b); });
            for (int k = 0; k < hi - lo; k++) {
                const int slot = lo + k, e = order[k], c = count_of(e);
                tb.entry_ptr[slot] = c | (e << 8);
                for (int t = 0; t < c; t++) tb.terms[(size_t)t * EP + slot] = t3.terms[t3.entry_ptr[e] + t];
            }
        }
        return true;
    };
    auto fac_doubles = [](int nseg) -> size_t { return nseg == 4 ? std::max<size_t>(Qp3<4>::FAC, Qp4Fac<4>::FAC) : (nseg == 6 ? std::max<size_t>(Qp3<6>::FAC, Qp5Fac<6>::FAC) : Qp3<8>::FAC); };
    if (cfg->num_seg == 6 && narm == 1) {
        if (const char *e = std::getenv("MPCMP_QP19")) ctx->qp19 = std::atoi(e);
        if (ctx->qp19 != 3 && ctx->qp19 != 5) { ctx->err = "MPCMP_QP19 must be 3 (k_qp3) or 5 (k_qp5)"; return fail(MPCMP_EINVAL); }
    }
    if (cfg->num_seg >= 6) {
        Qp3Pat pat;
        if (!build_v3(cfg->num_seg, tab, pat)) { ctx->err = "internal: structure table generation failed"; return fail(MPCMP_EINVAL); }
        TRY(dalloc(ctx, &ctx->d_fac, (size_t)max_batch * narm * fac_doubles(cfg->num_seg)));
        TRY(dalloc(ctx, &ctx->d_pat, 1));
        HIPTRY(hipMemcpy(ctx->d_pat, &pat, sizeof(Qp3Pat), hipMemcpyHostToDevice));
    } else if (!build_tables(cfg->num_seg, tab)) { ctx->err = "internal: structure table generation failed"; return fail(MPCMP_EINVAL); }
    TRY(dalloc(ctx, &ctx->d_ext_of_int, tab.ext_of_int.size()));
    TRY(dalloc(ctx, &ctx->d_entry_ptr, tab.entry_ptr.size()));
    TRY(dalloc(ctx, &ctx->d_terms, tab.terms.size()));
    HIPTRY(hipMemcpy(ctx->d_ext_of_int, tab.ext_of_int.data(), tab.ext_of_int.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPTRY(hipMemcpy(ctx->d_entry_ptr, tab.entry_ptr.data(), tab.entry_ptr.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPTRY(hipMemcpy(ctx->d_terms, tab.terms.data(), tab.terms.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    const size_t B = max_batch, N = ctx->N, n = ctx->n, mn = ctx->mn;
    WS &w = ctx->ws;
    w.model = ctx->d_model; w.ext_of_int = ctx->d_ext_of_int; w.entry_ptr = ctx->d_entry_ptr; w.terms = ctx->d_terms;
    TRY(dalloc(ctx, &w.z, B * n)); TRY(dalloc(ctx, &w.lam, B * mn)); TRY(dalloc(ctx, &w.ceq, B * ctx->meq));
    HIPTRY(hipMemset(w.lam, 0, sizeof(double) * B * mn));        // (mpcmp_config.carry_multipliers: a fresh context starts every slot from lambda = 0)
    ctx->lam_count = B * mn;
    TRY(dalloc(ctx, &w.g, B * 8 * N * narm)); TRY(dalloc(ctx, &w.Gk, B * N * 176 * narm)); TRY(dalloc(ctx, &w.p, B * n));
    TRY(dalloc(ctx, &w.y, B * mn)); TRY(dalloc(ctx, &w.qpit, B)); TRY(dalloc(ctx, &w.perm, B)); TRY(dalloc(ctx, &w.okey, B)); TRY(dalloc(ctx, &w.done, 4));
    HIPTRY(hipMemset(w.done, 0, 4 * sizeof(int))); TRY(dalloc(ctx, &w.qp_total, B));
    TRY(dalloc(ctx, &w.status, B)); HIPTRY(hipMemset(w.status, 0, sizeof(*w.status) * B)); TRY(dalloc(ctx, &w.alpha, B)); TRY(dalloc(ctx, &w.dbg, (size_t)B * MPCMP_DBG_WORDS));
    const size_t nx = ctx->nx, nu = ctx->nu;
    TRY(dalloc(ctx, &ctx->d_x0, B * nx)); TRY(dalloc(ctx, &ctx->d_xf, B * nx));
    TRY(dalloc(ctx, &ctx->d_wx, B * nx * N)); TRY(dalloc(ctx, &ctx->d_wu, B * nu * N)); TRY(dalloc(ctx, &ctx->d_wT, B));
    TRY(dalloc(ctx, &ctx->d_sx, B * nx * N)); TRY(dalloc(ctx, &ctx->d_su, B * nu * N)); TRY(dalloc(ctx, &ctx->d_sT, B));
    TRY(dalloc(ctx, &ctx->d_info, B));
    if (narm == 2) {
        TRY(dalloc(ctx, &ctx->xch.buf, B * 2 * MPCMP_XCH_STRIDE));
        // per-arm staging of the warm-start generator: states, node trajectories, durations of B * narm arm problems
        TRY(dalloc(ctx, &ctx->d_ax0, B * nx)); TRY(dalloc(ctx, &ctx->d_axf, B * nx));
        TRY(dalloc(ctx, &ctx->d_awx, B * nx * N)); TRY(dalloc(ctx, &ctx->d_awu, B * nu * N)); TRY(dalloc(ctx, &ctx->d_awT, B * narm));
    }
    if (cfg->num_seg == 2 || cfg->num_seg == 4) {
        AsmStreams as;
        const int GS = cfg->num_seg == 4 ? Qp2<4>::GS : Qp2<2>::GS, RSv = Qp2<4>::RS;
        if (!build_streams(cfg->num_seg, 1024, Qp2<4>::HS, GS, RSv, as)) { ctx->err = "internal: assembly stream generation failed"; return fail(MPCMP_EINVAL); }
        TRY(dalloc(ctx, &ctx->d_stream, as.words.size()));
        HIPTRY(hipMemcpy(ctx->d_stream, as.words.data(), as.words.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        ctx->streams.words = ctx->d_stream; ctx->streams.npass = as.npass;
        for (int p = 0; p < 8; p++) { ctx->streams.off[p] = as.off[p]; ctx->streams.W[p] = as.W[p]; }
        ctx->streams.split_dst = as.split_dst; ctx->streams.split_scr = as.split_scr; ctx->streams.split_n = as.split_n;
    }
    if (cfg->num_seg == 4) {
        if (const char *e = std::getenv("MPCMP_QP13")) ctx->qp13 = std::atoi(e);
        if (ctx->qp13 != 2 && ctx->qp13 != 3 && ctx->qp13 != 4) { ctx->err = "MPCMP_QP13 must be 2 (k_qp2), 3 (k_qp3) or 4 (k_qp4, experimental: slower than k_qp2)"; return fail(MPCMP_EINVAL); }
#ifndef MPCMP_WITH_QP4
        if (ctx->qp13 == 4) { ctx->err = "MPCMP_QP13=4: this library was built without k_qp4 (experimental kernel: rebuild with -DMPCMP_WITH_QP4 -Itools/experiments)"; return fail(MPCMP_EINVAL); }
#endif
        if (ctx->qp13 != 2 && std::getenv("MPCMP_FORCE_V1")) { ctx->err = "MPCMP_FORCE_V1 and MPCMP_QP13 != 2 exclude each other"; return fail(MPCMP_EINVAL); }
        if (ctx->qp13 != 2) {
            StructureTables t4;
            Qp3Pat pat;
            if (!build_v3(4, t4, pat)) { ctx->err = "internal: structure table generation failed"; return fail(MPCMP_EINVAL); }
            TRY(dalloc(ctx, &ctx->d3_ext_of_int, t4.ext_of_int.size()));
            TRY(dalloc(ctx, &ctx->d3_entry_ptr, t4.entry_ptr.size()));
            TRY(dalloc(ctx, &ctx->d3_terms, t4.terms.size()));
            HIPTRY(hipMemcpy(ctx->d3_ext_of_int, t4.ext_of_int.data(), t4.ext_of_int.size() * sizeof(int), hipMemcpyHostToDevice));
            HIPTRY(hipMemcpy(ctx->d3_entry_ptr, t4.entry_ptr.data(), t4.entry_ptr.size() * sizeof(int), hipMemcpyHostToDevice));
            HIPTRY(hipMemcpy(ctx->d3_terms, t4.terms.data(), t4.terms.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            TRY(dalloc(ctx, &ctx->d_fac, (size_t)max_batch * fac_doubles(4)));
            TRY(dalloc(ctx, &ctx->d_pat, 1));
            HIPTRY(hipMemcpy(ctx->d_pat, &pat, sizeof(Qp3Pat), hipMemcpyHostToDevice));
#ifdef MPCMP_WITH_QP4
            std::vector<uint32_t> lane4((size_t)Qp4<4>::NF * Qp4<4>::NT);
            qp4_build_lanes<4>(pat, t4.ext_of_int.data(), lane4.data());
            TRY(dalloc(ctx, &ctx->d_lane4, lane4.size()));
            HIPTRY(hipMemcpy(ctx->d_lane4, lane4.data(), lane4.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
#endif
        }
    }
    TRY(dalloc(ctx, &ctx->d_retired, B)); HIPTRY(hipMemset(ctx->d_retired, 0, sizeof(int) * B));
    TRY(dalloc(ctx, &ctx->d_rh_count, 2)); HIPTRY(hipMemset(ctx->d_rh_count, 0, 2 * sizeof(unsigned long long)));
    // diagnostics (generic kernel everywhere, stream count, occupancy report): read here, per context
    ctx->force_v1 = std::getenv("MPCMP_FORCE_V1") != nullptr;
    ctx->single_stream = std::getenv("MPCMP_SINGLE_STREAM") != nullptr;
    ctx->debug_occ = std::getenv("MPCMP_DEBUG_OCC") != nullptr;
    if (const char *e = std::getenv("MPCMP_STREAMS")) ctx->parts_env = std::atoi(e);
    TRY(prepare_kernels(ctx));
#undef TRY
#undef HIPTRY
    *out = ctx;
    return MPCMP_OK;
}

extern "C" int mpcmp_set_config(mpcmp_ctx *ctx, const mpcmp_config *cfg) {
    if (!ctx || !cfg) return MPCMP_EINVAL;
    if (cfg->num_seg != ctx->nseg) { ctx->err = "num_seg cannot change after mpcmp_create"; return MPCMP_EINVAL; }
    if (int rc = validate(cfg, ctx->err)) return rc;
    if (ctx->narm == 2 && 8 + cfg->qp_iters + 1 + 8 * (cfg->qp_iters / cfg->check_every + 1) > MPCMP_XCH_STRIDE) {
        ctx->err = "multi-arm OCPs: qp_iters too large for the exchange slots"; return MPCMP_EINVAL;
    }
    ctx->cfg = *cfg;
    // a captured receding-horizon step holds the configuration by value as a kernel argument: re-capture on the next rh_run
    if (ctx->rh_exec) { (void)hipGraphExecDestroy(ctx->rh_exec); ctx->rh_exec = nullptr; }
    if (ctx->rh_graph) { (void)hipGraphDestroy(ctx->rh_graph); ctx->rh_graph = nullptr; }
    return MPCMP_OK;
}

// ------------------------------------------------------------------------------------------------
// launches
template <typename K>
static int set_lds(mpcmp_ctx *ctx, K kern, size_t bytes) {
    HIPCHK(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MPCMP_OK;
}

static hipEvent_t *next_events(mpcmp_ctx *ctx) {
    if (ctx->ev_used == ctx->ev.size()) {
        if (ctx->ev.size() >= mpcmp_ctx::MAX_EV) return nullptr;      // bounded: untimed until the caller folds the pairs
        mpcmp_ctx::EvPair p;
        if (hipEventCreate(&p.e[0]) != hipSuccess) return nullptr;
        if (hipEventCreate(&p.e[1]) != hipSuccess) { (void)hipEventDestroy(p.e[0]); return nullptr; }
        ctx->ev.push_back(p);
    }
    return ctx->ev[ctx->ev_used++].e;
}

// Kernel selection of a single-arm context (compile-time candidates, run-time choice by discretisation and the context's diagnostics)
template <int NSEG>
struct KSel {
    static constexpr bool V2C = (NSEG == 2 || NSEG == 4);      // role-specialised 1024-thread QP kernel
    static constexpr bool V3T = (NSEG >= 4);                   // k_qp3 instantiated (num_seg 4: diagnostics, MPCMP_QP13=3)
    static constexpr int N3 = V3T ? NSEG : 6, N2 = V2C ? NSEG : 4, N1 = (NSEG >= 6) ? 1 : NSEG;
    bool V3C, V4, V2, V5;
    explicit KSel(const mpcmp_ctx *ctx) {
        V3C = NSEG >= 6 || (NSEG == 4 && ctx->qp13 == 3);      // k_qp3: T bordered out, E-free interior solve (N = 19, 25)
        V4 = NSEG == 4 && ctx->qp13 == 4;                       // k_qp3f<4, 1, 4> + k_qp4: two OCPs per CU
        V2 = V2C && !ctx->force_v1 && !V3C && !V4;
        V5 = NSEG == 6 && ctx->qp19 == 5;
    }
};

// dynamic-LDS limits of every kernel a context of this discretisation can launch: set ONCE, in mpcmp_create
template <int NSEG>
static int prepare_impl(mpcmp_ctx *ctx) {
    using S = KSel<NSEG>;
    const S k(ctx);
    if (int rc = set_lds(ctx, k_init<NSEG>, InitLds<NSEG>::size * sizeof(double))) return rc;
    if (int rc = set_lds(ctx, k_step<NSEG>, StepLds<NSEG>::size * sizeof(double))) return rc;
    if (k.V3C) {
        if (int rc = set_lds(ctx, k_qp3<S::N3, 1>, Qp3<S::N3>::sizeL * sizeof(double))) return rc;
        if (int rc = set_lds(ctx, k_qp3f<S::N3, 1>, Qp3<S::N3>::sizeF * sizeof(double))) return rc;
    }
    if (k.V5) {
        if (int rc = set_lds(ctx, k_qp5<6>, Qp5<6>::size5 * sizeof(double))) return rc;
        if (int rc = set_lds(ctx, k_qp3f<6, 1, 5>, Qp3<6>::sizeF * sizeof(double))) return rc;
    }
#ifdef MPCMP_WITH_QP4
    if (k.V4) {
        const size_t l_qp4 = Qp4<4>::size * sizeof(double);
        if (int rc = set_lds(ctx, k_qp4<4>, l_qp4)) return rc;
        if (int rc = set_lds(ctx, k_qp3f<4, 1, 4>, Qp3<4>::sizeF * sizeof(double))) return rc;
        if (ctx->debug_occ) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_qp4<4>, 384, l_qp4);
            std::fprintf(stderr, "k_qp4: %zu B of dynamic LDS, %d workgroups per CU\n", l_qp4, nb);
        }
    }
#endif
    if (k.V2) { if (int rc = set_lds(ctx, k_qp2<S::N2>, Qp2<S::N2>::size * sizeof(double))) return rc; }
    else if (!k.V3C && !k.V4) { if (int rc = set_lds(ctx, k_qp<S::N1>, QpLds<S::N1>::size * sizeof(double))) return rc; }
    return MPCMP_OK;
}

template <int NSEG>
static int solve_impl(mpcmp_ctx *ctx, int B, const double *d_x0, const double *d_xf, const double *d_wx,
                      const double *d_wu, const double *d_wT, double *d_sx, double *d_su, double *d_sT,
                      mpcmp_info *d_info, hipStream_t st, int only_qp, int reguess = 0) {
    using D = Dim<NSEG>;
    using S = KSel<NSEG>;
    const S ks(ctx);
    const bool V3C = ks.V3C, V4 = ks.V4, V2 = ks.V2;
    const mpcmp_config cfg = run_config(ctx);
    WS w = ctx->ws;
    w.x0 = d_x0; w.xf = d_xf;
    w.retired = ctx->rh_step ? ctx->d_retired : nullptr;
    if (ctx->rh_step) { w.alt_x = ctx->d_wx; w.alt_u = ctx->d_wu; w.alt_T = ctx->d_wT; }
    const size_t l_init = InitLds<NSEG>::size * sizeof(double), l_qp = QpLds<S::N1>::size * sizeof(double),
                 l_step = StepLds<NSEG>::size * sizeof(double);
    const size_t l_qp3 = Qp3<S::N3>::sizeL * sizeof(double), l_qp3f = Qp3<S::N3>::sizeF * sizeof(double);
#ifdef MPCMP_WITH_QP4
    const size_t l_qp4 = Qp4<4>::size * sizeof(double);
#endif
    const size_t l_qp2 = Qp2<S::N2>::size * sizeof(double);
    // A large batch is solved as two half-batches on two streams.  Every SQP iteration is a chain of dependent launches
    // (QP -> order -> step), and a QP launch ends in a tail in which most CUs are idle (its problems run 25..700 ADMM
    // iterations, four workgroups per CU); with two independent chains in flight the tail of one half is filled by the other
    // half's next launch.  Replay of the bench workload's iteration counts: -5.8 % makespan.  Results are unaffected (problems
    // are independent); small batches stay on one stream.
    const bool dual = !ctx->single_stream && !only_qp && B >= 512;      // (also under stream capture: the fork/join events carry the other streams into the graph)
    // parts: two; three for the N = 13 kernel from 1024 problems on (round 5, three repeats each: 15.74 - 15.77 k against 15.58 - 15.61 k traj/s on the
    // headline batch; 512 problems — the receding-horizon step — and N = 19 are faster with two: 135.7 k vs 130.0 k re-solves/s, 67.1 k vs 65.2 k traj/s)
    const int parts_auto = (NSEG == 4 && V2 && B >= 1024) ? 3 : 2;
    const int nhalf = dual ? (ctx->parts_env < 1 ? parts_auto : (ctx->parts_env > 4 ? 4 : ctx->parts_env)) : 1;
    int Bh[4] = {0, 0, 0, 0}, boff[4] = {0, 0, 0, 0};
    for (int h = 0, acc = 0; h < nhalf; h++) { Bh[h] = (B - acc + (nhalf - h) - 1) / (nhalf - h); boff[h] = acc; acc += Bh[h]; }
    hipStream_t sh[4] = {st, ctx->stream_x[0], ctx->stream_x[1], ctx->stream_x[2]};
    WS wh[4];
    const double *hx[4] = {nullptr, nullptr, nullptr, nullptr}, *hu[4] = {nullptr, nullptr, nullptr, nullptr}, *hT[4] = {nullptr, nullptr, nullptr, nullptr};
    double *ox[4], *ou[4], *oT[4];
    mpcmp_info *oi[4];
    for (int h = 0; h < nhalf; h++) {
        const size_t b0 = (size_t)boff[h];
        WS v = w;
        v.x0 += 14 * b0; v.xf += 14 * b0; v.z += D::n * b0; v.lam += D::mn * b0; v.ceq += D::meq * b0; v.g += 8 * D::N * b0;
        v.Gk += (size_t)D::N * 176 * b0; v.p += D::n * b0; v.y += D::mn * b0; v.qpit += b0; v.perm += b0; v.okey += b0; v.done += h;
        v.qp_total += b0; v.status += b0; v.alpha += b0; v.dbg += (size_t)MPCMP_DBG_WORDS * b0;
        if (v.retired) v.retired += b0;
        if (v.alt_x) { v.alt_x += 14 * D::N * b0; v.alt_u += 7 * D::N * b0; v.alt_T += b0; }
        wh[h] = v;
        if (d_wx) { hx[h] = d_wx + 14 * D::N * b0; hu[h] = d_wu + 7 * D::N * b0; hT[h] = d_wT + b0; }
        ox[h] = d_sx ? d_sx + 14 * D::N * b0 : nullptr; ou[h] = d_su ? d_su + 7 * D::N * b0 : nullptr;
        oT[h] = d_sT ? d_sT + b0 : nullptr; oi[h] = d_info ? d_info + b0 : nullptr;
    }
    if (nhalf > 1) { HIPCHK(ctx, hipEventRecord(ctx->ev_fork, st)); for (int h = 1; h < nhalf; h++) HIPCHK(ctx, hipStreamWaitEvent(sh[h], ctx->ev_fork, 0)); }
    for (int h = 0; h < nhalf; h++)
        hipLaunchKernelGGL(k_init<NSEG>, dim3(Bh[h]), dim3(D::NT), l_init, sh[h], cfg, wh[h], hx[h], hu[h], hT[h], reguess);
    const int iters = only_qp ? 1 : cfg.sqp_iters;
    for (int it = 0; it < iters; it++) {
        for (int h = 0; h < nhalf; h++) {
            hipEvent_t *ev = (ctx->timing && !ctx->capturing) ? next_events(ctx) : nullptr;
            if (ev) HIPCHK(ctx, hipEventRecord(ev[0], sh[h]));
#ifdef MPCMP_WITH_QP4
            if (V4) {
                WS w3 = wh[h];
                w3.ext_of_int = ctx->d3_ext_of_int; w3.entry_ptr = ctx->d3_entry_ptr; w3.terms = ctx->d3_terms;
                double *fh = ctx->d_fac + (size_t)boff[h] * Qp4Fac<4>::FAC;
                hipLaunchKernelGGL((k_qp3f<4, 1, 4>), dim3(Bh[h]), dim3(1024), Qp3<4>::sizeF * sizeof(double), sh[h], cfg, w3, ctx->d_pat, ctx->xch, Bh[h], fh);
                hipLaunchKernelGGL((k_qp4<4>), dim3(Bh[h]), dim3(384), l_qp4, sh[h], cfg, w3, (const uint32_t *)ctx->d_lane4, Bh[h], (const double *)fh);
            }
#else
            if (V4) { ctx->err = "k_qp4 not built"; return MPCMP_EINVAL; }
#endif
            else if (ks.V5) {
                double *fh = ctx->d_fac + (size_t)boff[h] * Qp5Fac<6>::FAC;
                hipLaunchKernelGGL((k_qp3f<6, 1, 5>), dim3(Bh[h]), dim3(1024), Qp3<6>::sizeF * sizeof(double), sh[h], cfg, wh[h], ctx->d_pat, ctx->xch, Bh[h], fh);
                hipLaunchKernelGGL((k_qp5<6>), dim3(Bh[h]), dim3(768), Qp5<6>::size5 * sizeof(double), sh[h], cfg, wh[h], ctx->d_pat, ctx->xch, Bh[h], (const double *)fh);
            }
            else if (V3C) {
                WS w3 = wh[h];
                if (NSEG == 4) { w3.ext_of_int = ctx->d3_ext_of_int; w3.entry_ptr = ctx->d3_entry_ptr; w3.terms = ctx->d3_terms; }
                double *fh = ctx->d_fac + (size_t)boff[h] * Qp3<S::N3>::FAC;
                hipLaunchKernelGGL((k_qp3f<S::N3, 1>), dim3(Bh[h]), dim3(1024), l_qp3f, sh[h], cfg, w3, ctx->d_pat, ctx->xch, Bh[h], fh);
                hipLaunchKernelGGL((k_qp3<S::N3, 1>), dim3(Bh[h]), dim3(512), l_qp3, sh[h], cfg, w3, ctx->d_pat, ctx->xch, Bh[h], (const double *)fh);
            }
            else if (V2) hipLaunchKernelGGL((k_qp2<S::N2>), dim3(Bh[h]), dim3(1024), l_qp2, sh[h], cfg, wh[h], ctx->streams);
            else hipLaunchKernelGGL((k_qp<S::N1>), dim3(Bh[h]), dim3(Dim<S::N1>::NT), l_qp, sh[h], cfg, wh[h]);
            if (ev) HIPCHK(ctx, hipEventRecord(ev[1], sh[h]));
            if (only_qp) continue;
            hipLaunchKernelGGL(k_step<NSEG>, dim3(Bh[h]), dim3(D::NT), l_step, sh[h], cfg, wh[h], it == iters - 1 ? 1 : 0, it,
                               ox[h], ou[h], oT[h], oi[h]);
        }
    }
    for (int h = 1; h < nhalf; h++) { HIPCHK(ctx, hipEventRecord(ctx->ev_join[h - 1], sh[h])); HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_join[h - 1], 0)); }
    HIPCHK(ctx, hipGetLastError());
    return MPCMP_OK;
}

template <int NSEG, int NARM>
static int prepare_impl_m(mpcmp_ctx *ctx) {
    using D = DimM<NSEG, NARM>;
    const size_t l_m = D::size * sizeof(double);
    if (int rc = set_lds(ctx, k_init_m<NSEG, NARM>, l_m)) return rc;
    if (int rc = set_lds(ctx, k_step_m<NSEG, NARM>, l_m)) return rc;
    if (int rc = set_lds(ctx, k_qp3<NSEG, NARM>, Qp3<NSEG>::sizeL * sizeof(double))) return rc;
    if (int rc = set_lds(ctx, k_qp3f<NSEG, NARM>, Qp3<NSEG>::sizeF * sizeof(double))) return rc;
    return MPCMP_OK;
}

// dynamic-LDS limits of the context's kernels (hipFuncSetAttribute): once per context
static int prepare_kernels(mpcmp_ctx *ctx) {
    if (ctx->narm == 2) return ctx->nseg == 6 ? prepare_impl_m<6, 2>(ctx) : prepare_impl_m<8, 2>(ctx);
    switch (ctx->nseg) {
        case 1: return prepare_impl<1>(ctx);
        case 2: return prepare_impl<2>(ctx);
        case 4: return prepare_impl<4>(ctx);
        case 6: return prepare_impl<6>(ctx);
        case 8: return prepare_impl_m<8, 1>(ctx);
    }
    return MPCMP_EINVAL;
}

// N = 25 and multi-arm OCPs: k_init_m -> K x [k_qp3 (one workgroup per arm), k_step_m]
template <int NSEG, int NARM>
static int solve_impl_m(mpcmp_ctx *ctx, int B, const double *d_x0, const double *d_xf, const double *d_wx, const double *d_wu,
                        const double *d_wT, double *d_sx, double *d_su, double *d_sT, mpcmp_info *d_info, hipStream_t st,
                        int only_qp, int reguess) {
    using D = DimM<NSEG, NARM>;
    constexpr int N = D::N;
    const mpcmp_config cfg = run_config(ctx);
    WS w = ctx->ws;
    w.x0 = d_x0; w.xf = d_xf;
    w.retired = ctx->rh_step ? ctx->d_retired : nullptr;
    if (ctx->rh_step) { w.alt_x = ctx->d_wx; w.alt_u = ctx->d_wu; w.alt_T = ctx->d_wT; }
    const size_t l_m = D::size * sizeof(double), l_qp3 = Qp3<NSEG>::sizeL * sizeof(double), l_qp3f = Qp3<NSEG>::sizeF * sizeof(double);
    // two parts on two streams once one part alone fills the chip (one CU per arm): see solve_impl
    const bool dual = !ctx->single_stream && !only_qp && B * NARM >= 512;
    const int nhalf = dual ? 2 : 1;
    int Bh[2] = {0, 0}, boff[2] = {0, 0};
    for (int h = 0, acc = 0; h < nhalf; h++) { Bh[h] = (B - acc + (nhalf - h) - 1) / (nhalf - h); boff[h] = acc; acc += Bh[h]; }
    hipStream_t sh[2] = {st, ctx->stream_x[0]};
    WS wh[2];
    Xch xh[2];
    const double *hx[2] = {nullptr, nullptr}, *hu[2] = {nullptr, nullptr}, *hT[2] = {nullptr, nullptr};
    double *ox[2], *ou[2], *oT[2];
    mpcmp_info *oi[2];
    for (int h = 0; h < nhalf; h++) {
        const size_t b0 = (size_t)boff[h];
        WS v = w;
        v.x0 += 14 * NARM * b0; v.xf += 14 * NARM * b0; v.z += D::n * b0; v.lam += D::mn * b0; v.ceq += (size_t)NARM * D::meq * b0;
        v.g += (size_t)NARM * 8 * N * b0; v.Gk += (size_t)NARM * N * 176 * b0; v.p += D::n * b0; v.y += D::mn * b0; v.qpit += b0; v.perm += b0;
        v.okey += b0; v.done += h; v.qp_total += b0; v.status += b0; v.alpha += b0; v.dbg += (size_t)MPCMP_DBG_WORDS * b0;
        if (v.retired) v.retired += b0;
        if (v.alt_x) { v.alt_x += (size_t)14 * NARM * N * b0; v.alt_u += (size_t)7 * NARM * N * b0; v.alt_T += b0; }
        wh[h] = v;
        xh[h].buf = ctx->xch.buf ? ctx->xch.buf + (size_t)2 * MPCMP_XCH_STRIDE * b0 : nullptr;
        if (d_wx) { hx[h] = d_wx + (size_t)14 * NARM * N * b0; hu[h] = d_wu + (size_t)7 * NARM * N * b0; hT[h] = d_wT + b0; }
        ox[h] = d_sx ? d_sx + (size_t)14 * NARM * N * b0 : nullptr; ou[h] = d_su ? d_su + (size_t)7 * NARM * N * b0 : nullptr;
        oT[h] = d_sT ? d_sT + b0 : nullptr; oi[h] = d_info ? d_info + b0 : nullptr;
    }
    if (nhalf > 1) { HIPCHK(ctx, hipEventRecord(ctx->ev_fork, st)); HIPCHK(ctx, hipStreamWaitEvent(sh[1], ctx->ev_fork, 0)); }
    for (int h = 0; h < nhalf; h++)
        hipLaunchKernelGGL((k_init_m<NSEG, NARM>), dim3(Bh[h]), dim3(D::NT), l_m, sh[h], cfg, ctx->d_model, wh[h], xh[h], hx[h], hu[h], hT[h], reguess);
    const int iters = only_qp ? 1 : cfg.sqp_iters;
    for (int it = 0; it < iters; it++) {
        for (int h = 0; h < nhalf; h++) {
            hipEvent_t *ev = (ctx->timing && !ctx->capturing) ? next_events(ctx) : nullptr;
            if (ev) HIPCHK(ctx, hipEventRecord(ev[0], sh[h]));
            const int grid = NARM == 1 ? Bh[h] : ((Bh[h] + 7) / 8) * 16;       // arm workgroups of one OCP are 8 apart (k_qp3)
            double *fh = ctx->d_fac + (size_t)boff[h] * NARM * Qp3<NSEG>::FAC;
            hipLaunchKernelGGL((k_qp3f<NSEG, NARM>), dim3(grid), dim3(1024), l_qp3f, sh[h], cfg, wh[h], ctx->d_pat, xh[h], Bh[h], fh);
            hipLaunchKernelGGL((k_qp3<NSEG, NARM>), dim3(grid), dim3(512), l_qp3, sh[h], cfg, wh[h], ctx->d_pat, xh[h], Bh[h], (const double *)fh);
            if (ev) HIPCHK(ctx, hipEventRecord(ev[1], sh[h]));
            if (only_qp) continue;
            hipLaunchKernelGGL((k_step_m<NSEG, NARM>), dim3(Bh[h]), dim3(D::NT), l_m, sh[h], cfg, ctx->d_model, wh[h], xh[h],
                               it == iters - 1 ? 1 : 0, it, ox[h], ou[h], oT[h], oi[h]);
        }
    }
    if (nhalf > 1) { HIPCHK(ctx, hipEventRecord(ctx->ev_join[0], sh[1])); HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_join[0], 0)); }
    HIPCHK(ctx, hipGetLastError());
    return MPCMP_OK;
}

static int solve_dispatch(mpcmp_ctx *ctx, int B, const double *d_x0, const double *d_xf, const double *d_wx,
                          const double *d_wu, const double *d_wT, double *d_sx, double *d_su, double *d_sT,
                          mpcmp_info *d_info, hipStream_t st, int only_qp, int reguess = 0) {
    if (!ctx) return MPCMP_EINVAL;
    if (B < 1) return MPCMP_EINVAL;
    if (B > ctx->max_batch) { ctx->err = "batch exceeds the context capacity"; return MPCMP_ETOOBIG; }
    if (!d_x0 || !d_xf) return MPCMP_EINVAL;
    if ((d_wx != nullptr) != (d_wu != nullptr) || (d_wx != nullptr) != (d_wT != nullptr)) { ctx->err = "warm_x, warm_u, warm_T must be all given or all NULL"; return MPCMP_EINVAL; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->narm == 2) {
        if (only_qp) { ctx->err = "mpcmp_qp_batch is defined for single-arm contexts only"; return MPCMP_EINVAL; }
        if (!d_wx) { ctx->err = "multi-arm solves need a warm start (mpcmp_warm_start_jerk_batch)"; return MPCMP_EINVAL; }
        if (ctx->nseg == 6) return solve_impl_m<6, 2>(ctx, B, d_x0, d_xf, d_wx, d_wu, d_wT, d_sx, d_su, d_sT, d_info, st, only_qp, reguess);
        return solve_impl_m<8, 2>(ctx, B, d_x0, d_xf, d_wx, d_wu, d_wT, d_sx, d_su, d_sT, d_info, st, only_qp, reguess);
    }
    switch (ctx->nseg) {
        case 1: return solve_impl<1>(ctx, B, d_x0, d_xf, d_wx, d_wu, d_wT, d_sx, d_su, d_sT, d_info, st, only_qp, reguess);
        case 2: return solve_impl<2>(ctx, B, d_x0, d_xf, d_wx, d_wu, d_wT, d_sx, d_su, d_sT, d_info, st, only_qp, reguess);
        case 4: return solve_impl<4>(ctx, B, d_x0, d_xf, d_wx, d_wu, d_wT, d_sx, d_su, d_sT, d_info, st, only_qp, reguess);
        case 6: return solve_impl<6>(ctx, B, d_x0, d_xf, d_wx, d_wu, d_wT, d_sx, d_su, d_sT, d_info, st, only_qp, reguess);
        case 8: return solve_impl_m<8, 1>(ctx, B, d_x0, d_xf, d_wx, d_wu, d_wT, d_sx, d_su, d_sT, d_info, st, only_qp, reguess);
    }
    return MPCMP_EINVAL;
}

extern "C" int mpcmp_solve_batch_device(mpcmp_ctx *ctx, int B, const double *d_x0, const double *d_xf,
                                        const double *d_wx, const double *d_wu, const double *d_wT, double *d_sx,
                                        double *d_su, double *d_sT, mpcmp_info *d_info, void *hip_stream) {
    if (!d_sx || !d_su || !d_sT) return MPCMP_EINVAL;
    return solve_dispatch(ctx, B, d_x0, d_xf, d_wx, d_wu, d_wT, d_sx, d_su, d_sT, d_info, (hipStream_t)hip_stream, 0);
}

extern "C" int mpcmp_solve_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *wx,
                                 const double *wu, const double *wT, double *sx, double *su, double *sT, mpcmp_info *info) {
    if (!ctx || !x0 || !xf || !sx || !su || !sT) return MPCMP_EINVAL;
    if (B < 1) return MPCMP_EINVAL;
    if (B > ctx->max_batch) { ctx->err = "batch exceeds the context capacity"; return MPCMP_ETOOBIG; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t N = ctx->N, nx = ctx->nx, nu = ctx->nu;
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_x0, x0, sizeof(double) * nx * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_xf, xf, sizeof(double) * nx * B, hipMemcpyHostToDevice, st));
    const bool warm = wx && wu && wT;
    if (warm) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_wx, wx, sizeof(double) * nx * N * B, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_wu, wu, sizeof(double) * nu * N * B, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_wT, wT, sizeof(double) * B, hipMemcpyHostToDevice, st));
    } else if (wx || wu || wT) { ctx->err = "warm_x, warm_u, warm_T must be all given or all NULL"; return MPCMP_EINVAL; }
    if (int rc = solve_dispatch(ctx, B, ctx->d_x0, ctx->d_xf, warm ? ctx->d_wx : nullptr, warm ? ctx->d_wu : nullptr,
                                warm ? ctx->d_wT : nullptr, ctx->d_sx, ctx->d_su, ctx->d_sT, ctx->d_info, st, 0))
        return rc;
    HIPCHK(ctx, hipMemcpyAsync(sx, ctx->d_sx, sizeof(double) * nx * N * B, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(su, ctx->d_su, sizeof(double) * nu * N * B, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(sT, ctx->d_sT, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    if (info) HIPCHK(ctx, hipMemcpyAsync(info, ctx->d_info, sizeof(mpcmp_info) * B, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return MPCMP_OK;
}

#define SINGLE_ARM_ONLY(ctx) do { if ((ctx)->narm != 1) { (ctx)->err = "this entry point is defined for single-arm contexts only"; return MPCMP_EINVAL; } } while (0)

extern "C" int mpcmp_warm_start_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, double *wx, double *wu, double *wT) {
    if (!ctx || !x0 || !xf || !wx || !wu || !wT || B < 1) return MPCMP_EINVAL;
    SINGLE_ARM_ONLY(ctx);
    if (B > ctx->max_batch) return MPCMP_ETOOBIG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t N = ctx->N, n = ctx->n;
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_x0, x0, sizeof(double) * 14 * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_xf, xf, sizeof(double) * 14 * B, hipMemcpyHostToDevice, st));
    // k_init alone: the built-in initialiser writes the iterate z = [xs|us|T]
    const mpcmp_config save = run_config(ctx);
    WS w = ctx->ws; w.x0 = ctx->d_x0; w.xf = ctx->d_xf;
#define LAUNCH_INIT(NS) { const size_t l = InitLds<NS>::size * sizeof(double); \
        hipLaunchKernelGGL(k_init<NS>, dim3(B), dim3(Dim<NS>::NT), l, st, save, w, (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, 0); }
    switch (ctx->nseg) { case 1: LAUNCH_INIT(1) break; case 2: LAUNCH_INIT(2) break; case 4: LAUNCH_INIT(4) break; case 6: LAUNCH_INIT(6) break;
        case 8: { const size_t l = DimM<8, 1>::size * sizeof(double);
                  hipLaunchKernelGGL((k_init_m<8, 1>), dim3(B), dim3(512), l, st, save, ctx->d_model, w, ctx->xch, (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, 0); } break; }
#undef LAUNCH_INIT
    HIPCHK(ctx, hipGetLastError());
    std::vector<double> z((size_t)B * n);
    HIPCHK(ctx, hipMemcpyAsync(z.data(), ctx->ws.z, sizeof(double) * n * B, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    for (int b = 0; b < B; b++) {
        std::memcpy(wx + (size_t)b * 14 * N, z.data() + (size_t)b * n, sizeof(double) * 14 * N);
        std::memcpy(wu + (size_t)b * 7 * N, z.data() + (size_t)b * n + 14 * N, sizeof(double) * 7 * N);
        wT[b] = z[(size_t)b * n + 21 * N];
    }
    return MPCMP_OK;
}

// Device scratch of the host-buffer leaf entry points, owned by the context and only ever grown: after the first call of a
// given size these entry points do not allocate (a control loop calls get_MPC_point / get_RK_point every tick).
struct TmpBuf {
    mpcmp_ctx *ctx;
    size_t off = 0;
    bool ok = true;
    TmpBuf(mpcmp_ctx *c, size_t total_bytes) : ctx(c) {
        total_bytes += 256 * 16;                       // alignment slack for up to 16 sub-buffers
        if (total_bytes > c->scratch_bytes) {
            if (c->scratch) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
            void *q = nullptr;
            const size_t want = total_bytes < (1u << 20) ? (1u << 20) : total_bytes;
            if (hipMalloc(&q, want) != hipSuccess) { ok = false; return; }
            c->scratch = q; c->scratch_bytes = want;
        }
    }
    template <typename T> T *get(size_t count) {
        if (!ok) return nullptr;
        off = (off + 255) / 256 * 256;
        T *r = reinterpret_cast<T *>(static_cast<char *>(ctx->scratch) + off);
        off += count * sizeof(T);
        if (off > ctx->scratch_bytes) { ok = false; return nullptr; }
        return r;
    }
};

// ---- jerk-limited, time-synchronised warm start / comparison trajectory (stands in for Ruckig) ----
// vmax / amax: NULL = the context's margin-applied bounds (the *_lim_* entry points pass a caller's own: ruckig::InputParameter::max_velocity / max_acceleration)
static int jerk_limits(mpcmp_ctx *ctx, const double *vmax, const double *amax, const double *jmax, JerkLimits &lim) {
    for (int j = 0; j < 7; j++) {
        lim.v[j] = vmax ? vmax[j] : ctx->cfg.ubx[7 + j]; lim.a[j] = amax ? amax[j] : ctx->cfg.ubu[j]; lim.j[j] = jmax[j];
        if (!(lim.v[j] > 0.0) || !(lim.a[j] > 0.0) || !(lim.j[j] > 0.0)) { ctx->err = "velocity, acceleration and jerk limits must be positive"; return MPCMP_EINVAL; }
    }
    return MPCMP_OK;
}

// boundary accelerations d_acc0 / d_accT: device [B][7], either may be NULL (= zero); single-arm contexts only (a multi-arm warm start has none)
static int warm_start_jerk_device(mpcmp_ctx *ctx, int B, const double *d_x0, const double *d_xf, const double *d_acc0, const double *d_accT,
                                  const double *vmax, const double *amax, const double *jmax, double *d_wx, double *d_wu, double *d_wT, void *hip_stream,
                                  const int *need_status = nullptr, int need_mask = 0) {
    if (!ctx || !d_x0 || !d_xf || !jmax || !d_wx || !d_wu || !d_wT || B < 1) return MPCMP_EINVAL;
    if ((d_acc0 || d_accT) && ctx->narm != 1) { ctx->err = "boundary accelerations: single-arm contexts only"; return MPCMP_EINVAL; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    JerkLimits lim;
    if (int rc = jerk_limits(ctx, vmax, amax, jmax, lim)) return rc;
    if (ctx->narm == 2) {
        // per-arm generator on B * narm arm problems, then the merge to the common (slowest arm's) duration: multi_kernels.hpp
        if (B > ctx->max_batch) return MPCMP_ETOOBIG;
        hipStream_t s2 = (hipStream_t)hip_stream;
        const int cnt = B * 2 * 14;
        hipLaunchKernelGGL((k_split_states<2>), dim3((cnt + 255) / 256), dim3(256), 0, s2, B, d_x0, ctx->d_ax0);
        hipLaunchKernelGGL((k_split_states<2>), dim3((cnt + 255) / 256), dim3(256), 0, s2, B, d_xf, ctx->d_axf);
        hipLaunchKernelGGL(k_warm_jerk, dim3(B * 2), dim3(64), 0, s2, ctx->nseg, lim, ctx->d_ax0, ctx->d_axf, (const double *)nullptr, (const double *)nullptr, ctx->d_awx, ctx->d_awu, ctx->d_awT,
                           need_status, need_mask, 2);
        hipLaunchKernelGGL((k_warm_merge<2>), dim3(B), dim3(256), 0, s2, ctx->N, B, ctx->d_awx, ctx->d_awu, ctx->d_awT, d_x0, d_xf, d_wx, d_wu, d_wT);
        HIPCHK(ctx, hipGetLastError());
        return MPCMP_OK;
    }
    // the stream exactly as given (NULL = the legacy default stream), like the other *_device entry points: the solve that
    // consumes the warm start is enqueued on the same stream and is ordered behind this launch
    hipLaunchKernelGGL(k_warm_jerk, dim3(B), dim3(64), 0, (hipStream_t)hip_stream, ctx->nseg, lim, d_x0, d_xf, d_acc0, d_accT, d_wx, d_wu, d_wT, need_status, need_mask, 1);
    HIPCHK(ctx, hipGetLastError());
    return MPCMP_OK;
}
extern "C" int mpcmp_warm_start_jerk_acc_batch_device(mpcmp_ctx *ctx, int B, const double *d_x0, const double *d_xf, const double *d_acc0, const double *d_accT,
                                                      const double *jmax, double *d_wx, double *d_wu, double *d_wT, void *hip_stream) {
    return warm_start_jerk_device(ctx, B, d_x0, d_xf, d_acc0, d_accT, nullptr, nullptr, jmax, d_wx, d_wu, d_wT, hip_stream);
}
extern "C" int mpcmp_warm_start_jerk_batch_device(mpcmp_ctx *ctx, int B, const double *d_x0, const double *d_xf, const double *jmax,
                                                  double *d_wx, double *d_wu, double *d_wT, void *hip_stream) {
    return mpcmp_warm_start_jerk_acc_batch_device(ctx, B, d_x0, d_xf, nullptr, nullptr, jmax, d_wx, d_wu, d_wT, hip_stream);
}

// Boundary states the generator cannot honour: the acceleration limit yields to a boundary acceleration above it (include/mpcmp.h), but an
// acceleration that cannot be brought to zero inside the VELOCITY limit — |v0 + a0 |a0| / 2J| > V at the start, |vT - aT |aT| / 2J| > V at the target —
// would overshoot the state box of the OCP: MPCMP_EINVAL, as Ruckig rejects such an input (ErrorInvalidInput; the shim's otg.calculate returns it).
// Host-pointer entry points only: the *_device variants never read their inputs on the host.
static int check_boundary_acc(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *acc0, const double *accT,
                              const double *vmax, const double *jmax) {
    if ((!acc0 && !accT) || ctx->narm != 1) return MPCMP_OK;
    for (int b = 0; b < B; b++)
        for (int j = 0; j < 7; j++) {
            const double V = (vmax ? vmax[j] : ctx->cfg.ubx[7 + j]) * (1.0 + 1e-9), J = jmax[j];
            if (!(J > 0.0)) continue;      // (reported by jerk_limits)
            const double a0 = acc0 ? acc0[7 * b + j] : 0.0, aT = accT ? accT[7 * b + j] : 0.0;
            const double v0 = x0[14 * b + 7 + j], vT = xf[14 * b + 7 + j];
            const double va = v0 + a0 * std::fabs(a0) / (2.0 * J), vb = vT - aT * std::fabs(aT) / (2.0 * J);
            // (a boundary VELOCITY outside the limit is the caller's business, as before: that joint falls back to the quintic of the common duration)
            if ((std::fabs(v0) <= V && std::fabs(va) > V) || (std::fabs(vT) <= V && std::fabs(vb) > V)) {
                std::ostringstream os;
                os << "problem " << b << ", joint " << j << ": the boundary acceleration cannot be brought to zero inside the velocity limit";
                ctx->err = os.str();
                return MPCMP_EINVAL;
            }
        }
    return MPCMP_OK;
}

// host [B][7] boundary accelerations (either may be NULL) staged next to the states; returns the device pointers (or NULL)
static int stage_acc(mpcmp_ctx *ctx, TmpBuf &tb, int B, const double *acc0, const double *accT, hipStream_t st, const double **d0, const double **dT) {
    *d0 = *dT = nullptr;
    if (!acc0 && !accT) return MPCMP_OK;
    if (ctx->narm != 1) { ctx->err = "boundary accelerations: single-arm contexts only"; return MPCMP_EINVAL; }
    double *buf = tb.get<double>(14 * (size_t)B);
    if (!buf) { ctx->err = "hipMalloc failed"; return MPCMP_ERUNTIME; }
    if (acc0) { HIPCHK(ctx, hipMemcpyAsync(buf, acc0, sizeof(double) * 7 * B, hipMemcpyHostToDevice, st)); *d0 = buf; }
    if (accT) { HIPCHK(ctx, hipMemcpyAsync(buf + 7 * (size_t)B, accT, sizeof(double) * 7 * B, hipMemcpyHostToDevice, st)); *dT = buf + 7 * (size_t)B; }
    return MPCMP_OK;
}

extern "C" int mpcmp_warm_start_jerk_lim_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *acc0, const double *accT,
                                               const double *vmax, const double *amax, const double *jmax, double *wx, double *wu, double *wT) {
    if (!ctx || !x0 || !xf || !jmax || !wx || !wu || !wT || B < 1) return MPCMP_EINVAL;
    if (B > ctx->max_batch) return MPCMP_ETOOBIG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t N = ctx->N, nx = ctx->nx, nu = ctx->nu;
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_x0, x0, sizeof(double) * nx * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_xf, xf, sizeof(double) * nx * B, hipMemcpyHostToDevice, st));
    TmpBuf tb(ctx, (acc0 || accT) ? 14 * (size_t)B * sizeof(double) : 0);
    const double *da0, *daT;
    if (int rc = check_boundary_acc(ctx, B, x0, xf, acc0, accT, vmax, jmax)) return rc;
    if (int rc = stage_acc(ctx, tb, B, acc0, accT, st, &da0, &daT)) return rc;
    if (int rc = warm_start_jerk_device(ctx, B, ctx->d_x0, ctx->d_xf, da0, daT, vmax, amax, jmax, ctx->d_wx, ctx->d_wu, ctx->d_wT, st)) return rc;
    HIPCHK(ctx, hipMemcpyAsync(wx, ctx->d_wx, sizeof(double) * nx * N * B, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(wu, ctx->d_wu, sizeof(double) * nu * N * B, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(wT, ctx->d_wT, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return MPCMP_OK;
}

extern "C" int mpcmp_warm_start_jerk_acc_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *acc0, const double *accT,
                                               const double *jmax, double *wx, double *wu, double *wT) {
    return mpcmp_warm_start_jerk_lim_batch(ctx, B, x0, xf, acc0, accT, nullptr, nullptr, jmax, wx, wu, wT);
}
extern "C" int mpcmp_warm_start_jerk_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *jmax,
                                           double *wx, double *wu, double *wT) {
    return mpcmp_warm_start_jerk_acc_batch(ctx, B, x0, xf, nullptr, nullptr, jmax, wx, wu, wT);
}

extern "C" int mpcmp_jerk_trajectory_lim_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *acc0, const double *accT,
                                               const double *vmax, const double *amax, const double *jmax, int n_pts, double *out, double *T_out) {
    if (!ctx || !x0 || !xf || !jmax || !out || B < 1 || n_pts < 1) return MPCMP_EINVAL;
    SINGLE_ARM_ONLY(ctx);
    if (B > ctx->max_batch) return MPCMP_ETOOBIG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    JerkLimits lim;
    if (int rc = jerk_limits(ctx, vmax, amax, jmax, lim)) return rc;
    const size_t cnt = (size_t)B * (n_pts + 1) * 22;
    TmpBuf tb(ctx, (cnt + 14 * (size_t)B) * sizeof(double));
    double *dout = tb.get<double>(cnt);
    if (!dout) { ctx->err = "hipMalloc failed"; return MPCMP_ERUNTIME; }
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_x0, x0, sizeof(double) * 14 * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_xf, xf, sizeof(double) * 14 * B, hipMemcpyHostToDevice, st));
    const double *da0, *daT;
    if (int rc = check_boundary_acc(ctx, B, x0, xf, acc0, accT, vmax, jmax)) return rc;
    if (int rc = stage_acc(ctx, tb, B, acc0, accT, st, &da0, &daT)) return rc;
    hipLaunchKernelGGL(k_jerk_traj, dim3(B), dim3(64), 0, st, lim, ctx->d_x0, ctx->d_xf, da0, daT, n_pts, dout, ctx->d_wT);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, dout, sizeof(double) * cnt, hipMemcpyDeviceToHost, st));
    if (T_out) HIPCHK(ctx, hipMemcpyAsync(T_out, ctx->d_wT, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return MPCMP_OK;
}

extern "C" int mpcmp_jerk_trajectory_acc_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *acc0, const double *accT,
                                               const double *jmax, int n_pts, double *out, double *T_out) {
    return mpcmp_jerk_trajectory_lim_batch(ctx, B, x0, xf, acc0, accT, nullptr, nullptr, jmax, n_pts, out, T_out);
}
extern "C" int mpcmp_jerk_trajectory_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *jmax, int n_pts,
                                           double *out, double *T_out) {
    return mpcmp_jerk_trajectory_acc_batch(ctx, B, x0, xf, nullptr, nullptr, jmax, n_pts, out, T_out);
}

// MotionPlanner::get_RK_point (motionPlanner.hpp:130-142)
extern "C" int mpcmp_jerk_point_lim_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *acc0, const double *accT,
                                          const double *vmax, const double *amax, const double *jmax, const double *time, double *out, double *T_out) {
    if (!ctx || !x0 || !xf || !jmax || !time || !out || B < 1) return MPCMP_EINVAL;
    SINGLE_ARM_ONLY(ctx);
    if (B > ctx->max_batch) return MPCMP_ETOOBIG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    JerkLimits lim;
    if (int rc = jerk_limits(ctx, vmax, amax, jmax, lim)) return rc;
    TmpBuf tb(ctx, (29 + 14) * (size_t)B * sizeof(double));
    double *dt = tb.get<double>(B), *dout = tb.get<double>(28 * (size_t)B);
    if (!dt || !dout) { ctx->err = "hipMalloc failed"; return MPCMP_ERUNTIME; }
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_x0, x0, sizeof(double) * 14 * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_xf, xf, sizeof(double) * 14 * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(dt, time, sizeof(double) * B, hipMemcpyHostToDevice, st));
    const double *da0, *daT;
    if (int rc = check_boundary_acc(ctx, B, x0, xf, acc0, accT, vmax, jmax)) return rc;
    if (int rc = stage_acc(ctx, tb, B, acc0, accT, st, &da0, &daT)) return rc;
    hipLaunchKernelGGL(k_jerk_point, dim3(B), dim3(64), 0, st, ctx->d_model, lim, ctx->d_x0, ctx->d_xf, da0, daT, dt, dout, ctx->d_wT);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, dout, sizeof(double) * 28 * B, hipMemcpyDeviceToHost, st));
    if (T_out) HIPCHK(ctx, hipMemcpyAsync(T_out, ctx->d_wT, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return MPCMP_OK;
}

extern "C" int mpcmp_jerk_point_acc_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *acc0, const double *accT,
                                          const double *jmax, const double *time, double *out, double *T_out) {
    return mpcmp_jerk_point_lim_batch(ctx, B, x0, xf, acc0, accT, nullptr, nullptr, jmax, time, out, T_out);
}
extern "C" int mpcmp_jerk_point_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *jmax, const double *time,
                                      double *out, double *T_out) {
    return mpcmp_jerk_point_acc_batch(ctx, B, x0, xf, nullptr, nullptr, jmax, time, out, T_out);
}

// MotionPlanner::get_MPC_point (motionPlanner.hpp:118-128)
extern "C" int mpcmp_mpc_point_batch(mpcmp_ctx *ctx, int B, const double *sx, const double *su, const double *sT, const double *time,
                                     double *out) {
    if (!ctx || !sx || !su || !sT || !time || !out || B < 1) return MPCMP_EINVAL;
    SINGLE_ARM_ONLY(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t N = ctx->N;
    TmpBuf tb(ctx, (21 * N + 2 + 28) * (size_t)B * sizeof(double));
    double *dx = tb.get<double>(14 * N * B), *du = tb.get<double>(7 * N * B), *dT = tb.get<double>(B), *dt = tb.get<double>(B),
           *dout = tb.get<double>(28 * (size_t)B);
    if (!dx || !du || !dT || !dt || !dout) { ctx->err = "hipMalloc failed"; return MPCMP_ERUNTIME; }
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(dx, sx, sizeof(double) * 14 * N * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(du, su, sizeof(double) * 7 * N * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(dT, sT, sizeof(double) * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(dt, time, sizeof(double) * B, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_mpc_point, dim3((B + 63) / 64), dim3(64), 0, st, ctx->d_model, ctx->nseg, B, dx, du, dT, dt, dout);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, dout, sizeof(double) * 28 * B, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return MPCMP_OK;
}

extern "C" int mpcmp_qp_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *xs,
                              const double *us, const double *T, double *p, double *y, int *iters) {
    if (!ctx || !x0 || !xf || !xs || !us || !T || !p || !y || B < 1) return MPCMP_EINVAL;
    SINGLE_ARM_ONLY(ctx);
    if (B > ctx->max_batch) return MPCMP_ETOOBIG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t N = ctx->N;
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_x0, x0, sizeof(double) * 14 * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_xf, xf, sizeof(double) * 14 * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_wx, xs, sizeof(double) * 14 * N * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_wu, us, sizeof(double) * 7 * N * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_wT, T, sizeof(double) * B, hipMemcpyHostToDevice, st));
    if (int rc = solve_dispatch(ctx, B, ctx->d_x0, ctx->d_xf, ctx->d_wx, ctx->d_wu, ctx->d_wT, ctx->d_sx, ctx->d_su,
                                ctx->d_sT, nullptr, st, 1))
        return rc;
    HIPCHK(ctx, hipMemcpyAsync(p, ctx->ws.p, sizeof(double) * ctx->n * B, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(y, ctx->ws.y, sizeof(double) * ctx->mn * B, hipMemcpyDeviceToHost, st));
    if (iters) HIPCHK(ctx, hipMemcpyAsync(iters, ctx->ws.qpit, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return MPCMP_OK;
}

// ------------------------------------------------------------------------------------------------
// leaf kernels
extern "C" int mpcmp_rnea_batch(mpcmp_ctx *ctx, int n, const double *q, const double *qd, const double *qdd, double *tau) {
    if (!ctx || n < 1 || !q || !qd || !qdd || !tau) return MPCMP_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    TmpBuf tb(ctx, 28 * (size_t)n * sizeof(double));
    double *dq = tb.get<double>(7 * (size_t)n), *dv = tb.get<double>(7 * (size_t)n), *da = tb.get<double>(7 * (size_t)n), *dt = tb.get<double>(7 * (size_t)n);
    if (!dq || !dv || !da || !dt) { ctx->err = "hipMalloc failed"; return MPCMP_ERUNTIME; }
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(dq, q, sizeof(double) * 7 * n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(dv, qd, sizeof(double) * 7 * n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(da, qdd, sizeof(double) * 7 * n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_rnea_batch, dim3((n + 63) / 64), dim3(64), 0, st, ctx->d_model, n, dq, dv, da, dt);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(tau, dt, sizeof(double) * 7 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return MPCMP_OK;
}

extern "C" int mpcmp_eval_constraints_batch(mpcmp_ctx *ctx, int n, const double *x, const double *u, double *g, double *G) {
    if (!ctx || n < 1 || !x || !u || !g || !G) return MPCMP_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    TmpBuf tb(ctx, (14 + 7 + 8 + 176) * (size_t)n * sizeof(double));
    double *dx = tb.get<double>(14 * (size_t)n), *du = tb.get<double>(7 * (size_t)n), *dg = tb.get<double>(8 * (size_t)n), *dG = tb.get<double>(176 * (size_t)n);
    if (!dx || !du || !dg || !dG) { ctx->err = "hipMalloc failed"; return MPCMP_ERUNTIME; }
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(dx, x, sizeof(double) * 14 * n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(du, u, sizeof(double) * 7 * n, hipMemcpyHostToDevice, st));
    constexpr int NS = 4;
    const size_t l = EvalLds<NS>::size * sizeof(double);
    if (int rc = set_lds(ctx, k_eval_constraints<NS>, l)) return rc;
    const int N = Dim<NS>::N;
    hipLaunchKernelGGL(k_eval_constraints<NS>, dim3((n + N - 1) / N), dim3(Dim<NS>::NT), l, st, ctx->cfg, (const mpcmp_model *)ctx->d_model, n, dx, du, dg, dG);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(g, dg, sizeof(double) * 8 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(G, dG, sizeof(double) * 176 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return MPCMP_OK;
}

extern "C" int mpcmp_sample_batch_device(mpcmp_ctx *ctx, int B, const double *d_sx, const double *d_su, const double *d_sT,
                                         int n_pts, double *d_out, void *hip_stream) {
    if (!ctx || B < 1 || n_pts < 1 || !d_sx || !d_su || !d_sT || !d_out) return MPCMP_EINVAL;
    SINGLE_ARM_ONLY(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const long total = (long)B * (n_pts + 1);
    hipLaunchKernelGGL(k_sample, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, (hipStream_t)hip_stream, ctx->d_model,
                       ctx->nseg, B, n_pts, d_sx, d_su, d_sT, d_out);
    HIPCHK(ctx, hipGetLastError());
    return MPCMP_OK;
}

extern "C" int mpcmp_sample_batch(mpcmp_ctx *ctx, int B, const double *sx, const double *su, const double *sT, int n_pts, double *out) {
    if (!ctx || B < 1 || n_pts < 1 || !sx || !su || !sT || !out) return MPCMP_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t N = ctx->N, np = (size_t)B * (n_pts + 1) * 29;
    TmpBuf tb(ctx, (21 * N * B + B + np) * sizeof(double));
    double *dx = tb.get<double>(14 * N * B), *du = tb.get<double>(7 * N * B), *dT = tb.get<double>(B), *dout = tb.get<double>(np);
    if (!dx || !du || !dT || !dout) { ctx->err = "hipMalloc failed"; return MPCMP_ERUNTIME; }
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(dx, sx, sizeof(double) * 14 * N * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(du, su, sizeof(double) * 7 * N * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(dT, sT, sizeof(double) * B, hipMemcpyHostToDevice, st));
    if (int rc = mpcmp_sample_batch_device(ctx, B, dx, du, dT, n_pts, dout, st)) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out, dout, sizeof(double) * np, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return MPCMP_OK;
}


// examples/benchmark.cpp:58-160 for a batch of trajectories (either the initial guess or the MPC solution)
extern "C" int mpcmp_traj_stats_batch(mpcmp_ctx *ctx, int B, const double *sx, const double *su, const double *sT, const double *xf,
                                      int n_pts, double *out) {
    if (!ctx || B < 1 || n_pts < 1 || !sx || !su || !sT || !xf || !out) return MPCMP_EINVAL;
    SINGLE_ARM_ONLY(ctx);
    // the kernel keeps the n_pts+1 samples of one trajectory (28 doubles each) in LDS: 160 KB per workgroup on gfx950
    if (sizeof(double) * 28 * (size_t)(n_pts + 1) > 150 * 1024) { ctx->err = "mpcmp_traj_stats_batch: n_pts must be <= 684 (the samples of one trajectory are held in LDS)"; return MPCMP_EINVAL; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t N = ctx->N;
    TmpBuf tb(ctx, (21 * N * B + B + 14 * (size_t)B + 7 + 74 * (size_t)B) * sizeof(double));
    double *dx = tb.get<double>(14 * N * B), *du = tb.get<double>(7 * N * B), *dT = tb.get<double>(B), *df = tb.get<double>(14 * (size_t)B),
           *dj = tb.get<double>(7), *dout = tb.get<double>(74 * (size_t)B);
    if (!dx || !du || !dT || !df || !dj || !dout) { ctx->err = "hipMalloc failed"; return MPCMP_ERUNTIME; }
    double jerk[7], j10[7];
    mpcmp_default_limits(nullptr, nullptr, nullptr, nullptr, jerk, nullptr);
    for (int r = 0; r < 7; r++) j10[r] = 10.0 * jerk[r];                     // benchmark.cpp:128
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(dx, sx, sizeof(double) * 14 * N * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(du, su, sizeof(double) * 7 * N * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(dT, sT, sizeof(double) * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(df, xf, sizeof(double) * 14 * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(dj, j10, sizeof j10, hipMemcpyHostToDevice, st));
    const size_t lds = sizeof(double) * 28 * (size_t)(n_pts + 1);
    if (int rc = set_lds(ctx, k_traj_stats, lds)) return rc;
    hipLaunchKernelGGL(k_traj_stats, dim3(B), dim3(256), lds, st, ctx->d_model, ctx->nseg, n_pts, dx, du, dT, df, 1.7, 2.5, dj, dout);   // pandaWrapper.hpp:37-38
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, dout, sizeof(double) * 74 * B, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return MPCMP_OK;
}

// ------------------------------------------------------------------------------------------------
// receding-horizon driver (BASELINE config #5): B instances, every step = warm-started re-solve from the previous
// solution (re-guess rule of motionPlanner.cpp:199-207) + state advance along the new solution
// (get_MPC_point, motionPlanner.hpp:118-128).  The fixed launch sequence of one step is captured in a hipGraph.
extern "C" int mpcmp_reset_multipliers(mpcmp_ctx *ctx) {
    if (!ctx) return MPCMP_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemsetAsync(ctx->ws.lam, 0, sizeof(double) * ctx->lam_count, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->ws.status, 0, sizeof(*ctx->ws.status) * ctx->max_batch, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MPCMP_OK;
}

extern "C" int mpcmp_rh_init(mpcmp_ctx *ctx, int B, const double *x0, const double *xf) {
    if (!ctx || !x0 || !xf || B < 1) return MPCMP_EINVAL;
    if (B > ctx->max_batch) return MPCMP_ETOOBIG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemsetAsync(ctx->ws.lam, 0, sizeof(double) * ctx->lam_count, st));       // new instances: no multipliers to carry, no failed solve behind them
    HIPCHK(ctx, hipMemsetAsync(ctx->ws.status, 0, sizeof(*ctx->ws.status) * ctx->max_batch, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_retired, 0, sizeof(int) * ctx->max_batch, st));                       // nobody has arrived
    HIPCHK(ctx, hipMemsetAsync(ctx->d_rh_count, 0, 2 * sizeof(unsigned long long), st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_x0, x0, sizeof(double) * ctx->nx * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_xf, xf, sizeof(double) * ctx->nx * B, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    ctx->rh_B = B; ctx->rh_first = true;
    if (ctx->rh_exec) { (void)hipGraphExecDestroy(ctx->rh_exec); ctx->rh_exec = nullptr; }
    if (ctx->rh_graph) { (void)hipGraphDestroy(ctx->rh_graph); ctx->rh_graph = nullptr; }
    return MPCMP_OK;
}

static int rh_enqueue_step(mpcmp_ctx *ctx, double dt, bool first, hipStream_t st) {
    const int B = ctx->rh_B;
    // The driver's start guess, every step, from every instance's current state: the jerk-limited time-synchronised trajectory the reference takes from
    // Ruckig for solve_trajectory(true) (motionPlanner.cpp:146-175), with the velocity / acceleration limits of the configuration and the jerk
    // margin every example of the reference uses (0.1 x max jerk: examples/offline_trajectory.cpp:9, benchmark.cpp).  It starts the first solve and
    // RE-starts an instance whose previous solve is no guess (hard failure, T outside its box: k_init); the other re-solves re-guess from the
    // previous solution (solve_trajectory(false), motionPlanner.cpp:199-207).
    double jm[7];
    mpcmp_default_limits(nullptr, nullptr, nullptr, nullptr, jm, nullptr);
    for (int j = 0; j < 7; j++) jm[j] *= 0.1;
    // (after the first step only for the instances that restart: the generator costs ~ 0.15 ms per 512 instances, 4 % of a step)
    if (int rc = warm_start_jerk_device(ctx, B, ctx->d_x0, ctx->d_xf, nullptr, nullptr, nullptr, nullptr, jm, ctx->d_wx, ctx->d_wu, ctx->d_wT, st,
                                        first ? nullptr : ctx->ws.status, MPCMP_STATUS_NAN | MPCMP_STATUS_NOT_PD | MPCMP_STATUS_XCH_DEAD | MPCMP_STATUS_T_OUT_OF_BOX))
        return rc;
    const double *wx = first ? ctx->d_wx : ctx->d_sx, *wu = first ? ctx->d_wu : ctx->d_su, *wT = first ? ctx->d_wT : ctx->d_sT;
    ctx->rh_step = true;           // the driver's defaults of the start flags (run_config) and the retired instances (WS.retired)
    const int rc = solve_dispatch(ctx, B, ctx->d_x0, ctx->d_xf, wx, wu, wT, ctx->d_sx, ctx->d_su, ctx->d_sT, ctx->d_info, st, 0, first ? 0 : 1);
    ctx->rh_step = false;
    if (rc) return rc;
    hipLaunchKernelGGL(k_advance, dim3((B + 255) / 256), dim3(256), 0, st, ctx->nseg, ctx->nx, B, dt, ctx->cfg.eps_target, ctx->d_sx, ctx->d_sT,
                       ctx->ws.status, ctx->d_xf, ctx->d_x0, ctx->d_retired, ctx->d_info, ctx->d_rh_count);
    HIPCHK(ctx, hipGetLastError());
    return MPCMP_OK;
}

extern "C" int mpcmp_rh_run(mpcmp_ctx *ctx, int steps, double dt, int use_graph) {
    if (!ctx || steps < 1 || !(dt > 0.0) || ctx->rh_B < 1) return MPCMP_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int done = 0;
    if (ctx->rh_first) {                       // first solve: built-in initialiser, not part of the graph
        if (int rc = rh_enqueue_step(ctx, dt, true, st)) return rc;
        ctx->rh_first = false; done = 1;
    }
    if (use_graph && steps - done > 0) {
        if (!ctx->rh_exec || ctx->rh_dt != dt) {
            if (ctx->rh_exec) { (void)hipGraphExecDestroy(ctx->rh_exec); ctx->rh_exec = nullptr; }
            if (ctx->rh_graph) { (void)hipGraphDestroy(ctx->rh_graph); ctx->rh_graph = nullptr; }
            ctx->capturing = true;
            HIPCHK(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int rc = rh_enqueue_step(ctx, dt, false, st);
            hipError_t e = hipStreamEndCapture(st, &ctx->rh_graph);
            ctx->capturing = false;
            if (rc) return rc;
            HIPCHK(ctx, e);
            HIPCHK(ctx, hipGraphInstantiate(&ctx->rh_exec, ctx->rh_graph, nullptr, nullptr, 0));
            ctx->rh_dt = dt;
        }
        for (; done < steps; done++) HIPCHK(ctx, hipGraphLaunch(ctx->rh_exec, st));
    } else {
        for (; done < steps; done++)
            if (int rc = rh_enqueue_step(ctx, dt, false, st)) return rc;
    }
    HIPCHK(ctx, hipStreamSynchronize(st));
    return MPCMP_OK;
}

extern "C" int mpcmp_rh_get(mpcmp_ctx *ctx, double *x0_now, double *sx, double *su, double *sT, mpcmp_info *info) {
    if (!ctx || ctx->rh_B < 1) return MPCMP_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t B = ctx->rh_B, N = ctx->N;
    hipStream_t st = ctx->stream;
    if (x0_now) HIPCHK(ctx, hipMemcpyAsync(x0_now, ctx->d_x0, sizeof(double) * ctx->nx * B, hipMemcpyDeviceToHost, st));
    if (sx) HIPCHK(ctx, hipMemcpyAsync(sx, ctx->d_sx, sizeof(double) * ctx->nx * N * B, hipMemcpyDeviceToHost, st));
    if (su) HIPCHK(ctx, hipMemcpyAsync(su, ctx->d_su, sizeof(double) * ctx->nu * N * B, hipMemcpyDeviceToHost, st));
    if (sT) HIPCHK(ctx, hipMemcpyAsync(sT, ctx->d_sT, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    if (info) HIPCHK(ctx, hipMemcpyAsync(info, ctx->d_info, sizeof(mpcmp_info) * B, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return MPCMP_OK;
}

extern "C" int mpcmp_rh_stats(mpcmp_ctx *ctx, long long *resolves_done, long long *arrived) {
    if (!ctx || ctx->rh_B < 1) return MPCMP_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    unsigned long long c[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(c, ctx->d_rh_count, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (resolves_done) *resolves_done = (long long)c[0];
    if (arrived) *arrived = (long long)c[1];
    return MPCMP_OK;
}

// diagnostics: copy one workspace array of the last solve to the host (which: 0 z, 1 lambda, 2 c_eq, 3 g, 4 p, 5 y)
extern "C" int mpcmp_debug_fetch(mpcmp_ctx *ctx, int which, double *out, long count) {
    if (!ctx || !out || count < 1) return MPCMP_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipDeviceSynchronize());
    const double *src = which == 0 ? ctx->ws.z : which == 1 ? ctx->ws.lam : which == 2 ? ctx->ws.ceq : which == 3 ? ctx->ws.g
                      : which == 4 ? ctx->ws.p : which == 5 ? ctx->ws.y : nullptr;
    if (!src) return MPCMP_EINVAL;
    HIPCHK(ctx, hipMemcpy(out, src, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost));
    return MPCMP_OK;
}

// diagnostics: raw phase stamps of the last k_qp launch (all zero unless built with -DMPCMP_STAMPS)
extern "C" int mpcmp_debug_stamps(mpcmp_ctx *ctx, int B, unsigned long long *out) {
    if (!ctx || !out || B < 1 || B > ctx->max_batch) return MPCMP_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipMemcpy(out, ctx->ws.dbg, sizeof(unsigned long long) * MPCMP_DBG_WORDS * B, hipMemcpyDeviceToHost));
    return MPCMP_OK;
}

extern "C" int mpcmp_kernel_timing(mpcmp_ctx *ctx, int reset, const char **name, double *ms_total, int *launches) {
    if (!ctx) return MPCMP_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // fold finished event pairs into the accumulators (caller has synchronised the stream)
    for (size_t i = 0; i < ctx->ev_used; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev[i].e[0], ctx->ev[i].e[1]) == hipSuccess) { ctx->qp_ms += ms; ctx->qp_launches++; }
    }
    ctx->ev_used = 0;
    ctx->timing = true;            // event recording starts with the first call (bench.py calls it once before the timed region)
    if (name) *name = (ctx->nseg == 6 && ctx->narm == 1 && ctx->qp19 == 5) ? "k_qp5" : ctx->nseg >= 6 ? "k_qp3" : (ctx->nseg == 4 && ctx->qp13 == 3 ? "k_qp3" : (ctx->nseg == 4 && ctx->qp13 == 4 ? "k_qp4" : ((ctx->nseg == 2 || ctx->nseg == 4) ? "k_qp2" : "k_qp")));
    if (ms_total) *ms_total = ctx->qp_ms;
    if (launches) *launches = ctx->qp_launches;
    if (reset) { ctx->qp_ms = 0.0; ctx->qp_launches = 0; }
    return MPCMP_OK;
}
