/*
 * mpcmp.h — C ABI of the MI355X-native batched minimum-time joint-space MPC solver.
 *
 * Drop-in boundary for the hot path of AlbericDeLajarte/mpc_motion_planner. The reference has no FFI
 * layer: its boundary is the C++ class `MotionPlanner` (mpc_solver/motionPlanner.hpp:16-176) whose
 * solve_trajectory() (mpc_solver/motionPlanner.cpp:177-208) calls polympc's mpc.solve() once per
 * (start,target) pair on one CPU thread (examples/benchmark.cpp:16 loops it 1000 times).  This ABI is what
 * a batched `MotionPlanner` binds instead; include/mpcmp_motion_planner.hpp is that binding.
 *
 * Conventions: every entry point returns 0 on success or a negative MPCMP_E* code and never throws; plain
 * pointers and sizes only; doubles everywhere (the reference is `double`, robot_ocp.hpp:38).  `*_device`
 * entry points take DEVICE pointers and a hipStream_t (as void*) and are asynchronous on that stream;
 * the others take HOST pointers and are synchronous.
 *
 * Layouts (node-major, ascending time — motionPlanner.cpp:158-174,202-203):
 *   x0, xf   [B][14]        = [q(7); qd(7)]            current_state / target_state (motionPlanner.hpp:40-41)
 *   sol_x    [B][N][14]     state at collocation node k  (mpc.solution_x(),  traj_state_t)
 *   sol_u    [B][N][7]      control (= qdd) at node k    (mpc.solution_u(),  traj_control_t)
 *   sol_T    [B]            final time                   (mpc.solution_p()[0])
 *   N = 3*num_seg + 1       (POLY_ORDER 3, robot_ocp.hpp:31-36)
 */
#ifndef MPCMP_H
#define MPCMP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MPCMP_NDOF 7
#define MPCMP_NX 14
#define MPCMP_NU 7
#define MPCMP_NG 8

#define MPCMP_OK 0
#define MPCMP_EINVAL (-1)      /* bad argument / unsupported configuration */
#define MPCMP_ENODEVICE (-2)   /* no HIP device, or the HIP runtime failed: the product has NO CPU fallback */
#define MPCMP_ERUNTIME (-3)    /* a HIP call failed; see mpcmp_last_error */
#define MPCMP_ETOOBIG (-4)     /* batch larger than the context's capacity */

typedef struct mpcmp_ctx mpcmp_ctx;

/* Serial chain of 7 revolute-z joints + tool frame; replaces pinocchio::Model built from the URDF
 * (robot_ocp.hpp:50-53, robot_utils/pandaWrapper.cpp:3-12). Same memory layout as the oracle's model. */
typedef struct {
    double R0[7][9];   /* fixed rotation of the joint placement (row-major), parent <- joint at q=0 */
    double p[7][3];    /* joint origin in the parent joint frame                                     */
    double mass[7];
    double com[7][3];  /* COM of the lumped body in its joint frame                                  */
    double I[7][9];    /* rotational inertia about the COM                                           */
    double tool[3];    /* frame "panda_tool" in the joint-7 frame (height constraint, robot_ocp.hpp:52,91) */
    double link8[3];   /* frame "panda_link8" in the joint-7 frame                                    */
    double gravity[3];
} mpcmp_model;

/* Solver settings + bounds; replaces mpc.settings()/qp_settings()/..._bounds() (motionPlanner.cpp:15-20,56-100). */
typedef struct {
    int    num_seg;          /* NUM_SEG, robot_ocp.hpp:32 (4 -> 13 nodes, 6 -> 19 nodes as shipped, 8 -> 25)*/
    int    sqp_iters;        /* mpc.settings().max_iter,             motionPlanner.cpp:15 (2 as shipped)   */
    int    qp_iters;         /* mpc.qp_settings().max_iter,          motionPlanner.cpp:16 (700)            */
    int    ls_iters;         /* mpc.settings().line_search_max_iter, motionPlanner.cpp:17 (10)             */
    int    check_every;      /* ADMM termination-test interval (25)                                         */
    int    quirk_dtau_dT;    /* keep the non-physical d tau/dT column of robot_ocp.hpp:124,138 (1)          */
    double eps_abs, eps_rel; /* mpc.qp_settings().eps_*,             motionPlanner.cpp:19-20 (1e-3)        */
    double rho, sigma, alpha, rho_eq_scale;   /* box-ADMM parameters (0.02, 1e-6, 1.4, 1e3: fitted) */
    double ls_eta, ls_tau;   /* Armijo fraction and backtracking factor (0.25, 0.5)                         */
    double hess_reg;         /* Gershgorin shift, polympc_redef.hpp:68 (1e-3)                               */
    double eps_target;       /* terminal box half-width, motionPlanner.hpp:44 (1e-2)                        */
    double lbx[14], ubx[14]; /* mpc.state_bounds,       motionPlanner.cpp:66-70 */
    double lbu[7], ubu[7];   /* mpc.control_bounds,     motionPlanner.cpp:73    */
    double lbg[8], ubg[8];   /* mpc.constraints_bounds, motionPlanner.cpp:92-98 */
    double lbT, ubT;         /* mpc.parameters_bounds,  motionPlanner.cpp:76-79 */
    /* QP start of the SQP iterations after the first (the build's choice: polympc is absent).  0 (default) = every QP starts cold,
     * x = z = y = 0; 1 = warm duals: y_0 = the NLP multipliers lambda_k, x_0 = 0, z_0 = clip(0, l, u).  On the headline batch the warm start
     * nearly halves the ADMM iterations per trajectory with better feasibility; the stored reference solve is ONE SQP iteration
     * and cannot tell the two apart. */
    int    qp_warm_start;    /* also -1: off everywhere, including the receding-horizon driver (whose default is ON, see mpcmp_rh_run) */
    /* Multipliers at the start of a solve.  0 (default) = lambda_0 = 0 for every problem of every call (independent problems: the batch semantics).
     * 1 = the multipliers a problem slot was left with by the context's previous solve are the start of the next one (zero in a fresh context and
     * after mpcmp_reset_multipliers / mpcmp_rh_init): what a re-solve on ONE MotionPlanner object most plausibly does upstream (polympc keeps its
     * dual iterate until lam_guess is called: SURVEY.md 3.2, unverified) and what a receding-horizon loop wants.  Slot b of a call = problem b. */
    int    carry_multipliers; /* also -1: off everywhere, including the receding-horizon driver (whose default is ON, see mpcmp_rh_run) */
} mpcmp_config;

/* zero the carried multipliers of every problem slot (mpcmp_config.carry_multipliers) */
int mpcmp_reset_multipliers(mpcmp_ctx *ctx);

/* Per-problem result record; replaces mpc.info() (never read by the reference, motionPlanner.cpp:191). */
typedef struct {
    double T;             /* final time                                                   */
    double viol_l1;       /* l1 constraint violation of the returned iterate              */
    double defect_inf;    /* inf-norm of the collocation defects                          */
    double path_viol_inf; /* inf-norm violation of torque / height bounds at the nodes    */
    double term_err_inf;  /* || x_N - x_target ||_inf                                     */
    double last_alpha;    /* step length of the last SQP iteration                        */
    int    qp_iters_total;/* ADMM iterations executed over all SQP iterations             */
    int    sqp_iters;
    int    status;        /* 0 = converged QPs and an iterate inside every tolerance; else MPCMP_STATUS_* bits              */
    int    qp_capped;     /* number of SQP iterations whose QP ran out of qp_iters before its termination test was met    */
} mpcmp_info;
/* status bits.  The reference never reads mpc.info().status (motionPlanner.cpp:191); this record is the only failure channel a
 * batched caller has ("Failure detection"). */
#define MPCMP_STATUS_NAN          1   /* NaN / Inf in the returned iterate                                                   */
#define MPCMP_STATUS_NOT_PD       2   /* a KKT factorisation lost positive definiteness                                      */
#define MPCMP_STATUS_XCH_DEAD     4   /* multi-arm OCP: the partner workgroup never answered an exchange                     */
#define MPCMP_STATUS_QP_CAPPED    8   /* at least one QP stopped at qp_iters (qp_capped counts them): a truncated-ADMM step  */
#define MPCMP_STATUS_OUTSIDE_TOL 16   /* returned iterate outside tolerance: collocation defect or path violation > eps_abs,
                                         or terminal error > eps_target + eps_abs                                           */
#define MPCMP_STATUS_T_OUT_OF_BOX 32  /* final time outside [lbT, ubT]                                                       */
#define MPCMP_STATUS_ARRIVED      64  /* receding-horizon driver only: the instance has arrived and is no longer re-solved; the
                                         other bits and fields of the record are those of its last solve (mpcmp_rh_run)       */

/* ---- configuration helpers (host, no GPU needed) ---- */
int mpcmp_default_model(mpcmp_model *m);                                    /* Panda arm, panda_arm.urdf */
int mpcmp_model_from_urdf(const char *urdf_path, mpcmp_model *m);           /* MotionPlanner(std::string urdf_path), motionPlanner.cpp:3 */
/* General form (robot_utils/pandaWrapper.cpp:3-12 hands any URDF to Pinocchio): every serial chain of seven revolute joints that
 * hangs on the root link, in file order.  Fixed joints anywhere (rotated or not) are folded into the next joint placement and
 * their links lumped into the body of the joint frame they hang on; a chain's base placement is folded into its first joint;
 * rotated inertial frames and joint axes other than +z are handled.  *n_chains <= max_chains models are written; the result
 * feeds mpcmp_create_multi.  MPCMP_EINVAL (see mpcmp_last_error(NULL)): unsupported joint type, branching chain, != 7 joints. */
int mpcmp_models_from_urdf(const char *urdf_path, int max_chains, mpcmp_model *models, int *n_chains);
int mpcmp_default_limits(double *qmin, double *qmax, double *vmax, double *amax, double *jmax,
                         double *taumax);                                    /* pandaWrapper.hpp:29-34 */
int mpcmp_default_config(mpcmp_config *c, int num_seg, int sqp_iters);      /* motionPlanner.cpp:15-24 */
int mpcmp_set_margins(mpcmp_config *c, double margin_position, double margin_velocity,
                      double margin_acceleration, double margin_torque);    /* set_constraint_margins, motionPlanner.cpp:56-90 */
int mpcmp_set_min_height(mpcmp_config *c, double min_height);               /* set_min_height, motionPlanner.cpp:92-100 */
int mpcmp_num_nodes(int num_seg);
int mpcmp_time_nodes(int num_seg, double *tau);                             /* mpc.ocp().time_nodes, ascending */
const char *mpcmp_version(void);

/* ---- context ---- */
/* model == NULL -> compiled-in Panda. Allocates all device workspaces for up to max_batch problems;
 * no allocation happens in the solve calls. Fails with MPCMP_ENODEVICE when no GPU is present. */
int mpcmp_create(const mpcmp_config *cfg, const mpcmp_model *model, int device, int max_batch, mpcmp_ctx **out);
/* Multi-arm robot (BASELINE.json configs[3]: 14-DoF dual Panda): `models` points at narm (1 or 2) models — independent 7-joint
 * chains on one base; a chain's base placement is folded into its first joint placement (mpcmp_models_from_urdf).  The OCP is
 * the reference's (robot_ocp.hpp:31-213) with doubled sizes NX = 14 narm, NU = 7 narm, NG = 8 narm; the arms couple only
 * through the final time.  Layouts then are x0, xf [B][14 narm] = [q(7 narm); qd(7 narm)], sol_x [B][N][14 narm],
 * sol_u [B][N][7 narm]; the per-arm limit tables of mpcmp_config apply to every arm.  narm = 2 needs num_seg 6 or 8 and a warm
 * start (mpcmp_warm_start_jerk_batch[_device]); the leaf / resampling entry points are single-arm only. */
int mpcmp_create_multi(const mpcmp_config *cfg, const mpcmp_model *models, int narm, int device, int max_batch, mpcmp_ctx **out);
int mpcmp_destroy(mpcmp_ctx *ctx);
int mpcmp_set_config(mpcmp_ctx *ctx, const mpcmp_config *cfg);             /* num_seg must not change */
const char *mpcmp_last_error(const mpcmp_ctx *ctx);

/* ---- the hot path: B independent OCPs == B calls of MotionPlanner::solve_trajectory ---- */
/* warm_x/warm_u/warm_T: x_guess/u_guess/p_guess (motionPlanner.cpp:172-174); all NULL -> built-in
 * initialiser standing in for warm_start_RK (motionPlanner.cpp:146-175). info may be NULL. */
int mpcmp_solve_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf,
                      const double *warm_x, const double *warm_u, const double *warm_T,
                      double *sol_x, double *sol_u, double *sol_T, mpcmp_info *info);
int mpcmp_solve_batch_device(mpcmp_ctx *ctx, int B, const double *d_x0, const double *d_xf,
                             const double *d_warm_x, const double *d_warm_u, const double *d_warm_T,
                             double *d_sol_x, double *d_sol_u, double *d_sol_T, mpcmp_info *d_info,
                             void *hip_stream);
/* built-in initialiser alone (what solve_batch uses when warm_* are NULL) */
int mpcmp_warm_start_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf,
                           double *warm_x, double *warm_u, double *warm_T);

/* ---- leaf kernels exposed for parity tests and for callers that only need the rigid-body layer ---- */
/* pinocchio::rnea (robot_ocp.hpp:91; motionPlanner.hpp:92,111,127,141): n x [q7],[qd7],[qdd7] -> n x [tau7] */
int mpcmp_rnea_batch(mpcmp_ctx *ctx, int n, const double *q, const double *qd, const double *qdd, double *tau);
/* minTime_ocp::evalConstraints AD overload (robot_ocp.hpp:98-163): n x [x14],[u7] -> g [n][8], G [n][8][22] */
int mpcmp_eval_constraints_batch(mpcmp_ctx *ctx, int n, const double *x, const double *u, double *g, double *G);
/* one QP of the SQP (linearise at (x,u,T,lam=0) + box-ADMM): p [B][21N+1] (layout xs|us|T), y [B][m+n],
 * iters [B]; m = 14(N-1) + 8N rows ordered [dynamics | path], then the n box rows. */
int mpcmp_qp_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *xs,
                   const double *us, const double *T, double *p, double *y, int *iters);

/* ---- resampling: MotionPlanner::get_MPC_trajectory<n_pts> (motionPlanner.hpp:99-116) ---- */
/* out [B][n_pts+1][29] = time, q(7), qd(7), qdd(7), tau(7) — the row format of analysis/optimal_solution.txt
 * (examples/offline_trajectory.cpp:88-105) */
int mpcmp_sample_batch(mpcmp_ctx *ctx, int B, const double *sol_x, const double *sol_u, const double *sol_T,
                       int n_pts, double *out);
int mpcmp_sample_batch_device(mpcmp_ctx *ctx, int B, const double *d_sol_x, const double *d_sol_u,
                              const double *d_sol_T, int n_pts, double *d_out, void *hip_stream);

/* ---- jerk-limited, time-synchronised warm start (stands in for Ruckig, mpc_solver/motionPlanner.cpp:146-175) ----
 * Per joint: S-curve velocity transition, cruise, S-curve transition; minimum time of the slowest joint, every other joint
 * re-planned to exactly that duration.  Boundary accelerations: zero in the plain entry points (every example of the reference),
 * given in the *_acc_* ones (set_current_state / set_target_state forward them to Ruckig, motionPlanner.cpp:36-38,50-52).
 * Velocity / acceleration limits are the context's margin-applied bounds (mpcmp_set_margins), jmax[7] = margin_jerk *
 * max_jerk (motionPlanner.cpp:86-88).  Outputs in the layout mpcmp_solve_batch takes as warm start: warm_x [B][N][14],
 * warm_u [B][N][7], warm_T [B] (x_guess / u_guess / p_guess, motionPlanner.cpp:158-174). */
int mpcmp_warm_start_jerk_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *jmax,
                                double *warm_x, double *warm_u, double *warm_T);
int mpcmp_warm_start_jerk_batch_device(mpcmp_ctx *ctx, int B, const double *d_x0, const double *d_xf, const double *jmax,
                                       double *d_warm_x, double *d_warm_u, double *d_warm_T, void *hip_stream);
/* the same trajectory sampled uniformly, out [B][n_pts+1][22] = t, q(7), qd(7), qdd(7); T_out [B] (may be NULL):
 * MotionPlanner::get_ruckig_trajectory<n_pts> (motionPlanner.hpp:73-96) without the torque column block */
int mpcmp_jerk_trajectory_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *jmax, int n_pts,
                                double *out, double *T_out);

/* MotionPlanner::get_RK_point (motionPlanner.hpp:130-142): the same trajectory at ONE physical time per problem, clamped to
 * its duration, with the RNEA torque: out [B][28] = q(7), qd(7), qdd(7), tau(7); T_out [B] (may be NULL) = the durations. */
int mpcmp_jerk_point_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *jmax, const double *time,
                           double *out, double *T_out);
/* The three entry points above with boundary accelerations acc0 / accT [B][7] (either may be NULL = zero; with both NULL the result is
 * the plain entry point's bit for bit).  Single-arm contexts.  The acceleration limit yields to a given boundary acceleration above it.  A boundary
 * acceleration that drives a velocity from inside its limit to outside it however hard the jerk brakes (|v0| <= vmax < |v0 + a0 |a0| / 2J| at the start,
 * |vT| <= vmax < |vT - aT |aT| / 2J| at the target) is rejected with MPCMP_EINVAL by the host-pointer entry points, as Ruckig rejects such an input; the *_device variant does not read its inputs on the
 * host and does not check. */
int mpcmp_warm_start_jerk_acc_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *acc0, const double *accT,
                                    const double *jmax, double *warm_x, double *warm_u, double *warm_T);
int mpcmp_warm_start_jerk_acc_batch_device(mpcmp_ctx *ctx, int B, const double *d_x0, const double *d_xf, const double *d_acc0, const double *d_accT,
                                           const double *jmax, double *d_warm_x, double *d_warm_u, double *d_warm_T, void *hip_stream);
int mpcmp_jerk_trajectory_acc_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *acc0, const double *accT,
                                    const double *jmax, int n_pts, double *out, double *T_out);
int mpcmp_jerk_point_acc_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *acc0, const double *accT,
                               const double *jmax, const double *time, double *out, double *T_out);
/* The same three with the generator's velocity / acceleration limits given by the caller, vmax / amax [7] (either may be NULL = the context's
 * margin-applied bounds): what ruckig::InputParameter::max_velocity / max_acceleration are to otg.calculate (motionPlanner.cpp:149).  The
 * context's configuration is neither read for these limits nor changed. */
int mpcmp_warm_start_jerk_lim_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *acc0, const double *accT,
                                    const double *vmax, const double *amax, const double *jmax, double *warm_x, double *warm_u, double *warm_T);
int mpcmp_jerk_trajectory_lim_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *acc0, const double *accT,
                                    const double *vmax, const double *amax, const double *jmax, int n_pts, double *out, double *T_out);
int mpcmp_jerk_point_lim_batch(mpcmp_ctx *ctx, int B, const double *x0, const double *xf, const double *acc0, const double *accT,
                               const double *vmax, const double *amax, const double *jmax, const double *time, double *out, double *T_out);
/* MotionPlanner::get_MPC_point (motionPlanner.hpp:118-128): the solution at ONE physical time per problem, including the
 * reference's clamp (time >= T: the normalised time becomes T, not 1), with the RNEA torque: out [B][28] as above. */
int mpcmp_mpc_point_batch(mpcmp_ctx *ctx, int B, const double *sol_x, const double *sol_u, const double *sol_T, const double *time,
                          double *out);

/* ---- scenario helpers of the robot wrapper (host side, as in the reference; never called by the batched solve) ---- */
/* World-aligned 6x7 Jacobian of the tool frame, rows [linear(3); angular(3)], row-major (J = blockdiag(R,R) * J_local,
 * robot_utils/pandaWrapper.cpp:70-75,97-101); optional tool position p[3] and rotation R[9] (row-major). */
int mpcmp_tool_jacobian(const mpcmp_model *model, const double *q, double *J, double *p, double *R);
/* PandaWrapper::forward_velocities (pandaWrapper.cpp:90-107): out[6] = J(q) qd = [linear; angular] task velocity. */
int mpcmp_forward_velocities(const mpcmp_model *model, const double *q, const double *qd, double *out);
/* PandaWrapper::inverse_velocities (pandaWrapper.cpp:62-88): qd = J^T (J J^T + 1e-5 I)^-1 [lin; ang]. */
int mpcmp_inverse_velocities(const mpcmp_model *model, const double *q, const double *lin, const double *ang, double *qd);
/* PandaWrapper::inverse_kinematic (pandaWrapper.cpp:14-60): damped least squares on the tool frame, error = log6 of the
 * pose error in the desired frame, step -J_local^T (J J^T + 1e-2 I)^-1 err * 0.1, stop at |err| < 1e-4 or 1000 iterations.
 * The reference starts from pinocchio::randomConfiguration; here the start q_init[7] is an argument (NULL = zeros).
 * R is row-major.  Returns MPCMP_OK when converged, 1 when the iteration cap was hit (q still holds the last iterate). */
int mpcmp_inverse_kinematics(const mpcmp_model *model, const double *R, const double *p, const double *q_init, double *q,
                             int *iters);

/* ---- trajectory checks of the reference benchmark (examples/benchmark.cpp:58-160) ---- */
/* out [B][74] = min(28) | max(28) of q,qd,qdd,tau over n_pts+1 uniform samples | x(T) - target (14) |
 * flags (4; 1 = pass): jerk (|d qdd/dt| <= 10 max_jerk), linear task velocity <= 1.7, angular <= 2.5, tool z >= 0.
 * One 162-number row of analysis/benchmark_data.txt = [guess min, guess max, mpc min, mpc max, guess err, mpc err,
 * guess flags, mpc flags, target] (benchmark.cpp:164-194). */
int mpcmp_traj_stats_batch(mpcmp_ctx *ctx, int B, const double *sol_x, const double *sol_u, const double *sol_T,
                           const double *xf, int n_pts, double *out);

/* ---- receding-horizon driver (BASELINE.json config #5) ---- */
/* B instances; every step = re-solve warm-started from the previous solution with the reference's re-guess rule
 * (head := current state, tail := target; motionPlanner.cpp:199-207), then the current state advances along the
 * new solution by dt (MotionPlanner::get_MPC_point, motionPlanner.hpp:118-128).  The first step starts from the
 * jerk-limited trajectory (below).  use_graph != 0 replays one hipGraph-captured step (fixed iteration counts: no host round trip). */
int mpcmp_rh_init(mpcmp_ctx *ctx, int B, const double *x0, const double *xf);
/* Arrival.  The reference's re-solve loop has no end (the caller of solve_trajectory(false) decides, motionPlanner.cpp:199-207), and an OCP whose
 * start already lies in its terminal box degenerates (T -> lbT = 0).  After every solve the driver therefore RETIRES an instance
 *   - whose plan ends within the control period (T <= dt): it follows the plan to its last node first, or
 *   - whose advanced state lies inside the terminal box, |x - x_target|_inf <= eps_target (motionPlanner.hpp:44).
 * A retired instance keeps its state, last solution and record (status |= MPCMP_STATUS_ARRIVED) and is not re-solved: its workgroups return
 * at once.  An instance whose solve failed hard (NaN, lost positive definiteness, dead exchange) or left the box of T holds its state and
 * re-solves from the driver's start guess with zero multipliers.
 * Start guess of the driver (first solve of every instance, and such restarts): the jerk-limited time-synchronised trajectory from the current
 * state to the target — what the reference takes from Ruckig for solve_trajectory(true), motionPlanner.cpp:146-175 — with the velocity and
 * acceleration limits of the configuration and the jerk margin of the reference's examples (0.1 x max jerk, examples/offline_trajectory.cpp:9);
 * every other re-solve re-guesses from the previous solution (solve_trajectory(false), motionPlanner.cpp:199-207).
 * Defaults of the driver: the re-solves run with carry_multipliers = 1 and qp_warm_start = 1 (a re-solve continues from the multipliers and
 * duals of the one before) unless the configuration sets a flag to -1.  mpcmp_solve_batch is unaffected (0 = off there). */
int mpcmp_rh_run(mpcmp_ctx *ctx, int steps, double dt, int use_graph);
int mpcmp_rh_get(mpcmp_ctx *ctx, double *x0_now, double *sol_x, double *sol_u, double *sol_T, mpcmp_info *info);
/* counters since mpcmp_rh_init: re-solves actually executed (instance-steps of live instances) and instances retired so far; either may be NULL */
int mpcmp_rh_stats(mpcmp_ctx *ctx, long long *resolves_done, long long *arrived);

/* ---- measurement hooks used by bench.py ---- */
/* name and accumulated device time (ms, HIP events on the solve stream) of the dominant kernel since the
 * last reset; launches = number of launches accumulated.  Event recording is OFF until the first call (a plain solve
 * records nothing and allocates nothing); afterwards at most 4096 launches are held between two calls. */
int mpcmp_kernel_timing(mpcmp_ctx *ctx, int reset, const char **name, double *ms_total, int *launches);

/* diagnostics: the first `count` doubles of one workspace array of the last solve (which: 0 iterate z, 1 multipliers, 2 collocation
 * defects c_eq, 3 path constraint values g, 4 QP step p, 5 QP duals y), in the device's internal layout */
int mpcmp_debug_fetch(mpcmp_ctx *ctx, int which, double *out, long count);

/* diagnostics: per-problem phase cycle stamps of the last k_qp launch, [B][160] (16 workgroup stamps, then
 * [16 waves][8] busy cycles per ADMM phase, then 16 stamps of the step kernel); zeros unless the library was built with -DMPCMP_STAMPS (tools/stamps.py). */
int mpcmp_debug_stamps(mpcmp_ctx *ctx, int B, unsigned long long *out);

#ifdef __cplusplus
}
#endif
#endif
