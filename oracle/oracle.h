/*
 * oracle.h — CPU restatement (TEST INFRASTRUCTURE ONLY) of the hot path of
 * AlbericDeLajarte/mpc_motion_planner: minimum-time joint-space OCP for the 7-DoF Panda.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or load
 * this library. The product (mpc_motion_planner_amd/) never does.
 *
 * Parity status
 *   - rigid-body layer (RNEA / FK / Jacobian):  PINNED by the reference's stored Pinocchio
 *     outputs (tests/golden/kat_rnea.csv, kat_fk_link8.csv, kat_jac.json).
 *   - RNEA derivatives / mass matrix: pinned indirectly (finite differences of the pinned RNEA).
 *   - jerk-limited warm start (jerk.c, restating the Ruckig call): PINNED by the reference's stored Ruckig trajectory
 *     (tests/golden/gold_traj.json: duration to 6 digits, all seven joints to the stored precision).
 *   - collocation / SQP / box-ADMM layer: "PARITY UNPINNED" at digit level. The arithmetic of
 *     that layer lives in polympc (https://gitlab.epfl.ch/listov/polympc.git, branch
 *     collocation_fix_jw, commit not recoverable), which is an empty submodule in the
 *     reference snapshot. It is restated from the reference's call sites
 *     (mpc_solver/robot_ocp.hpp, polympc_redef.hpp, motionPlanner.cpp) and the published
 *     OSQP/SQP algorithms; the one stored solve (tests/golden/gold_traj.json) pins it at
 *     regime level only (T between the converged optimum and the Ruckig warm start).
 *
 * Reference files followed: mpc_solver/robot_ocp.hpp (whole), mpc_solver/polympc_redef.hpp
 * (whole), mpc_solver/motionPlanner.cpp:15-20,27-100,146-208, mpc_solver/motionPlanner.hpp:99-172,
 * robot_utils/pandaWrapper.hpp:28-40, robot_utils/panda-model/panda_arm.urdf.
 */
#ifndef MPCMP_ORACLE_H
#define MPCMP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NJ 7          /* joints of the serial chain                              */
#define ORC_NX 14         /* state  x=[q;qd]         robot_ocp.hpp:38                 */
#define ORC_NU 7          /* control u=qdd           robot_ocp.hpp:38                 */
#define ORC_NG 8          /* path constraints [tau(7); z_tool]  robot_ocp.hpp:38,91   */
#define ORC_NV 21         /* NX+NU per node                                           */
#define ORC_MAXSEG 8

typedef struct {
    double R0[ORC_NJ][9];  /* fixed part of joint placement rotation (row-major), parent <- joint */
    double p[ORC_NJ][3];   /* joint origin in parent joint frame                                  */
    double mass[ORC_NJ];
    double com[ORC_NJ][3]; /* centre of mass of the (lumped) body in its joint frame              */
    double I[ORC_NJ][9];   /* rotational inertia about the COM, joint-frame axes                  */
    double tool[3];        /* frame panda_tool in joint-7 frame   (panda_arm.urdf:134-153)        */
    double link8[3];       /* frame panda_link8 in joint-7 frame  (panda_arm.urdf:134-139)        */
    double gravity[3];     /* (0,0,-9.81)                                                         */
} orc_model;

/* solver configuration. Everything the reference leaves to polympc defaults is explicit here. */
typedef struct {
    int    num_seg;         /* NUM_SEG (robot_ocp.hpp:32); POLY_ORDER is fixed at 3 (robot_ocp.hpp:31) */
    int    sqp_iters;       /* mpc.settings().max_iter              motionPlanner.cpp:15 */
    int    qp_iters;        /* mpc.qp_settings().max_iter           motionPlanner.cpp:16 */
    int    ls_iters;        /* line_search_max_iter                 motionPlanner.cpp:17 */
    int    check_every;     /* ADMM termination test interval (25)                       */
    int    quirk_dtau_dT;   /* keep robot_ocp.hpp:124,138 column 21                      */
    double eps_abs, eps_rel;/* motionPlanner.cpp:19-20                                   */
    double rho, sigma, alpha, rho_eq_scale; /* ADMM parameters (build's choice) */
    double ls_eta, ls_tau;  /* Armijo slope fraction / shrink factor                     */
    double hess_reg;        /* +0.001 of polympc_redef.hpp:68                            */
    double eps_target;      /* terminal box half width 1e-2, motionPlanner.hpp:44        */
    double lbx[ORC_NX], ubx[ORC_NX]; /* state box      motionPlanner.cpp:66-70 */
    double lbu[ORC_NU], ubu[ORC_NU]; /* control box    motionPlanner.cpp:73    */
    double lbg[ORC_NG], ubg[ORC_NG]; /* path bounds    motionPlanner.cpp:92-98 */
    double lbT, ubT;                 /* motionPlanner.cpp:76-79                */
    int    qp_warm_start;            /* 0: every QP starts cold; 1: y_0 = lambda_k, x_0 = 0, z_0 = clip(0, l, u) (include/mpcmp.h) */
    int    carry_multipliers;   /* 1: orc_solve_carry starts from the multipliers handed in (mpcmp_config.carry_multipliers) */
} orc_config;

typedef struct {
    double T;            /* final time                                             */
    double viol_l1;      /* l1 constraint violation of the returned iterate        */
    double defect_inf;   /* inf-norm of collocation defects                        */
    double path_viol_inf;/* inf-norm violation of torque/height bounds at nodes    */
    double term_err_inf; /* || x_N - x_target ||_inf                               */
    double last_alpha;   /* step length taken in the last SQP iteration            */
    int    qp_iters_total;
    int    sqp_iters;
    int    status;       /* bits: 1 NaN, 2 factorisation lost positive definiteness, 8 a QP hit qp_iters, 16 returned iterate outside
                            tolerance (defect / path violation > eps_abs, terminal error > eps_target + eps_abs), 32 T outside its box */
    int    qp_capped;    /* number of SQP iterations whose QP hit qp_iters          */
} orc_info;

/* ---- model ---- */
void orc_default_model(orc_model *m);                 /* compiled-in Panda (panda_arm.urdf) */
void orc_default_limits(double *qmin, double *qmax, double *vmax, double *amax, double *jmax,
                        double *taumax);              /* pandaWrapper.hpp:29-34 */
void orc_default_config(orc_config *c, int num_seg, int sqp_iters);
void orc_set_margins(orc_config *c, double mp, double mv, double ma, double mt); /* motionPlanner.cpp:56-100 */

/* ---- rigid body ---- */
void orc_rnea(const orc_model *m, const double *q, const double *v, const double *a, double *tau);
/* JVP-based analytic derivatives: dq,dv,M are 7x7 row-major (row = torque index) */
void orc_rnea_derivatives(const orc_model *m, const double *q, const double *v, const double *a,
                          double *tau, double *dtau_dq, double *dtau_dv, double *M);
/* the same partial derivatives and mass matrix in closed form (world-frame spatial algebra: the formulation of pinocchio::computeRNEADerivatives /
   crba, robot_ocp.hpp:118-122): an independent second implementation, checked against the directional one to round-off */
void orc_rnea_derivatives_analytic(const orc_model *m, const double *q, const double *v, const double *a,
                                   double *tau, double *dtau_dq, double *dtau_dv, double *M);
void orc_fk(const orc_model *m, const double *q, double *p_joint7, double *R_joint7 /*9*/,
            double *p_link8, double *p_tool);
/* world-aligned 6x7 Jacobian of a point rigidly attached to joint 7 at local offset `off` */
void orc_frame_jacobian(const orc_model *m, const double *q, const double *off, double *J /*6x7 row-major*/);
/* g(8) and dg/d[x,u,T] (8x22 row-major) exactly as robot_ocp.hpp:98-163 */
void orc_eval_constraints(const orc_model *m, int quirk, const double *x, const double *u,
                          double *g, double *G /*8x22 or NULL*/);

/* ---- discretisation ---- */
int  orc_num_nodes(int num_seg);
void orc_time_nodes(int num_seg, double *tau);   /* ascending, [0,1] */
void orc_diff_matrix(double *D /*4x4*/);         /* cubic CGL differentiation matrix, ascending nodes */

/* ---- warm start stand-in for Ruckig (motionPlanner.cpp:146-175) ---- */
void orc_warm_start(const orc_config *c, const double *amax_used, const double *x0, const double *xf,
                    double *xg, double *ug, double *Tg);

/* ---- jerk-limited, time-synchronised warm start (jerk.c): restatement of what Ruckig provides at motionPlanner.cpp:146-175;
 *      pinned by the stored Ruckig trajectory (tests/golden/gold_traj.json) ---- */
void orc_warm_start_jerk(int num_seg, const double *vmax, const double *amax, const double *jmax, const double *x0, const double *xf,
                         double *xg, double *ug, double *Tg);
/* out (n_pts+1) x 22 = t, q(7), v(7), a(7): get_ruckig_trajectory (motionPlanner.hpp:73-96) */
void orc_jerk_trajectory(const double *vmax, const double *amax, const double *jmax, const double *x0, const double *xf, int n_pts,
                         double *out, double *T_out);
/* the same with boundary accelerations acc0 / accT [7] (NULL = zero), as the reference forwards them to Ruckig (motionPlanner.cpp:36-38,50-52) */
void orc_warm_start_jerk_acc(int num_seg, const double *vmax, const double *amax, const double *jmax, const double *x0, const double *xf,
                             const double *acc0, const double *accT, double *xg, double *ug, double *Tg);
void orc_jerk_trajectory_acc(const double *vmax, const double *amax, const double *jmax, const double *x0, const double *xf,
                             const double *acc0, const double *accT, int n_pts, double *out, double *T_out);

/* ---- the hot path: one OCP ---- */
/* z layout: xs[N][14], us[N][7], T.  lam (optional, size m_eq+m_in+n) receives final multipliers */
void orc_solve(const orc_model *m, const orc_config *c, const double *x0, const double *xf,
               const double *xg, const double *ug, double Tg,
               double *xs, double *us, double *T, orc_info *info);
/* Multi-arm form (BASELINE.json configs[3], 14-DoF dual Panda = the same OCP with doubled sizes NX=28, NU=14, NG=16):
 * `m` points at narm consecutive models (independent 7-joint chains on one base; the base placement of a chain is folded into
 * its first joint placement).  x0, xf [14 narm] = [q(7 narm); qd(7 narm)]; xg, xs [N][14 narm]; ug, us [N][7 narm].
 * The per-arm limit tables of orc_config apply to every arm.  narm = 1 is orc_solve. */
void orc_solve_multi(const orc_model *m, int narm, const orc_config *c, const double *x0, const double *xf,
                     const double *xg, const double *ug, double Tg,
                     double *xs, double *us, double *T, orc_info *info);
/* a re-solve on one planner object (mpcmp_config.carry_multipliers): lam_io [orc_num_multipliers] in / out; the start only with c->carry_multipliers */
void orc_solve_carry(const orc_model *m, int narm, const orc_config *c, const double *x0, const double *xf,
                     const double *xg, const double *ug, double Tg, double *lam_io, double *xs, double *us, double *T, orc_info *info);
int orc_num_multipliers(const orc_config *c, int narm);
void orc_solve_batch_multi(const orc_model *m, int narm, const orc_config *c, int B, const double *x0, const double *xf,
                           const double *xg, const double *ug, const double *Tg,
                           double *xs, double *us, double *T, orc_info *info, int threads);
int orc_debug_qp_multi(const orc_model *m, int narm, const orc_config *c, const double *x0, const double *xf,
                       const double *xs, const double *us, double T, const double *lam, double *p, double *y);
/* batch helper used by the cpu_baseline leg: sequential loop, or `threads` std pthreads */
void orc_solve_batch(const orc_model *m, const orc_config *c, int B, const double *x0, const double *xf,
                     const double *xg, const double *ug, const double *Tg,
                     double *xs, double *us, double *T, orc_info *info, int threads);

/* ---- resampling (motionPlanner.hpp:99-128) ---- */
void orc_sample(const orc_model *m, int num_seg, const double *xs, const double *us, double T,
                int n_pts, double *out /* (n_pts+1) x 29: t,q,v,a,tau */);

/* get_MPC_point (motionPlanner.hpp:118-128) incl. its clamp; out = q(7), v(7), a(7), tau(7) */
void orc_mpc_point(const orc_model *m, int num_seg, const double *xs, const double *us, double T, double time, double *out);

/* state advance + arrival rule of the receding-horizon driver (mpcmp_rh_run, include/mpcmp.h): x_io [14] in / out; returns 1 if the instance is retired */
int orc_rh_advance(const orc_config *c, const double *xs, const double *us, double T, int status, double dt, const double *xf, double *x_io);

/* examples/benchmark.cpp:58-160: out[74] = min(28) | max(28) | x(T)-target (14) | flags jerk, lin vel, ang vel, collision */
void orc_traj_stats(const orc_model *m, int num_seg, const double *xs, const double *us, double T, const double *xf,
                    int n_pts, double *out);

/* collocation defects at all four local nodes of every segment: out [num_seg][4][14] (see ocp.c) */
void orc_collocation_defects(int num_seg, const double *xs, const double *us, double T, double *out);

/* ---- pieces exposed for unit tests of the QP layer ---- */
/* Assemble the QP of one SQP iteration at (xs,us,T,lam) and run ADMM; returns iterations used. */
int orc_debug_qp(const orc_model *m, const orc_config *c, const double *x0, const double *xf,
                 const double *xs, const double *us, double T, const double *lam,
                 double *p /*n*/, double *y /*m+n*/);

#ifdef __cplusplus
}
#endif
#endif
