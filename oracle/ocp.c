/*
 * ocp.c — NLP / QP / SQP layer of the CPU oracle (TEST INFRASTRUCTURE, see oracle.h).
 *
 * Restates, for one (start,target) pair, what MotionPlanner::solve_trajectory -> mpc.solve()
 * computes (mpc_solver/motionPlanner.cpp:177-208):
 *   - minTime_ocp                       mpc_solver/robot_ocp.hpp:31-213
 *   - MySolver (SQP customisation)      mpc_solver/polympc_redef.hpp:41-147
 *   - boxADMM QP                        mpc_solver/motionPlanner.hpp:10-11 (type alias only)
 *   - solver settings / bounds          mpc_solver/motionPlanner.cpp:15-20, 27-100
 * polympc itself (collocation assembly, SQPBase::solve, boxADMM::solve) is an EMPTY submodule in
 * /root/reference (.gitmodules:1-4), so that layer follows the published algorithms
 * (Chebyshev-Gauss-Lobatto collocation on splines; OSQP-form ADMM, Stellato et al. 2020;
 * l1-merit line-search SQP, Nocedal&Wright ch.18) with every free choice explicit in orc_config.
 * PARITY UNPINNED at digit level for this layer (see oracle.h).
 *
 * NLP:
 *   z = [x_0..x_{N-1} | u_0..u_{N-1} | T],  N = 3*NUM_SEG+1, ascending time
 *   min T
 *   s.t. sum_j D[i][j] x_{3s+j} - ts*T*[qd_k;u_k] = 0     k=3s+i, i in {0,1,2}    (14 rows / node)
 *        lbg <= [rnea(q_k,qd_k,u_k); z_tool(q_k)] <= ubg   all nodes              ( 8 rows / node)
 *        x_0 = x_start, x_{N-1} in x_target +- eps, boxes on x,u,T
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define MAXN (3 * ORC_MAXSEG + 1)
#define MAXROWNZ 24

int orc_num_nodes(int num_seg) { return 3 * num_seg + 1; }

void orc_time_nodes(int num_seg, double *tau) {
    /* cubic Chebyshev-Gauss-Lobatto points xi = -cos(pi*j/3) = {-1,-1/2,1/2,1} on each segment */
    static const double xi[4] = {-1.0, -0.5, 0.5, 1.0};
    for (int s = 0; s < num_seg; s++)
        for (int j = 0; j < 4; j++) tau[3 * s + j] = (s + 0.5 * (xi[j] + 1.0)) / num_seg;
}

void orc_diff_matrix(double *D) {
    /* Lagrange differentiation matrix on xi = {-1,-1/2,1/2,1}: D[i][j] = l_j'(xi_i) */
    static const double xi[4] = {-1.0, -0.5, 0.5, 1.0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double den = 1.0;
            for (int k = 0; k < 4; k++) if (k != j) den *= (xi[j] - xi[k]);
            double num = 0.0;
            for (int l = 0; l < 4; l++) if (l != j) {
                double pr = 1.0;
                for (int k = 0; k < 4; k++) if (k != j && k != l) pr *= (xi[i] - xi[k]);
                num += pr;
            }
            D[4 * i + j] = num / den;
        }
}

void orc_set_margins(orc_config *c, double mp, double mv, double ma, double mt) {
    /* motionPlanner.cpp:56-100 */
    double qmin[7], qmax[7], vmax[7], amax[7], tmax[7];
    orc_default_limits(qmin, qmax, vmax, amax, 0, tmax);
    for (int j = 0; j < 7; j++) {
        double s = (1.0 - mp) * (qmax[j] - qmin[j]) / 2.0;
        c->lbx[j] = qmin[j] + s; c->ubx[j] = qmax[j] - s;
        c->lbx[7 + j] = -mv * vmax[j]; c->ubx[7 + j] = mv * vmax[j];
        c->lbu[j] = -ma * amax[j]; c->ubu[j] = ma * amax[j];
        c->lbg[j] = -mt * tmax[j]; c->ubg[j] = mt * tmax[j];
    }
    c->lbg[7] = 0.05; c->ubg[7] = INFINITY;   /* pandaWrapper.hpp:40, motionPlanner.cpp:95-96 */
    c->lbT = 0.0; c->ubT = 10.0;              /* motionPlanner.cpp:77-78 */
}

void orc_default_config(orc_config *c, int num_seg, int sqp_iters) {
    memset(c, 0, sizeof *c);
    c->num_seg = num_seg; c->sqp_iters = sqp_iters;
    c->qp_iters = 700; c->ls_iters = 10;       /* motionPlanner.cpp:16-17 */
    c->check_every = 25; c->quirk_dtau_dT = 1;
    c->eps_abs = 1e-3; c->eps_rel = 1e-3;      /* motionPlanner.cpp:19-20 */
    c->rho = 0.02; c->sigma = 1e-6; c->alpha = 1.4; c->rho_eq_scale = 1e3;   /* chosen by tools/polympc_param_fit.py on GOLD-TRAJ (profiles/r02_polympc_param_fit.json) */
    c->ls_eta = 0.25; c->ls_tau = 0.5;
    c->hess_reg = 1e-3;                        /* polympc_redef.hpp:68 */
    c->eps_target = 1e-2;                      /* motionPlanner.hpp:44 */
    orc_set_margins(c, 1.0, 1.0, 1.0, 1.0);    /* motionPlanner.cpp:24 */
}

/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int N, n, meq, min, m;      /* nodes, variables, rows */
    int narm, nq, nx, nu, ng;   /* arms; per node: positions 7*narm, states 14*narm, controls 7*narm, path rows 8*narm */
    double ts;
    double D[16];
    /* sparse general rows */
    int    *nnz; int *col; double *val;   /* m x MAXROWNZ */
    double *ceq, *g;            /* constraint values */
    double *Gk;                 /* N x narm x 8 x 22 path Jacobians (one evalConstraints block per arm) */
    double *hd, *ha;            /* Hessian: diagonal (n) and arrow (n, entry T unused) */
    double *l, *u, *rho;        /* QP bounds and ADMM rho, size m+n */
    double *zlb, *zub;          /* variable boxes, size n */
    /* linear solver */
    int *perm, *iperm, *first;  /* internal->external, external->internal, skyline start */
    double *K;                  /* n x n (permuted), overwritten by L */
} work;

/* Multi-arm generalisation (BASELINE.json configs[3], 14-DoF dual Panda: the same OCP, robot_ocp.hpp:38,80-163, with doubled
   sizes NX=28, NU=14, NG=16): the "robot" is narm independent 7-joint chains on one base, state x_k = [q(7 narm); qd(7 narm)],
   control u_k = qdd(7 narm), path rows g_k = [tau(7 narm); z_tool of every arm (narm)].  narm = 1 is the reference as is. */
static inline int IX(const work *w, int k, int r) { return w->nx * k + r; }
static inline int IU(const work *w, int k, int r) { return w->nx * w->N + w->nu * k + r; }
static inline int IT(const work *w) { return (w->nx + w->nu) * w->N; }
/* component of the per-arm limit tables (lbx/ubx: [q(7); qd(7)]) that state entry r of a node refers to */
static inline int XCOMP(const work *w, int r) { return r < w->nq ? r % 7 : 7 + (r - w->nq) % 7; }

static work *work_new(const orc_config *c, int narm) {
    work *w = (work *)calloc(1, sizeof(work));
    int N = w->N = orc_num_nodes(c->num_seg);
    w->narm = narm; w->nq = 7 * narm; w->nx = 14 * narm; w->nu = 7 * narm; w->ng = 8 * narm;
    w->n = (w->nx + w->nu) * N + 1; w->meq = w->nx * (N - 1); w->min = w->ng * N; w->m = w->meq + w->min;
    w->ts = 1.0 / (2.0 * c->num_seg);       /* (t_f - t_0)/(2 NUM_SEG) with set_time_limits(0,1), motionPlanner.cpp:18 */
    orc_diff_matrix(w->D);
    int n = w->n, m = w->m;
    w->nnz = (int *)calloc(m, sizeof(int)); w->col = (int *)calloc((size_t)m * MAXROWNZ, sizeof(int));
    w->val = (double *)calloc((size_t)m * MAXROWNZ, sizeof(double));
    w->ceq = (double *)calloc(w->meq, sizeof(double)); w->g = (double *)calloc(w->min, sizeof(double));
    w->Gk = (double *)calloc((size_t)N * narm * 8 * 22, sizeof(double));
    w->hd = (double *)calloc(n, sizeof(double)); w->ha = (double *)calloc(n, sizeof(double));
    w->l = (double *)calloc(m + n, sizeof(double)); w->u = (double *)calloc(m + n, sizeof(double));
    w->rho = (double *)calloc(m + n, sizeof(double));
    w->zlb = (double *)calloc(n, sizeof(double)); w->zub = (double *)calloc(n, sizeof(double));
    w->perm = (int *)calloc(n, sizeof(int)); w->iperm = (int *)calloc(n, sizeof(int));
    w->first = (int *)calloc(n, sizeof(int));
    w->K = (double *)calloc((size_t)n * n, sizeof(double));
    /* chain ordering, arm after arm (the arms couple only through T):
       [x_0 | u_3s, x_3s+1, u_3s+1, x_3s+2, u_3s+2, x_3s+3 | ... | u_{N-1}] per arm, then T;  x_k of an arm = its q then its qd */
    int p = 0;
    for (int a = 0; a < narm; a++) {
#define PUSH_X(k) do { for (int r = 0; r < 7; r++) w->perm[p++] = IX(w, (k), 7 * a + r); \
                       for (int r = 0; r < 7; r++) w->perm[p++] = IX(w, (k), w->nq + 7 * a + r); } while (0)
#define PUSH_U(k) do { for (int r = 0; r < 7; r++) w->perm[p++] = IU(w, (k), 7 * a + r); } while (0)
        PUSH_X(0);
        for (int s = 0; s < c->num_seg; s++) {
            PUSH_U(3 * s);
            for (int i = 1; i <= 2; i++) { PUSH_X(3 * s + i); PUSH_U(3 * s + i); }
            PUSH_X(3 * s + 3);
        }
        PUSH_U(N - 1);
#undef PUSH_X
#undef PUSH_U
    }
    w->perm[p++] = IT(w);
    for (int i = 0; i < n; i++) w->iperm[w->perm[i]] = i;
    return w;
}
static void work_free(work *w) {
    free(w->nnz); free(w->col); free(w->val); free(w->ceq); free(w->g); free(w->Gk); free(w->hd); free(w->ha);
    free(w->l); free(w->u); free(w->rho); free(w->zlb); free(w->zub); free(w->perm); free(w->iperm);
    free(w->first); free(w->K); free(w);
}

/* structure probes against the reference's ONE stored solve; a bit mask in the environment, 0 = the specification */
#ifdef ORC_PROBE_BUILD      /* hypothesis probes (tools/polympc_param_fit.py --probe): only in liboracle_probe.so (make -C oracle probe) */
static int orc_probe(void) { const char *e = getenv("ORC_PROBE"); return e ? atoi(e) : 0; }
#else                       /* liboracle.so, the specification every parity test and the CPU baseline use: the probes are compiled out */
static int orc_probe(void) { return 0; }
#endif

static void set_boxes(work *w, const orc_config *c, const double *x0, const double *xf) {
    int N = w->N;
    for (int k = 0; k < N; k++) {
        for (int r = 0; r < w->nx; r++) {
            double lo = c->lbx[XCOMP(w, r)], hi = c->ubx[XCOMP(w, r)];
            if (k == 0) { lo = hi = x0[r]; }                                           /* motionPlanner.cpp:47 */
            if (k == N - 1) { lo = xf[r] - c->eps_target; hi = xf[r] + c->eps_target; } /* motionPlanner.cpp:33 */
            w->zlb[IX(w, k, r)] = lo; w->zub[IX(w, k, r)] = hi;
        }
        for (int r = 0; r < w->nu; r++) { w->zlb[IU(w, k, r)] = c->lbu[r % 7]; w->zub[IU(w, k, r)] = c->ubu[r % 7]; }
    }
    w->zlb[IT(w)] = c->lbT; w->zub[IT(w)] = c->ubT;
}

/* [q_a; qd_a] (14) and qdd_a (7) of arm a at node k */
static void arm_state(const work *w, const double *z, int k, int a, double *xa, double *ua) {
    for (int j = 0; j < 7; j++) {
        xa[j] = z[IX(w, k, 7 * a + j)]; xa[7 + j] = z[IX(w, k, w->nq + 7 * a + j)];
        ua[j] = z[IU(w, k, 7 * a + j)];
    }
}

/* values of all constraints at z (no derivatives): used by the line search (robot_ocp.hpp:80-96) */
static void eval_values(const orc_model *mdl, const work *w, const double *z, double *ceq, double *g) {
    int N = w->N; double T = z[IT(w)];
    for (int k = 0; k < N - 1; k++) {
        int s = k / 3, i = k % 3;
        for (int r = 0; r < w->nx; r++) {
            double acc = 0.0;
            for (int j = 0; j < 4; j++) acc += w->D[4 * i + j] * z[IX(w, 3 * s + j, r)];
            double f = (r < w->nq) ? z[IX(w, k, w->nq + r)] : z[IU(w, k, r - w->nq)];   /* robot_ocp.hpp:55-73 */
            ceq[w->nx * k + r] = acc - w->ts * T * f;
        }
    }
    for (int k = 0; k < N; k++)
        for (int a = 0; a < w->narm; a++) {
            double xa[14], ua[7], ga[8];
            arm_state(w, z, k, a, xa, ua);
            orc_eval_constraints(mdl + a, 0, xa, ua, ga, 0);
            for (int j = 0; j < 7; j++) g[w->ng * k + 7 * a + j] = ga[j];
            g[w->ng * k + w->nq + a] = ga[7];
        }
}

static double viol(double v, double lo, double hi) { return v < lo ? lo - v : (v > hi ? v - hi : 0.0); }
/* index into the per-arm bound tables lbg/ubg ([tau(7); z_tool]) of path row r of a node */
static inline int GCOMP(const work *w, int r) { return r < w->nq ? r % 7 : 7; }

static double l1_violation(const work *w, const orc_config *c, const double *z, const double *ceq, const double *g) {
    double s = 0.0;
    for (int i = 0; i < w->meq; i++) s += fabs(ceq[i]);
    for (int k = 0; k < w->N; k++) for (int r = 0; r < w->ng; r++) s += viol(g[w->ng * k + r], c->lbg[GCOMP(w, r)], c->ubg[GCOMP(w, r)]);
    for (int i = 0; i < w->n; i++) s += viol(z[i], w->zlb[i], w->zub[i]);
    return s;
}

/* full linearisation at (z, lam): rows of A, constraint values, Hessian (polympc_redef.hpp:133-147 forces this
   every iteration), QP bounds */
static void linearise(const orc_model *mdl, const orc_config *c, work *w, const double *z, const double *lam) {
    int N = w->N, n = w->n; double T = z[IT(w)], ts = w->ts;
    memset(w->hd, 0, sizeof(double) * n); memset(w->ha, 0, sizeof(double) * n);
    for (int k = 0; k < N - 1; k++) {
        int s = k / 3, i = k % 3;
        for (int r = 0; r < w->nx; r++) {
            int row = w->nx * k + r, nz = 0;
            int *col = w->col + (size_t)row * MAXROWNZ; double *val = w->val + (size_t)row * MAXROWNZ;
            double acc = 0.0;
            for (int j = 0; j < 4; j++) {
                double d = w->D[4 * i + j];
                acc += d * z[IX(w, 3 * s + j, r)];
                col[nz] = IX(w, 3 * s + j, r); val[nz++] = d;
            }
            int fcol = (r < w->nq) ? IX(w, k, w->nq + r) : IU(w, k, r - w->nq);
            double f = z[fcol];
            col[nz] = fcol; val[nz++] = -ts * T;      /* never coincides with a D column */
            col[nz] = IT(w); val[nz++] = -ts * f;
            w->nnz[row] = nz;
            w->ceq[row] = acc - ts * T * f;
            /* exact Lagrangian Hessian: only d2/dT d(f-variable) = -ts * lam_row */
            w->ha[fcol] += -ts * lam[row];
        }
    }
    for (int k = 0; k < N; k++)
        for (int a = 0; a < w->narm; a++) {
            double *G = w->Gk + ((size_t)k * w->narm + a) * 176;
            double xa[14], ua[7], ga[8];
            arm_state(w, z, k, a, xa, ua);
            orc_eval_constraints(mdl + a, c->quirk_dtau_dT, xa, ua, ga, G);
            for (int j = 0; j < 7; j++) w->g[w->ng * k + 7 * a + j] = ga[j];
            w->g[w->ng * k + w->nq + a] = ga[7];
            for (int r = 0; r < 8; r++) {
                int row = w->meq + w->ng * k + (r < 7 ? 7 * a + r : w->nq + a), nz = 0;
                int *col = w->col + (size_t)row * MAXROWNZ; double *val = w->val + (size_t)row * MAXROWNZ;
                if (r < 7) {
                    for (int j = 0; j < 7; j++) { col[nz] = IX(w, k, 7 * a + j); val[nz++] = G[22 * r + j]; }
                    for (int j = 0; j < 7; j++) { col[nz] = IX(w, k, w->nq + 7 * a + j); val[nz++] = G[22 * r + 7 + j]; }
                    for (int j = 0; j < 7; j++) { col[nz] = IU(w, k, 7 * a + j); val[nz++] = G[22 * r + 14 + j]; }
                    col[nz] = IT(w); val[nz++] = G[22 * r + 21];
                } else {
                    for (int j = 0; j < 7; j++) { col[nz] = IX(w, k, 7 * a + j); val[nz++] = G[22 * r + j]; }
                }
                w->nnz[row] = nz;
            }
        }
    /* Gershgorin regularisation, polympc_redef.hpp:57-70 (sparse variant, +hess_reg) */
    double rT = 0.0;
    for (int i = 0; i < n - 1; i++) {
        double ri = fabs(w->ha[i]);
        rT += ri;
        w->hd[i] = ri + c->hess_reg;      /* a_ii = 0 -> a_ii - r_i <= 0 always */
    }
    w->hd[n - 1] = rT + c->hess_reg;
    /* QP bounds in the step p */
    for (int i = 0; i < w->meq; i++) { w->l[i] = w->u[i] = -w->ceq[i]; }
    for (int k = 0; k < N; k++) for (int r = 0; r < w->ng; r++) {
        w->l[w->meq + w->ng * k + r] = c->lbg[GCOMP(w, r)] - w->g[w->ng * k + r];
        w->u[w->meq + w->ng * k + r] = c->ubg[GCOMP(w, r)] - w->g[w->ng * k + r];
    }
    for (int i = 0; i < n; i++) { w->l[w->m + i] = w->zlb[i] - z[i]; w->u[w->m + i] = w->zub[i] - z[i]; }
    /* per-row rho: OSQP rule, equality rows (l==u) scaled by rho_eq_scale */
    for (int i = 0; i < w->m + n; i++) {
        w->rho[i] = (w->u[i] - w->l[i] < 1e-4) ? c->rho * c->rho_eq_scale : c->rho;
        /* HYPOTHESIS PROBE (tools/polympc_param_fit.py --probe; off unless ORC_PROBE has bit 0): variable boxes keep the plain rho even when
           they pin a variable (x_0); only general equality rows get rho_eq */
        if (orc_probe() & 1 && i >= w->m) w->rho[i] = c->rho;
    }
}

/* K = H + sigma I + diag(rho_box) + A^T diag(rho) A in the chain ordering, then skyline Cholesky */
static int factor(const orc_config *c, work *w) {
    int n = w->n, m = w->m;
    double *K = w->K;
    memset(K, 0, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; i++) {
        int pi = w->iperm[i];
        K[(size_t)pi * n + pi] += w->hd[i] + c->sigma + w->rho[m + i];
        if (i != n - 1 && w->ha[i] != 0.0) {
            int pT = w->iperm[n - 1];
            int a = pi > pT ? pi : pT, b = pi > pT ? pT : pi;
            K[(size_t)a * n + b] += w->ha[i];
        }
    }
    for (int r = 0; r < m; r++) {
        const int *col = w->col + (size_t)r * MAXROWNZ; const double *val = w->val + (size_t)r * MAXROWNZ;
        int nz = w->nnz[r]; double rho = w->rho[r];
        for (int a = 0; a < nz; a++) for (int b = 0; b < nz; b++) {
            int pa = w->iperm[col[a]], pb = w->iperm[col[b]];
            if (pa >= pb) K[(size_t)pa * n + pb] += rho * val[a] * val[b];
        }
    }
    for (int i = 0; i < n; i++) {
        int f = i;
        for (int j = 0; j < i; j++) if (K[(size_t)i * n + j] != 0.0) { f = j; break; }
        w->first[i] = f;
    }
    for (int i = 0; i < n; i++) {
        int fi = w->first[i];
        for (int j = fi; j <= i; j++) {
            int fj = w->first[j], k0 = fi > fj ? fi : fj;
            double s = K[(size_t)i * n + j];
            for (int k = k0; k < j; k++) s -= K[(size_t)i * n + k] * K[(size_t)j * n + k];
            if (j < i) K[(size_t)i * n + j] = s / K[(size_t)j * n + j];
            else { if (!(s > 0.0)) return 1; K[(size_t)i * n + i] = sqrt(s); }
        }
    }
    return 0;
}

static void kkt_solve(const work *w, const double *b /*external order*/, double *x /*external order*/, double *tmp) {
    int n = w->n; const double *L = w->K;
    for (int i = 0; i < n; i++) {
        double s = b[w->perm[i]];
        for (int k = w->first[i]; k < i; k++) s -= L[(size_t)i * n + k] * tmp[k];
        tmp[i] = s / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double xi = tmp[i] / L[(size_t)i * n + i];
        tmp[i] = xi;
        for (int k = w->first[i]; k < i; k++) tmp[k] -= L[(size_t)i * n + k] * xi;
    }
    for (int i = 0; i < n; i++) x[w->perm[i]] = tmp[i];
}

static void A_mul(const work *w, const double *x, double *out /*m*/) {
    for (int r = 0; r < w->m; r++) {
        const int *col = w->col + (size_t)r * MAXROWNZ; const double *val = w->val + (size_t)r * MAXROWNZ;
        double s = 0.0;
        for (int a = 0; a < w->nnz[r]; a++) s += val[a] * x[col[a]];
        out[r] = s;
    }
}
static void AT_mul_add(const work *w, const double *y /*m*/, double *out /*n, accumulated*/) {
    for (int r = 0; r < w->m; r++) {
        const int *col = w->col + (size_t)r * MAXROWNZ; const double *val = w->val + (size_t)r * MAXROWNZ;
        double yr = y[r];
        for (int a = 0; a < w->nnz[r]; a++) out[col[a]] += val[a] * yr;
    }
}
static void H_mul(const work *w, const double *x, double *out) {
    int n = w->n; double xT = x[n - 1], s = w->hd[n - 1] * xT;
    for (int i = 0; i < n - 1; i++) { out[i] = w->hd[i] * x[i] + w->ha[i] * xT; s += w->ha[i] * x[i]; }
    out[n - 1] = s;
}
static double inf_norm(const double *v, int n) { double s = 0; for (int i = 0; i < n; i++) { double a = fabs(v[i]); if (a > s) s = a; } return s; }

/* box-ADMM in OSQP form on the stacked constraint [A; I], reduced KKT. cold start. */
/* lam0: the NLP multipliers (rows, then variables) the QP's duals start from when c->qp_warm_start is set; NULL or flag 0: cold start */
static int admm(const orc_config *c, work *w, double *x, double *y, int *status, const double *lam0) {
    int n = w->n, m = w->m, mn = m + n;
    double *zz = (double *)calloc(mn, sizeof(double)), *xt = (double *)calloc(n, sizeof(double));
    double *rhs = (double *)calloc(n, sizeof(double)), *zt = (double *)calloc(mn, sizeof(double));
    double *tmp = (double *)calloc(n, sizeof(double)), *wv = (double *)calloc(mn, sizeof(double));
    double *t1 = (double *)calloc(mn, sizeof(double)), *t2 = (double *)calloc(n, sizeof(double));
    memset(x, 0, sizeof(double) * n); memset(y, 0, sizeof(double) * mn);
    if (lam0 && c->qp_warm_start)         /* warm duals: y_0 = lambda_k, x_0 = 0, z_0 = clip([A; I] x_0, l, u) */
        for (int i = 0; i < mn; i++) { y[i] = lam0[i]; zz[i] = 0.0 < w->l[i] ? w->l[i] : (0.0 > w->u[i] ? w->u[i] : 0.0); }
    const double sigma = c->sigma, alpha = c->alpha;
    int it = 0;
    double rho_cur = c->rho;
    const char *adapt_env = getenv("ORC_ADAPT");
    const double adapt_tol = adapt_env ? atof(adapt_env) : 0.0;
    *status = 0;
    if (factor(c, w)) { *status = 2; goto done; }
    for (it = 1; it <= c->qp_iters; it++) {
        for (int i = 0; i < mn; i++) wv[i] = w->rho[i] * zz[i] - y[i];
        for (int i = 0; i < n; i++) rhs[i] = sigma * x[i] + wv[m + i];
        rhs[n - 1] -= 1.0;                           /* q = cost gradient = e_T (robot_ocp.hpp:201-213) */
        AT_mul_add(w, wv, rhs);
        kkt_solve(w, rhs, xt, tmp);
        A_mul(w, xt, zt);
        for (int i = 0; i < n; i++) zt[m + i] = xt[i];
        for (int i = 0; i < n; i++) x[i] = alpha * xt[i] + (1.0 - alpha) * x[i];
        for (int i = 0; i < mn; i++) {
            double zr = alpha * zt[i] + (1.0 - alpha) * zz[i];
            double zn = zr + y[i] / w->rho[i];
            zn = zn < w->l[i] ? w->l[i] : (zn > w->u[i] ? w->u[i] : zn);
            y[i] += w->rho[i] * (zr - zn);
            zz[i] = zn;
        }
        if (it % c->check_every == 0) {
            /* r_prim = ||[A;I]x - z||, r_dual = ||Hx + q + [A;I]^T y|| */
            A_mul(w, x, t1);
            for (int i = 0; i < n; i++) t1[m + i] = x[i];
            double nAx = inf_norm(t1, mn), nz = inf_norm(zz, mn), rp = 0.0;
            for (int i = 0; i < mn; i++) { double d = fabs(t1[i] - zz[i]); if (d > rp) rp = d; }
            H_mul(w, x, t2);
            double nHx = inf_norm(t2, n);
            for (int i = 0; i < n; i++) tmp[i] = y[m + i];
            AT_mul_add(w, y, tmp);
            double nAty = inf_norm(tmp, n), rd = 0.0;
            for (int i = 0; i < n; i++) { double d = fabs(t2[i] + tmp[i] + (i == n - 1 ? 1.0 : 0.0)); if (d > rd) rd = d; }
            double ep = c->eps_abs + c->eps_rel * (nAx > nz ? nAx : nz);
            double mx = nHx > nAty ? nHx : nAty; if (mx < 1.0) mx = 1.0;   /* ||q||_inf = 1 */
            double ed = c->eps_abs + c->eps_rel * mx;
            if (rp <= ep && rd <= ed) break;
            if (adapt_tol > 0.0) {
                /* HYPOTHESIS PROBE (tools/polympc_adaptive_rho_probe.py; off unless ORC_ADAPT=<tolerance> is set): OSQP's adaptive rho,
                   rho <- rho sqrt((r_p / max(|Ax|,|z|)) / (r_d / max(|Hx|,|A^T y|,|q|))), applied with a refactorisation when it
                   changes by more than the tolerance.  polympc is absent, so whether its boxADMM adapts rho cannot be read; against
                   the reference's stored solve the probe is 3x - 50x worse than a fixed rho: not adopted. */
                double pn = rp / ((nAx > nz ? nAx : nz) + 1e-10), dn = rd / (mx + 1e-10);
                double est = rho_cur * sqrt(pn / (dn + 1e-10));
                if (est < 1e-6) est = 1e-6;
                if (est > 1e6) est = 1e6;
                if (est > rho_cur * adapt_tol || est < rho_cur / adapt_tol) {
                    const double f = est / rho_cur;
                    for (int i = 0; i < mn; i++) w->rho[i] *= f;
                    rho_cur = est;
                    if (factor(c, w)) { *status = 2; break; }
                }
            }
        }
    }
    if (it > c->qp_iters) { it = c->qp_iters; *status |= 8; }    /* ran out of iterations without meeting the termination test */
    if (orc_probe() & 2) for (int i = 0; i < n; i++) x[i] = zz[m + i];      /* PROBE: return the projected copy z of the step, not x */
done:
    free(zz); free(xt); free(rhs); free(zt); free(tmp); free(wv); free(t1); free(t2);
    return it;
}

static void pack(const work *w, const double *xs, const double *us, double T, double *z) {
    memcpy(z, xs, sizeof(double) * w->nx * w->N); memcpy(z + w->nx * w->N, us, sizeof(double) * w->nu * w->N); z[IT(w)] = T;
}

/* one OCP for narm arms (narm consecutive orc_model): x0, xf [14 narm] = [q(7 narm); qd(7 narm)], xg/xs [N][14 narm],
   ug/us [N][7 narm] */
/* lam_io [m + n] (rows, then variables), may be NULL: with c->carry_multipliers the SQP starts from it instead of lambda_0 = 0 (mpcmp_config.carry_multipliers:
   the multipliers the previous solve of the same planner left behind); the final multipliers are written back whatever the flag says */
static void solve_core(const orc_model *mdl, int narm, const orc_config *c, const double *x0, const double *xf,
                       const double *xg, const double *ug, double Tg, double *lam_io, double *xs, double *us, double *Tout, orc_info *info) {
    work *w = work_new(c, narm);
    int n = w->n, m = w->m, mn = m + n;
    double *z = (double *)calloc(n, sizeof(double)), *lam = (double *)calloc(mn, sizeof(double));
    if (lam_io && c->carry_multipliers) memcpy(lam, lam_io, sizeof(double) * mn);
    double *p = (double *)calloc(n, sizeof(double)), *y = (double *)calloc(mn, sizeof(double));
    double *zs = (double *)calloc(n, sizeof(double)), *ce = (double *)calloc(w->meq, sizeof(double));
    double *gg = (double *)calloc(w->min, sizeof(double));
    orc_info inf; memset(&inf, 0, sizeof inf);
    pack(w, xg, ug, Tg, z);
    set_boxes(w, c, x0, xf);
    if (orc_probe() & 4) for (int r = 0; r < w->nu; r++) w->zlb[IU(w, w->N - 1, r)] = w->zub[IU(w, w->N - 1, r)] = z[IU(w, w->N - 1, r)];   /* PROBE: u_{N-1} pinned to the warm start */
    linearise(mdl, c, w, z, lam);
    for (int it = 0; it < c->sqp_iters; it++) {
        int st;
        inf.qp_iters_total += admm(c, w, p, y, &st, lam);
        if (st) inf.status |= st;
        if (st & 8) inf.qp_capped++;
        /* l1 merit line search, polympc_redef.hpp:73-121 */
        double mu = inf_norm(lam, mn);                                   /* :86 */
        double constr = l1_violation(w, c, z, w->ceq, w->g);             /* :79 */
        double phi = z[n - 1] + mu * constr;                             /* :93 */
        double Dphi = p[n - 1] - mu * constr;                            /* :94  cost gradient = e_T */
        double alpha = 1.0;
        for (int i = 1; i < c->ls_iters; i++) {                          /* :97 */
            for (int k = 0; k < n; k++) zs[k] = z[k] + alpha * p[k];
            eval_values(mdl, w, zs, ce, gg);
            double phis = zs[n - 1] + mu * l1_violation(w, c, zs, ce, gg);
            if (phis <= phi + alpha * c->ls_eta * Dphi) break;            /* :108 */
            alpha *= c->ls_tau;
        }
        for (int k = 0; k < n; k++) z[k] += alpha * p[k];
        for (int k = 0; k < mn; k++) lam[k] += alpha * (y[k] - lam[k]);
        inf.last_alpha = alpha; inf.sqp_iters = it + 1;
        linearise(mdl, c, w, z, lam);
    }
    /* report */
    inf.T = z[n - 1];
    inf.viol_l1 = l1_violation(w, c, z, w->ceq, w->g);
    inf.defect_inf = inf_norm(w->ceq, w->meq);
    for (int k = 0; k < w->N; k++) for (int r = 0; r < w->ng; r++) {
        double v = viol(w->g[w->ng * k + r], c->lbg[GCOMP(w, r)], c->ubg[GCOMP(w, r)]); if (v > inf.path_viol_inf) inf.path_viol_inf = v;
    }
    for (int r = 0; r < w->nx; r++) { double d = fabs(z[IX(w, w->N - 1, r)] - xf[r]); if (d > inf.term_err_inf) inf.term_err_inf = d; }
    for (int k = 0; k < n; k++) if (!isfinite(z[k])) inf.status |= 1;
    if (inf.defect_inf > c->eps_abs || inf.path_viol_inf > c->eps_abs || inf.term_err_inf > c->eps_target + c->eps_abs) inf.status |= 16;
    if (!(inf.T >= c->lbT - 1e-9 && inf.T <= c->ubT + 1e-9)) inf.status |= 32;
    memcpy(xs, z, sizeof(double) * w->nx * w->N); memcpy(us, z + w->nx * w->N, sizeof(double) * w->nu * w->N); *Tout = z[n - 1];
    if (info) *info = inf;
    /* what the next solve of this planner may start from: nothing after a hard failure or a final time outside its box (the multipliers of a solve
       that left the feasible region of T are not a start; mpcmp.h, carry_multipliers) */
    if (lam_io) {
        if (inf.status & (1 | 2 | 4 | 32)) memset(lam_io, 0, sizeof(double) * mn);
        else memcpy(lam_io, lam, sizeof(double) * mn);
    }
    free(z); free(lam); free(p); free(y); free(zs); free(ce); free(gg);
    work_free(w);
}

void orc_solve_multi(const orc_model *mdl, int narm, const orc_config *c, const double *x0, const double *xf,
                     const double *xg, const double *ug, double Tg, double *xs, double *us, double *Tout, orc_info *info) {
    solve_core(mdl, narm, c, x0, xf, xg, ug, Tg, NULL, xs, us, Tout, info);
}
/* a re-solve on one planner object: lam_io [m + n] in / out (m + n = orc_num_multipliers) */
void orc_solve_carry(const orc_model *mdl, int narm, const orc_config *c, const double *x0, const double *xf,
                     const double *xg, const double *ug, double Tg, double *lam_io, double *xs, double *us, double *Tout, orc_info *info) {
    solve_core(mdl, narm, c, x0, xf, xg, ug, Tg, lam_io, xs, us, Tout, info);
}
int orc_num_multipliers(const orc_config *c, int narm) {
    work *w = work_new(c, narm);
    const int mn = w->m + w->n;
    work_free(w);
    return mn;
}

void orc_solve(const orc_model *mdl, const orc_config *c, const double *x0, const double *xf,
               const double *xg, const double *ug, double Tg, double *xs, double *us, double *Tout, orc_info *info) {
    orc_solve_multi(mdl, 1, c, x0, xf, xg, ug, Tg, xs, us, Tout, info);
}

/* Collocation defects of a node trajectory at ALL four local nodes of every segment (the NLP itself only constrains the first
   three): out[s][i][r] = sum_j D[i][j] x_{3s+j}[r] - ts*T*f_{3s+i}[r],  f = [qd; u] (robot_ocp.hpp:55-73).
   Used by the tests to check the discretisation (differentiation matrix, time scaling, row placement) against the reference's
   stored solve. */
void orc_collocation_defects(int num_seg, const double *xs, const double *us, double T, double *out) {
    double D[16];
    orc_diff_matrix(D);
    const double ts = 1.0 / (2.0 * num_seg);
    for (int s = 0; s < num_seg; s++)
        for (int i = 0; i < 4; i++)
            for (int r = 0; r < 14; r++) {
                double acc = 0.0;
                for (int j = 0; j < 4; j++) acc += D[4 * i + j] * xs[14 * (3 * s + j) + r];
                const int k = 3 * s + i;
                const double f = (r < 7) ? xs[14 * k + 7 + r] : us[7 * k + (r - 7)];
                out[(s * 4 + i) * 14 + r] = acc - ts * T * f;
            }
}

int orc_debug_qp_multi(const orc_model *mdl, int narm, const orc_config *c, const double *x0, const double *xf,
                       const double *xs, const double *us, double T, const double *lam, double *p, double *y) {
    work *w = work_new(c, narm);
    double *z = (double *)calloc(w->n, sizeof(double));
    double *l0 = (double *)calloc(w->m + w->n, sizeof(double));
    pack(w, xs, us, T, z);
    set_boxes(w, c, x0, xf);
    linearise(mdl, c, w, z, lam ? lam : l0);
    int st, it = admm(c, w, p, y, &st, lam);
    free(z); free(l0); work_free(w);
    return (st & ~8) ? -it : it;        /* (bit 3: the QP hit qp_iters — not an error of the leaf call) */
}

int orc_debug_qp(const orc_model *mdl, const orc_config *c, const double *x0, const double *xf,
                 const double *xs, const double *us, double T, const double *lam, double *p, double *y) {
    return orc_debug_qp_multi(mdl, 1, c, x0, xf, xs, us, T, lam, p, y);
}

/* ------------------------------------------------------------------------------------------ */
typedef struct { const orc_model *m; const orc_config *c; int narm, B, t, nt; const double *x0, *xf, *xg, *ug, *Tg;
                 double *xs, *us, *T; orc_info *info; } job;
static void *job_run(void *a) {
    job *j = (job *)a; int N = orc_num_nodes(j->c->num_seg); const size_t nx = 14 * (size_t)j->narm, nu = 7 * (size_t)j->narm;
    for (int b = j->t; b < j->B; b += j->nt)
        orc_solve_multi(j->m, j->narm, j->c, j->x0 + nx * b, j->xf + nx * b, j->xg + nx * N * b, j->ug + nu * N * b, j->Tg[b],
                        j->xs + nx * N * b, j->us + nu * N * b, j->T + b, j->info ? j->info + b : 0);
    return 0;
}
void orc_solve_batch_multi(const orc_model *m, int narm, const orc_config *c, int B, const double *x0, const double *xf,
                           const double *xg, const double *ug, const double *Tg, double *xs, double *us, double *T,
                           orc_info *info, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; job jb[256];
    for (int t = 0; t < threads; t++) {
        job j = {m, c, narm, B, t, threads, x0, xf, xg, ug, Tg, xs, us, T, info};
        jb[t] = j;
        if (threads == 1) job_run(&jb[0]); else pthread_create(&th[t], 0, job_run, &jb[t]);
    }
    if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(th[t], 0);
}
void orc_solve_batch(const orc_model *m, const orc_config *c, int B, const double *x0, const double *xf,
                     const double *xg, const double *ug, const double *Tg, double *xs, double *us, double *T,
                     orc_info *info, int threads) {
    orc_solve_batch_multi(m, 1, c, B, x0, xf, xg, ug, Tg, xs, us, T, info, threads);
}

/* ------------------------------------------------------------------------------------------ */
/* Stand-in for the Ruckig warm start (motionPlanner.cpp:146-175): per-joint quintic with zero boundary
   accelerations (Ruckig is also called with zero current/target acceleration, motionPlanner.cpp:27-54),
   common duration = smallest T on a geometric grid for which |qd|<=vmax and |qdd|<=amax at 65 samples. */
static void quintic_coef(double q0, double v0, double q1, double v1, double T, double *c) {
    double h = q1 - q0, T2 = T * T, T3 = T2 * T;
    c[0] = q0; c[1] = v0; c[2] = 0.0;
    c[3] = (20.0 * h - (8.0 * v1 + 12.0 * v0) * T) / (2.0 * T3);
    c[4] = (-30.0 * h + (14.0 * v1 + 16.0 * v0) * T) / (2.0 * T3 * T);
    c[5] = (12.0 * h - 6.0 * (v1 + v0) * T) / (2.0 * T3 * T2);
}
static void quintic_eval(const double *c, double t, double *q, double *v, double *a) {
    *q = c[0] + t * (c[1] + t * (c[2] + t * (c[3] + t * (c[4] + t * c[5]))));
    *v = c[1] + t * (2 * c[2] + t * (3 * c[3] + t * (4 * c[4] + t * 5 * c[5])));
    *a = 2 * c[2] + t * (6 * c[3] + t * (12 * c[4] + t * 20 * c[5]));
}
void orc_warm_start(const orc_config *c, const double *amax_used, const double *x0, const double *xf,
                    double *xg, double *ug, double *Tg) {
    int N = orc_num_nodes(c->num_seg);
    double tau[MAXN]; orc_time_nodes(c->num_seg, tau);
    double T = 0.05, coef[7][6];
    for (int it = 0; it < 200; it++) {
        int ok = 1;
        for (int j = 0; j < 7 && ok; j++) {
            quintic_coef(x0[j], x0[7 + j], xf[j], xf[7 + j], T, coef[j]);
            for (int s = 0; s <= 64; s++) {
                double q, v, a; quintic_eval(coef[j], T * s / 64.0, &q, &v, &a);
                if (fabs(v) > c->ubx[7 + j] || fabs(a) > amax_used[j]) { ok = 0; break; }
            }
        }
        if (ok || T * 1.05 > c->ubT) break;
        T *= 1.05;
    }
    for (int j = 0; j < 7; j++) quintic_coef(x0[j], x0[7 + j], xf[j], xf[7 + j], T, coef[j]);
    for (int k = 0; k < N; k++) for (int j = 0; j < 7; j++) {
        double q, v, a; quintic_eval(coef[j], tau[k] * T, &q, &v, &a);
        xg[14 * k + j] = q; xg[14 * k + 7 + j] = v; ug[7 * k + j] = a;
    }
    for (int r = 0; r < 14; r++) { xg[r] = x0[r]; xg[14 * (N - 1) + r] = xf[r]; }
    *Tg = T;
}

/* ------------------------------------------------------------------------------------------ */
/* MPC<>::solution_x_at / solution_u_at + rnea  (motionPlanner.hpp:99-116): Lagrange interpolation on the
   segment that contains t */
void orc_sample(const orc_model *mdl, int num_seg, const double *xs, const double *us, double T, int n_pts, double *out) {
    static const double xi[4] = {-1.0, -0.5, 0.5, 1.0};
    for (int ip = 0; ip <= n_pts; ip++) {
        double t = (double)ip / n_pts;                 /* motionPlanner.hpp:103 */
        int s = (int)floor(t * num_seg); if (s >= num_seg) s = num_seg - 1; if (s < 0) s = 0;
        double x = 2.0 * (t * num_seg - s) - 1.0, L[4];
        for (int j = 0; j < 4; j++) {
            double v = 1.0;
            for (int k = 0; k < 4; k++) if (k != j) v *= (x - xi[k]) / (xi[j] - xi[k]);
            L[j] = v;
        }
        double q[7], v[7], a[7], tau[7];
        for (int r = 0; r < 7; r++) {
            q[r] = v[r] = a[r] = 0.0;
            for (int j = 0; j < 4; j++) {
                q[r] += L[j] * xs[14 * (3 * s + j) + r];
                v[r] += L[j] * xs[14 * (3 * s + j) + 7 + r];
                a[r] += L[j] * us[7 * (3 * s + j) + r];
            }
        }
        orc_rnea(mdl, q, v, a, tau);
        double *o = out + (size_t)ip * 29;
        o[0] = t * T;                                   /* motionPlanner.hpp:115 */
        memcpy(o + 1, q, sizeof q); memcpy(o + 8, v, sizeof v); memcpy(o + 15, a, sizeof a); memcpy(o + 22, tau, sizeof tau);
    }
}

/* MotionPlanner::get_MPC_point (motionPlanner.hpp:118-128): solution at physical time `time`, with the reference's
   clamp (time >= T -> normalised time := T, not 1).  out: q(7), v(7), a(7), tau(7). */
void orc_mpc_point(const orc_model *mdl, int num_seg, const double *xs, const double *us, double T, double time, double *out) {
    static const double xi[4] = {-1.0, -0.5, 0.5, 1.0};
    double t = (time < T) ? time / T : T;
    int s = (int)floor(t * num_seg); if (s >= num_seg) s = num_seg - 1; if (s < 0) s = 0;
    double x = 2.0 * (t * num_seg - s) - 1.0, L[4];
    for (int j = 0; j < 4; j++) {
        double v = 1.0;
        for (int k = 0; k < 4; k++) if (k != j) v *= (x - xi[k]) / (xi[j] - xi[k]);
        L[j] = v;
    }
    double *q = out, *v = out + 7, *a = out + 14;
    for (int r = 0; r < 7; r++) {
        q[r] = v[r] = a[r] = 0.0;
        for (int j = 0; j < 4; j++) {
            q[r] += L[j] * xs[14 * (3 * s + j) + r];
            v[r] += L[j] * xs[14 * (3 * s + j) + 7 + r];
            a[r] += L[j] * us[7 * (3 * s + j) + r];
        }
    }
    orc_rnea(mdl, q, v, a, out + 21);
}

/* One state advance of the receding-horizon loop with its arrival rule (include/mpcmp.h, mpcmp_rh_run; kernel k_advance): single arm, nx = 14.
   status = the status word of the solve that produced (xs, us, T).  x_io [14]: current state in / out.  Returns 1 when the instance is retired
   by this step (it is then no longer re-solved), 0 otherwise.
     - hard failure (NaN 1, lost positive definiteness 2, dead exchange 4) or T outside its box (32): the state is held;
     - T <= dt: the plan ends within the control period: the state becomes its last node, the instance is retired;
     - else the state advances by dt along the solution (get_MPC_point, motionPlanner.hpp:118-128; for dt < T its clamp never acts) and the
       instance is retired when the new state lies inside the terminal box |x - xf|_inf <= eps_target (motionPlanner.hpp:44). */
int orc_rh_advance(const orc_config *c, const double *xs, const double *us, double T, int status, double dt, const double *xf, double *x_io) {
    (void)us;
    const int num_seg = c->num_seg, N = 3 * num_seg + 1;
    static const double xi[4] = {-1.0, -0.5, 0.5, 1.0};
    if (status & (1 | 2 | 4 | 32)) return 0;
    if (T <= dt) { memcpy(x_io, xs + 14 * (N - 1), sizeof(double) * 14); return 1; }
    double t = dt / T;
    int s = (int)floor(t * num_seg); if (s >= num_seg) s = num_seg - 1; if (s < 0) s = 0;
    double x = 2.0 * (t * num_seg - s) - 1.0, L[4], far = 0.0;
    for (int j = 0; j < 4; j++) {
        double v = 1.0;
        for (int k = 0; k < 4; k++) if (k != j) v *= (x - xi[k]) / (xi[j] - xi[k]);
        L[j] = v;
    }
    for (int r = 0; r < 14; r++) {
        double acc = 0.0;
        for (int j = 0; j < 4; j++) acc += L[j] * xs[14 * (3 * s + j) + r];
        x_io[r] = acc;
        if (fabs(acc - xf[r]) > far) far = fabs(acc - xf[r]);
    }
    return far <= c->eps_target;
}

/* examples/benchmark.cpp:58-160 for one trajectory: out[74] = min(28) | max(28) | x(T)-target (14) | flags (4, 1 = pass) */
void orc_traj_stats(const orc_model *mdl, int num_seg, const double *xs, const double *us, double T, const double *xf,
                    int n_pts, double *out) {
    double *smp = (double *)malloc(sizeof(double) * 29 * (size_t)(n_pts + 1));
    double jerk[7];
    orc_default_limits(0, 0, 0, 0, jerk, 0);
    orc_sample(mdl, num_seg, xs, us, T, n_pts, smp);
    int fl[4] = {1, 1, 1, 1};
    const double dT = T / n_pts;
    for (int ip = 0; ip <= n_pts; ip++) {
        const double *r = smp + (size_t)ip * 29;
        if (ip >= 1) for (int j = 0; j < 7; j++)
            if (fabs((r[15 + j] - r[15 + j - 29]) / dT) > 10.0 * jerk[j]) fl[0] = 0;
        double J[42], vl[3] = {0, 0, 0}, va[3] = {0, 0, 0}, pt[3];
        orc_frame_jacobian(mdl, r + 1, mdl->tool, J);
        for (int d = 0; d < 3; d++) for (int j = 0; j < 7; j++) { vl[d] += J[d * 7 + j] * r[8 + j]; va[d] += J[(3 + d) * 7 + j] * r[8 + j]; }
        if (sqrt(vl[0] * vl[0] + vl[1] * vl[1] + vl[2] * vl[2]) > 1.7) fl[1] = 0;
        if (sqrt(va[0] * va[0] + va[1] * va[1] + va[2] * va[2]) > 2.5) fl[2] = 0;
        orc_fk(mdl, r + 1, 0, 0, 0, pt);
        if (pt[2] < 0.0) fl[3] = 0;
    }
    for (int c = 0; c < 28; c++) {
        double mn = smp[1 + c], mx = smp[1 + c];
        for (int ip = 1; ip <= n_pts; ip++) { double v = smp[(size_t)ip * 29 + 1 + c]; if (v < mn) mn = v; if (v > mx) mx = v; }
        out[c] = mn; out[28 + c] = mx;
    }
    for (int c = 0; c < 14; c++) out[56 + c] = smp[(size_t)n_pts * 29 + 1 + c] - xf[c];
    for (int c = 0; c < 4; c++) out[70 + c] = fl[c];
    free(smp);
}
