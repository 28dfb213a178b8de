/*
 * rbd.c — rigid-body layer of the CPU oracle (TEST INFRASTRUCTURE, see oracle.h).
 *
 * Restates the Pinocchio calls made on the reference's hot path for the fixed 7-joint
 * revolute-z chain of robot_utils/panda-model/panda_arm.urdf:
 *   pinocchio::rnea                    robot_ocp.hpp:91,120   motionPlanner.hpp:92,111,127,141
 *   pinocchio::computeRNEADerivatives  robot_ocp.hpp:118
 *   pinocchio::crba (+ symmetrise)     robot_ocp.hpp:121-122
 *   forwardKinematics / updateFramePlacement / computeFrameJacobian (+ rotation to
 *   world-aligned)                     robot_ocp.hpp:87-88,145-155
 * Pinocchio itself is not in /root/reference; the published recursive Newton-Euler algorithm
 * is restated in 3-vector (Luh-Walker-Paul) form and PINNED by tests/golden/kat_*.
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

/* ---------- small vector helpers ---------- */
static inline void cross(const double *a, const double *b, double *c) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    c[0] = x; c[1] = y; c[2] = z;
}
static inline void matvec(const double *R, const double *v, double *o) { /* o = R v */
    double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    double y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    double z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void matTvec(const double *R, const double *v, double *o) { /* o = R^T v */
    double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
    double y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
    double z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void zcross(const double *v, double s, double *o) { /* o = s * (z_hat x v) */
    double x = -s * v[1], y = s * v[0];
    o[0] = x; o[1] = y; o[2] = 0.0;
}
static inline void add3(double *a, const double *b) { a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; }
static inline void sub3(double *a, const double *b) { a[0] -= b[0]; a[1] -= b[1]; a[2] -= b[2]; }

/* ---------- model: numbers restated from panda_arm.urdf ---------- */
void orc_default_model(orc_model *m) {
    /* joint origins: rpy roll only (urdf :17,34,51,68,85,102,119) */
    static const double roll[7] = {0.0, -1.57079632679, 1.57079632679, 1.57079632679,
                                   -1.57079632679, 1.57079632679, 1.57079632679};
    static const double xyz[7][3] = {{0, 0, 0.333}, {0, 0, 0}, {0, -0.316, 0}, {0.0825, 0, 0},
                                     {-0.0825, 0.384, 0}, {0, 0, 0}, {0.088, 0, 0}};
    /* inertials (urdf :9-13, 26-30, 43-47, 60-64, 77-81, 94-98, 111-115) */
    static const double mass[7] = {4.970684, 0.646926, 3.228604, 3.587895, 1.225946, 1.666555, 7.35522e-01};
    static const double com[7][3] = {{3.875e-03, 2.081e-03, -0.1750}, {-3.141e-03, -2.872e-02, 3.495e-03},
                                     {2.7518e-02, 3.9252e-02, -6.6502e-02}, {-5.317e-02, 1.04419e-01, 2.7454e-02},
                                     {-1.1953e-02, 4.1065e-02, -3.8437e-02}, {6.0149e-02, -1.4117e-02, -1.0517e-02},
                                     {1.0517e-02, -4.252e-03, 6.1597e-02}};
    /* ixx ixy ixz iyy iyz izz */
    static const double in6[7][6] = {{7.0337e-01, -1.3900e-04, 6.7720e-03, 7.0661e-01, 1.9169e-02, 9.1170e-03},
                                     {7.9620e-03, -3.9250e-03, 1.0254e-02, 2.8110e-02, 7.0400e-04, 2.5995e-02},
                                     {3.7242e-02, -4.7610e-03, -1.1396e-02, 3.6155e-02, -1.2805e-02, 1.0830e-02},
                                     {2.5853e-02, 7.7960e-03, -1.3320e-03, 1.9552e-02, 8.6410e-03, 2.8323e-02},
                                     {3.5549e-02, -2.1170e-03, -4.0370e-03, 2.9474e-02, 2.2900e-04, 8.6270e-03},
                                     {1.9640e-03, 1.0900e-04, -1.1580e-03, 4.3540e-03, 3.4100e-04, 5.4330e-03},
                                     {1.2516e-02, -4.2800e-04, -1.1960e-03, 1.0027e-02, -7.4100e-04, 4.8150e-03}};
    memset(m, 0, sizeof(*m));
    for (int i = 0; i < 7; i++) {
        double c = cos(roll[i]), s = sin(roll[i]);
        double R[9] = {1, 0, 0, 0, c, -s, 0, s, c}; /* Rx(roll) */
        memcpy(m->R0[i], R, sizeof R);
        memcpy(m->p[i], xyz[i], sizeof xyz[i]);
        m->mass[i] = mass[i];
        memcpy(m->com[i], com[i], sizeof com[i]);
        const double *I = in6[i];
        double F[9] = {I[0], I[1], I[2], I[1], I[3], I[4], I[2], I[4], I[5]};
        memcpy(m->I[i], F, sizeof F);
    }
    m->link8[0] = 0; m->link8[1] = 0; m->link8[2] = 0.107;        /* urdf :135 */
    m->tool[0] = 0; m->tool[1] = 0; m->tool[2] = 0.107 + 0.15;      /* urdf :135,149 */
    m->gravity[0] = 0; m->gravity[1] = 0; m->gravity[2] = -9.81;
    /* lump the two fixed children into body 7 (what Pinocchio's URDF parser does):
       link8: m=0, I=1e-3*Id at link8 origin (urdf :127-133); tool: m=1, I=1e-3*Id at tool origin (:141-147) */
    {
        const double mk[3] = {mass[6], 0.0, 1.0};
        const double ck[3][3] = {{com[6][0], com[6][1], com[6][2]}, {0, 0, 0.107}, {0, 0, 0.257}};
        double Ik[3][9];
        memcpy(Ik[0], m->I[6], sizeof Ik[0]);
        for (int k = 1; k < 3; k++) { memset(Ik[k], 0, sizeof Ik[k]); Ik[k][0] = Ik[k][4] = Ik[k][8] = 0.001; }
        double M = mk[0] + mk[1] + mk[2], c[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) for (int d = 0; d < 3; d++) c[d] += mk[k] * ck[k][d];
        for (int d = 0; d < 3; d++) c[d] /= M;
        double It[9] = {0};
        for (int k = 0; k < 3; k++) {
            double d[3] = {ck[k][0] - c[0], ck[k][1] - c[1], ck[k][2] - c[2]};
            double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            for (int r = 0; r < 3; r++) for (int s = 0; s < 3; s++)
                It[3 * r + s] += Ik[k][3 * r + s] + mk[k] * ((r == s ? d2 : 0.0) - d[r] * d[s]);
        }
        m->mass[6] = M; memcpy(m->com[6], c, sizeof c); memcpy(m->I[6], It, sizeof It);
    }
}

void orc_default_limits(double *qmin, double *qmax, double *vmax, double *amax, double *jmax, double *taumax) {
    /* robot_utils/pandaWrapper.hpp:29-34 */
    static const double a[6][7] = {{-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973},
                                   {2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973},
                                   {2.1750, 2.1750, 2.1750, 2.1750, 2.6100, 2.6100, 2.6100},
                                   {15.0, 7.5, 10.0, 12.5, 15.0, 20.0, 20.0},
                                   {7500, 3750, 5000, 6250, 7500, 10000, 10000},
                                   {87, 87, 87, 87, 12, 12, 12}};
    double *o[6] = {qmin, qmax, vmax, amax, jmax, taumax};
    for (int k = 0; k < 6; k++) if (o[k]) memcpy(o[k], a[k], sizeof a[k]);
}

/* joint rotation R_i(q) = R0_i * Rz(q) */
static void joint_rot(const orc_model *m, int i, double q, double *R) {
    double c = cos(q), s = sin(q);
    const double *A = m->R0[i];
    for (int r = 0; r < 3; r++) {
        R[3 * r + 0] = A[3 * r + 0] * c + A[3 * r + 1] * s;
        R[3 * r + 1] = -A[3 * r + 0] * s + A[3 * r + 1] * c;
        R[3 * r + 2] = A[3 * r + 2];
    }
}

typedef struct {
    double R[ORC_NJ][9];
    double w[ORC_NJ][3], wd[ORC_NJ][3], al[ORC_NJ][3]; /* omega, omega_dot, linear acc of frame origin */
    double u[ORC_NJ][3];   /* R^T omega_parent */
    double ud[ORC_NJ][3];  /* R^T omegadot_parent */
    double b[ORC_NJ][3];   /* parent-frame acceleration of joint origin */
    double F[ORC_NJ][3], N[ORC_NJ][3];
    double f[ORC_NJ][3], n[ORC_NJ][3];
    double Iw[ORC_NJ][3];
    double wxc[ORC_NJ][3]; /* omega x com */
} rnea_cache;

static void rnea_primal(const orc_model *m, const double *q, const double *v, const double *a,
                        rnea_cache *C, double *tau) {
    const int n = ORC_NJ;
    double w0[3] = {0, 0, 0}, wd0[3] = {0, 0, 0}, a0[3] = {-m->gravity[0], -m->gravity[1], -m->gravity[2]};
    for (int i = 0; i < n; i++) {
        const double *wp = i ? C->w[i - 1] : w0, *wdp = i ? C->wd[i - 1] : wd0, *ap = i ? C->al[i - 1] : a0;
        joint_rot(m, i, q[i], C->R[i]);
        matTvec(C->R[i], wp, C->u[i]);
        matTvec(C->R[i], wdp, C->ud[i]);
        /* omega */
        C->w[i][0] = C->u[i][0]; C->w[i][1] = C->u[i][1]; C->w[i][2] = C->u[i][2] + v[i];
        /* omega_dot = ud + a z + u x (v z) */
        C->wd[i][0] = C->ud[i][0] + C->u[i][1] * v[i];
        C->wd[i][1] = C->ud[i][1] - C->u[i][0] * v[i];
        C->wd[i][2] = C->ud[i][2] + a[i];
        /* b = a_p + wd_p x p + w_p x (w_p x p) */
        double t1[3], t2[3];
        cross(wdp, m->p[i], t1);
        cross(wp, m->p[i], t2); cross(wp, t2, t2);
        C->b[i][0] = ap[0] + t1[0] + t2[0]; C->b[i][1] = ap[1] + t1[1] + t2[1]; C->b[i][2] = ap[2] + t1[2] + t2[2];
        matTvec(C->R[i], C->b[i], C->al[i]);
        /* com acceleration */
        double ac[3];
        cross(C->wd[i], m->com[i], t1);
        cross(C->w[i], m->com[i], C->wxc[i]); cross(C->w[i], C->wxc[i], t2);
        ac[0] = C->al[i][0] + t1[0] + t2[0]; ac[1] = C->al[i][1] + t1[1] + t2[1]; ac[2] = C->al[i][2] + t1[2] + t2[2];
        for (int d = 0; d < 3; d++) C->F[i][d] = m->mass[i] * ac[d];
        matvec(m->I[i], C->w[i], C->Iw[i]);
        matvec(m->I[i], C->wd[i], t1);
        cross(C->w[i], C->Iw[i], t2);
        for (int d = 0; d < 3; d++) C->N[i][d] = t1[d] + t2[d];
    }
    for (int i = n - 1; i >= 0; i--) {
        double t[3];
        for (int d = 0; d < 3; d++) C->f[i][d] = C->F[i][d];
        cross(m->com[i], C->F[i], t);
        for (int d = 0; d < 3; d++) C->n[i][d] = C->N[i][d] + t[d];
        if (i + 1 < n) {
            double gf[3], gn[3];
            matvec(C->R[i + 1], C->f[i + 1], gf);
            matvec(C->R[i + 1], C->n[i + 1], gn);
            cross(m->p[i + 1], gf, t);
            add3(C->f[i], gf); add3(C->n[i], gn); add3(C->n[i], t);
        }
        tau[i] = C->n[i][2];
    }
}

void orc_rnea(const orc_model *m, const double *q, const double *v, const double *a, double *tau) {
    rnea_cache C;
    rnea_primal(m, q, v, a, &C, tau);
}

/* directional derivative of tau along (dq,dv,da) given the primal cache */
static void rnea_jvp(const orc_model *m, const double *v, const rnea_cache *C,
                     const double *dq, const double *dv, const double *da, double *dtau) {
    const int n = ORC_NJ;
    double dw[ORC_NJ][3], dwd[ORC_NJ][3], dal[ORC_NJ][3], dF[ORC_NJ][3], dN[ORC_NJ][3];
    double z3[3] = {0, 0, 0};
    double w0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
        const double *dwp = i ? dw[i - 1] : z3, *dwdp = i ? dwd[i - 1] : z3, *dap = i ? dal[i - 1] : z3;
        const double *wp = i ? C->w[i - 1] : w0;
        double du[3], dud[3], t1[3], t2[3], t3[3];
        /* du = R^T dw_p - dq z x u */
        matTvec(C->R[i], dwp, du); zcross(C->u[i], dq[i], t1); sub3(du, t1);
        matTvec(C->R[i], dwdp, dud); zcross(C->ud[i], dq[i], t1); sub3(dud, t1);
        dw[i][0] = du[0]; dw[i][1] = du[1]; dw[i][2] = du[2] + dv[i];
        /* wd = ud + a z + u x (v z):  u x (v z) = ( u1 v, -u0 v, 0 ) */
        dwd[i][0] = dud[0] + du[1] * v[i] + C->u[i][1] * dv[i];
        dwd[i][1] = dud[1] - du[0] * v[i] - C->u[i][0] * dv[i];
        dwd[i][2] = dud[2] + da[i];
        /* db = da_p + dwd_p x p + dw_p x (w_p x p) + w_p x (dw_p x p) */
        double db[3];
        cross(dwdp, m->p[i], t1);
        cross(wp, m->p[i], t2); cross(dwp, t2, t2);
        cross(dwp, m->p[i], t3); cross(wp, t3, t3);
        for (int d = 0; d < 3; d++) db[d] = dap[d] + t1[d] + t2[d] + t3[d];
        /* dal = R^T db - dq z x al */
        matTvec(C->R[i], db, dal[i]); zcross(C->al[i], dq[i], t1); sub3(dal[i], t1);
        /* dac = dal + dwd x c + dw x (w x c) + w x (dw x c) */
        double dac[3];
        cross(dwd[i], m->com[i], t1);
        cross(dw[i], C->wxc[i], t2);
        cross(dw[i], m->com[i], t3); cross(C->w[i], t3, t3);
        for (int d = 0; d < 3; d++) dac[d] = dal[i][d] + t1[d] + t2[d] + t3[d];
        for (int d = 0; d < 3; d++) dF[i][d] = m->mass[i] * dac[d];
        /* dN = I dwd + dw x (I w) + w x (I dw) */
        double Idw[3];
        matvec(m->I[i], dwd[i], t1);
        cross(dw[i], C->Iw[i], t2);
        matvec(m->I[i], dw[i], Idw); cross(C->w[i], Idw, t3);
        for (int d = 0; d < 3; d++) dN[i][d] = t1[d] + t2[d] + t3[d];
    }
    double df[ORC_NJ][3], dn[ORC_NJ][3];
    for (int i = n - 1; i >= 0; i--) {
        double t[3];
        for (int d = 0; d < 3; d++) df[i][d] = dF[i][d];
        cross(m->com[i], dF[i], t);
        for (int d = 0; d < 3; d++) dn[i][d] = dN[i][d] + t[d];
        if (i + 1 < n) {
            /* g = R_{i+1} f_{i+1};  dg = R (df + dq z x f) */
            double hf[3], hn[3], gf[3], gn[3];
            zcross(C->f[i + 1], dq[i + 1], hf); add3(hf, df[i + 1]);
            zcross(C->n[i + 1], dq[i + 1], hn); add3(hn, dn[i + 1]);
            matvec(C->R[i + 1], hf, gf);
            matvec(C->R[i + 1], hn, gn);
            cross(m->p[i + 1], gf, t);
            add3(df[i], gf); add3(dn[i], gn); add3(dn[i], t);
        }
        dtau[i] = dn[i][2];
    }
}

void orc_rnea_derivatives(const orc_model *m, const double *q, const double *v, const double *a,
                          double *tau, double *dtau_dq, double *dtau_dv, double *M) {
    rnea_cache C;
    rnea_primal(m, q, v, a, &C, tau);
    for (int j = 0; j < ORC_NJ; j++) {
        double e[ORC_NJ] = {0}, z[ORC_NJ] = {0}, col[ORC_NJ];
        e[j] = 1.0;
        rnea_jvp(m, v, &C, e, z, z, col);
        for (int i = 0; i < ORC_NJ; i++) dtau_dq[i * ORC_NJ + j] = col[i];
        rnea_jvp(m, v, &C, z, e, z, col);
        for (int i = 0; i < ORC_NJ; i++) dtau_dv[i * ORC_NJ + j] = col[i];
        rnea_jvp(m, v, &C, z, z, e, col);
        for (int i = 0; i < ORC_NJ; i++) M[i * ORC_NJ + j] = col[i];
    }
}

static void fk_all(const orc_model *m, const double *q, double Rw[ORC_NJ][9], double pw[ORC_NJ][3]);

/* ---------- analytic partial derivatives of RNEA (pinocchio::computeRNEADerivatives, robot_ocp.hpp:118) and the mass matrix (pinocchio::crba,
   robot_ocp.hpp:121-122) in closed form: 6-D spatial algebra in the WORLD frame, as the published algorithm is formulated (Carpentier & Mansard,
   "Analytical derivatives of rigid body dynamics algorithms", RSS 2018; the identities below are those of Singh, Russell & Wensing, "Efficient
   analytical derivatives of rigid-body dynamics using spatial vector algebra", RA-L 2022).  An independent second implementation: the solver's
   linearisation uses the directional derivatives above (rnea_jvp); tests/test_oracle_rigid_body.py checks that the two agree to round-off.

   Spatial vectors at the world origin: motion [omega; v_O], force [n_O; f].  Joint i: S_i = [z_i; p_i x z_i].  With v_k = sum_{l<=k} S_l qd_l,
   a_k = a_0 + sum_{l<=k} (S_l qdd_l + v_l x S_l qd_l), f_k = I_k a_k + v_k x* I_k v_k, tau_i = S_i^T sum_{k>=i} f_k (serial chain):
     d v_k / d qd_j = S_j                          d a_k / d qd_j = (2 v_j - v_k) x S_j                                   (j <= k)
     d S_l / d q_j  = S_j x S_l  (j < l)           d I_k / d q_j  = S_j x* I_k - I_k S_j x                                 (j <= k)
     d v_k / d q_j  = S_j x (v_k - v_j)            d a_k / d q_j  = sum_{j<l<=k} [(S_j x S_l) qdd_l + (S_j x (v_l - v_j)) x S_l qd_l + v_l x (S_j x S_l) qd_l] */
typedef struct { double w[3], v[3]; } sv6;   /* motion: (omega, v_O);  force: (n_O, f) */
static inline sv6 sv_zero(void) { sv6 r = {{0, 0, 0}, {0, 0, 0}}; return r; }
static inline sv6 sv_add(sv6 a, sv6 b) { for (int d = 0; d < 3; d++) { a.w[d] += b.w[d]; a.v[d] += b.v[d]; } return a; }
static inline sv6 sv_sub(sv6 a, sv6 b) { for (int d = 0; d < 3; d++) { a.w[d] -= b.w[d]; a.v[d] -= b.v[d]; } return a; }
static inline sv6 sv_scale(sv6 a, double s) { for (int d = 0; d < 3; d++) { a.w[d] *= s; a.v[d] *= s; } return a; }
static inline double sv_dot(sv6 m, sv6 f) { double r = 0; for (int d = 0; d < 3; d++) r += m.w[d] * f.w[d] + m.v[d] * f.v[d]; return r; }
static inline sv6 crm(sv6 a, sv6 m) {        /* a x m (motion) = [w x m_w; w x m_v + v x m_w] */
    sv6 r; double t[3];
    cross(a.w, m.w, r.w); cross(a.w, m.v, r.v); cross(a.v, m.w, t); add3(r.v, t); return r;
}
static inline sv6 crf(sv6 a, sv6 f) {        /* a x* f (force) = [w x n + v x f; w x f] */
    sv6 r; double t[3];
    cross(a.w, f.w, r.w); cross(a.v, f.v, t); add3(r.w, t); cross(a.w, f.v, r.v); return r;
}
typedef struct { double m, c[3], Ic[9]; } si6;   /* spatial inertia: mass, com (world, from the origin), rotational inertia about the com (world axes) */
static inline sv6 si_mul(const si6 *I, sv6 a) {  /* I a: f = m (v + w x c);  n_O = Ic w + c x f */
    sv6 r; double t[3];
    cross(a.w, I->c, t);
    for (int d = 0; d < 3; d++) r.v[d] = I->m * (a.v[d] + t[d]);
    matvec(I->Ic, a.w, r.w); cross(I->c, r.v, t); add3(r.w, t);
    return r;
}

void orc_rnea_derivatives_analytic(const orc_model *m, const double *q, const double *v, const double *a,
                                   double *tau, double *dtau_dq, double *dtau_dv, double *M) {
    const int n = ORC_NJ;
    double Rw[ORC_NJ][9], pw[ORC_NJ][3];
    fk_all(m, q, Rw, pw);
    sv6 S[ORC_NJ], vel[ORC_NJ], acc[ORC_NJ], f[ORC_NJ], fC[ORC_NJ];
    si6 I[ORC_NJ];
    sv6 a0 = sv_zero();
    for (int d = 0; d < 3; d++) a0.v[d] = -m->gravity[d];
    for (int i = 0; i < n; i++) {
        for (int d = 0; d < 3; d++) S[i].w[d] = Rw[i][3 * d + 2];
        cross(pw[i], S[i].w, S[i].v);
        /* inertia of body i in world axes, com from the world origin */
        double t[3], RI[9];
        matvec(Rw[i], m->com[i], t);
        for (int d = 0; d < 3; d++) I[i].c[d] = pw[i][d] + t[d];
        I[i].m = m->mass[i];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
            RI[3 * r + c] = Rw[i][3 * r + 0] * m->I[i][0 + c] + Rw[i][3 * r + 1] * m->I[i][3 + c] + Rw[i][3 * r + 2] * m->I[i][6 + c];
        }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
            I[i].Ic[3 * r + c] = RI[3 * r + 0] * Rw[i][3 * c + 0] + RI[3 * r + 1] * Rw[i][3 * c + 1] + RI[3 * r + 2] * Rw[i][3 * c + 2];
        const sv6 vp = i ? vel[i - 1] : sv_zero(), ap = i ? acc[i - 1] : a0;
        vel[i] = sv_add(vp, sv_scale(S[i], v[i]));
        acc[i] = sv_add(ap, sv_add(sv_scale(S[i], a[i]), sv_scale(crm(vel[i], S[i]), v[i])));
        f[i] = sv_add(si_mul(&I[i], acc[i]), crf(vel[i], si_mul(&I[i], vel[i])));
    }
    for (int i = n - 1; i >= 0; i--) {
        fC[i] = f[i];
        if (i + 1 < n) fC[i] = sv_add(fC[i], fC[i + 1]);
        tau[i] = sv_dot(S[i], fC[i]);
    }
    /* d f_k / d qd_j and d f_k / d q_j for j <= k */
    for (int j = 0; j < n; j++) {
        sv6 dfv_sum[ORC_NJ], dfq_sum[ORC_NJ];      /* suffix sums over k >= i of the partials of f_k */
        sv6 dfv[ORC_NJ], dfq[ORC_NJ];
        sv6 daq = sv_zero();                      /* running d a_k / d q_j */
        for (int k = 0; k < n; k++) { dfv[k] = sv_zero(); dfq[k] = sv_zero(); }
        for (int k = j; k < n; k++) {
            const si6 *Ik = &I[k];
            /* velocity partials */
            const sv6 dv_v = S[j];
            const sv6 da_v = crm(sv_sub(sv_scale(vel[j], 2.0), vel[k]), S[j]);
            dfv[k] = sv_add(si_mul(Ik, da_v), sv_add(crf(dv_v, si_mul(Ik, vel[k])), crf(vel[k], si_mul(Ik, dv_v))));
            /* configuration partials */
            if (k > j) {
                const sv6 dS = crm(S[j], S[k]);
                const sv6 dvl = crm(S[j], sv_sub(vel[k], vel[j]));
                daq = sv_add(daq, sv_add(sv_scale(dS, a[k]), sv_add(sv_scale(crm(dvl, S[k]), v[k]), sv_scale(crm(vel[k], dS), v[k]))));
            }
            const sv6 dv_q = crm(S[j], sv_sub(vel[k], vel[j]));
            /* (dI) x = S_j x* (I x) - I (S_j x x) */
            const sv6 dI_a = sv_sub(crf(S[j], si_mul(Ik, acc[k])), si_mul(Ik, crm(S[j], acc[k])));
            const sv6 dI_v = sv_sub(crf(S[j], si_mul(Ik, vel[k])), si_mul(Ik, crm(S[j], vel[k])));
            dfq[k] = sv_add(dI_a, sv_add(si_mul(Ik, daq), sv_add(crf(dv_q, si_mul(Ik, vel[k])), sv_add(crf(vel[k], dI_v), crf(vel[k], si_mul(Ik, dv_q))))));
        }
        for (int i = n - 1; i >= 0; i--) {
            dfv_sum[i] = dfv[i]; dfq_sum[i] = dfq[i];
            if (i + 1 < n) { dfv_sum[i] = sv_add(dfv_sum[i], dfv_sum[i + 1]); dfq_sum[i] = sv_add(dfq_sum[i], dfq_sum[i + 1]); }
        }
        for (int i = 0; i < n; i++) {
            dtau_dv[i * n + j] = sv_dot(S[i], dfv_sum[i]);
            double dq_ij = sv_dot(S[i], dfq_sum[i]);
            if (j < i) dq_ij += sv_dot(crm(S[j], S[i]), fC[i]);      /* (d S_i / d q_j)^T f_i^C */
            dtau_dq[i * n + j] = dq_ij;
        }
    }
    /* composite rigid body algorithm: M_ij = S_i^T I_i^C S_j (j <= i), I_i^C = sum_{k >= i} I_k */
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            sv6 h = sv_zero();
            for (int k = i; k < n; k++) h = sv_add(h, si_mul(&I[k], S[j]));
            M[i * n + j] = M[j * n + i] = sv_dot(S[i], h);
        }
}

/* world placement of every joint frame */
static void fk_all(const orc_model *m, const double *q, double Rw[ORC_NJ][9], double pw[ORC_NJ][3]) {
    double Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pp[3] = {0, 0, 0};
    for (int i = 0; i < ORC_NJ; i++) {
        double R[9], t[3];
        joint_rot(m, i, q[i], R);
        matvec(Rp, m->p[i], t);
        for (int d = 0; d < 3; d++) pw[i][d] = pp[d] + t[d];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
            Rw[i][3 * r + c] = Rp[3 * r + 0] * R[0 + c] + Rp[3 * r + 1] * R[3 + c] + Rp[3 * r + 2] * R[6 + c];
        memcpy(Rp, Rw[i], sizeof Rp); memcpy(pp, pw[i], sizeof pp);
    }
}

void orc_fk(const orc_model *m, const double *q, double *p7, double *R7, double *p_link8, double *p_tool) {
    double Rw[ORC_NJ][9], pw[ORC_NJ][3], t[3];
    fk_all(m, q, Rw, pw);
    if (p7) memcpy(p7, pw[6], 3 * sizeof(double));
    if (R7) memcpy(R7, Rw[6], 9 * sizeof(double));
    if (p_link8) { matvec(Rw[6], m->link8, t); for (int d = 0; d < 3; d++) p_link8[d] = pw[6][d] + t[d]; }
    if (p_tool) { matvec(Rw[6], m->tool, t); for (int d = 0; d < 3; d++) p_tool[d] = pw[6][d] + t[d]; }
}

void orc_frame_jacobian(const orc_model *m, const double *q, const double *off, double *J) {
    double Rw[ORC_NJ][9], pw[ORC_NJ][3], t[3], pe[3];
    fk_all(m, q, Rw, pw);
    matvec(Rw[6], off, t);
    for (int d = 0; d < 3; d++) pe[d] = pw[6][d] + t[d];
    for (int j = 0; j < ORC_NJ; j++) {
        double z[3] = {Rw[j][2], Rw[j][5], Rw[j][8]}; /* joint axis in world */
        double r[3] = {pe[0] - pw[j][0], pe[1] - pw[j][1], pe[2] - pw[j][2]}, lin[3];
        cross(z, r, lin);
        for (int d = 0; d < 3; d++) { J[d * ORC_NJ + j] = lin[d]; J[(3 + d) * ORC_NJ + j] = z[d]; }
    }
}

/* robot_ocp.hpp:80-96 (values) and :98-163 (values + Jacobian rows) */
void orc_eval_constraints(const orc_model *m, int quirk, const double *x, const double *u,
                          double *g, double *G) {
    const double *q = x, *v = x + 7;
    double ptool[3];
    if (!G) {
        orc_rnea(m, q, v, u, g);
        orc_fk(m, q, 0, 0, 0, ptool);
        g[7] = ptool[2];
        return;
    }
    double dq[49], dv[49], M[49];
    orc_rnea_derivatives(m, q, v, u, g, dq, dv, M);
    memset(G, 0, sizeof(double) * ORC_NG * 22);
    for (int i = 0; i < 7; i++) {
        double *row = G + i * 22;
        double dT = 0.0;
        for (int j = 0; j < 7; j++) {
            row[j] = dq[i * 7 + j];
            row[7 + j] = dv[i * 7 + j];
            /* data.M after crba + upper->lower copy (robot_ocp.hpp:121-122): exactly symmetric */
            row[14 + j] = (i <= j) ? M[i * 7 + j] : M[j * 7 + i];
            /* robot_ocp.hpp:124: dtau_dv*qd + dtau_da*qdd with dtau_da upper-triangular only */
            dT += dv[i * 7 + j] * v[j];
            if (j >= i) dT += M[i * 7 + j] * u[j];
        }
        row[21] = quirk ? dT : 0.0;
    }
    double J[42];
    orc_frame_jacobian(m, q, m->tool, J);
    orc_fk(m, q, 0, 0, 0, ptool);
    g[7] = ptool[2];
    for (int j = 0; j < 7; j++) G[7 * 22 + j] = J[2 * 7 + j];
}
