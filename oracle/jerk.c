/*
 * jerk.c — CPU restatement (TEST INFRASTRUCTURE ONLY) of the warm-start generator that stands in for Ruckig
 * (reference call sites: mpc_solver/motionPlanner.cpp:146-175 warm_start_RK, motionPlanner.hpp:73-96
 * get_ruckig_trajectory).  Ruckig itself (third-party, version unpinned, not installed) computes a jerk-limited,
 * time-optimal, time-synchronised trajectory between two states.  The reference forwards the boundary accelerations of
 * set_current_state / set_target_state to it (motionPlanner.cpp:36-38,50-52), zero in every example.  This file restates
 * that problem with the classical "double-S" construction:
 *
 *   per joint   : S-curve velocity transition (v0, a0) -> (vc, 0), cruise at vc, S-curve transition (vc, 0) -> (vf, aT)
 *                 (non-zero boundary accelerations: three-phase jerk profile +-J, 0, -+J from the given acceleration; the
 *                 arrival transition is the time reverse of a departure (vf, -aT) -> (vc, 0));
 *                 minimum time = largest feasible |vc| (cruise at the velocity limit if the distance allows it,
 *                 otherwise the cruise-free profile whose two transitions cover the distance exactly);
 *   all joints  : common duration = slowest joint; every other joint gets the profile of that duration found by scanning
 *                 the cruise velocity (and, if needed, scaled-down acceleration/jerk limits) for a sign change and
 *                 bisecting; a joint for which no profile is found falls back to a quintic of the common duration.
 *
 * Pinned by the one stored Ruckig trajectory of the reference (tests/golden/gold_traj.json, KAT-RK): duration to 6 s.f.,
 * positions of all seven joints to 1.2e-5 rad at the 201 stored samples.
 */
#include <math.h>
#include "oracle.h"

typedef struct { double v0, vc, vf, A, J, Ta, Tc, Td, T; int quintic; double c[6]; double p0; double a0, aT; } jprof;

/* S-curve velocity transition va -> vb with |a| <= A, |j| <= J and zero acceleration at both ends */
static void trans_times(double va, double vb, double A, double J, double *T, double *Tj, double *apk) {
    double dv = vb - va, s = dv >= 0 ? 1.0 : -1.0;
    dv = fabs(dv);
    if (dv >= A * A / J) { *Tj = A / J; *T = *Tj + dv / A; *apk = s * A; }
    else { *Tj = sqrt(dv / J); *T = 2.0 * *Tj; *apk = s * J * *Tj; }
}
static void trans_eval(double va, double vb, double A, double J, double t, double *p, double *v, double *a) {
    double T, Tj, apk;
    trans_times(va, vb, A, J, &T, &Tj, &apk);
    if (!(T > 0.0)) { *p = 0.0; *v = va; *a = 0.0; return; }
    if (t < 0.0) t = 0.0;
    if (t > T) t = T;
    const double j = apk >= 0 ? J : -J;
    if (t <= Tj) { *a = j * t; *v = va + 0.5 * j * t * t; *p = va * t + j * t * t * t / 6.0; return; }
    const double p1 = va * Tj + j * Tj * Tj * Tj / 6.0, v1 = va + 0.5 * j * Tj * Tj, T2 = T - 2.0 * Tj;
    if (t <= Tj + T2) { const double u = t - Tj; *a = apk; *v = v1 + apk * u; *p = p1 + v1 * u + 0.5 * apk * u * u; return; }
    const double p2 = p1 + v1 * T2 + 0.5 * apk * T2 * T2, v2 = v1 + apk * T2, u = t - Tj - T2;
    *a = apk - j * u; *v = v2 + apk * u - 0.5 * j * u * u; *p = p2 + v2 * u + 0.5 * apk * u * u - j * u * u * u / 6.0;
}
/* General departure: from (va, aa) to (vb, 0) in minimum time with |a| <= A, |j| <= J: jerk s J up to the peak acceleration ap (held for t2
 * if it reaches the limit), then jerk -s J down to zero; s = sign of the velocity change still needed once aa has been ramped to zero. */
typedef struct { double s, ap, t1, t2, t3; } gtr;
static gtr gtrans(double va, double aa, double vb, double A, double J) {
    gtr g;
    const double dv = vb - va, dv0 = aa * fabs(aa) / (2.0 * J);
    g.s = (dv - dv0) >= 0.0 ? 1.0 : -1.0;
    const double as = g.s * aa, dvs = g.s * dv;
    double ap2 = J * dvs + 0.5 * as * as, ap = ap2 > 0.0 ? sqrt(ap2) : 0.0;
    g.t2 = 0.0;
    if (A < fabs(aa)) A = fabs(aa);       /* (a given boundary acceleration above a scaled-down limit of the synchronisation scan: the limit yields) */
    if (ap > A) { ap = A; g.t2 = (dvs - (2.0 * A * A - as * as) / (2.0 * J)) / A; if (g.t2 < 0.0) g.t2 = 0.0; }
    g.t1 = (ap - as) / J; if (g.t1 < 0.0) g.t1 = 0.0;
    g.t3 = ap / J;
    g.ap = g.s * ap;
    return g;
}
static void gtrans_eval(double va, double aa, const gtr *g, double J, double t, double *p, double *v, double *a) {
    const double T = g->t1 + g->t2 + g->t3, j = g->s * J;
    if (t < 0.0) t = 0.0;
    if (t > T) t = T;
    if (t <= g->t1) { *a = aa + j * t; *v = va + aa * t + 0.5 * j * t * t; *p = va * t + 0.5 * aa * t * t + j * t * t * t / 6.0; return; }
    const double t1 = g->t1, a1 = aa + j * t1, v1 = va + aa * t1 + 0.5 * j * t1 * t1, p1 = va * t1 + 0.5 * aa * t1 * t1 + j * t1 * t1 * t1 / 6.0;
    if (t <= t1 + g->t2) { const double u = t - t1; *a = a1; *v = v1 + a1 * u; *p = p1 + v1 * u + 0.5 * a1 * u * u; return; }
    const double t2 = g->t2, v2 = v1 + a1 * t2, p2 = p1 + v1 * t2 + 0.5 * a1 * t2 * t2, u = t - t1 - t2;
    *a = a1 - j * u; *v = v2 + a1 * u - 0.5 * j * u * u; *p = p2 + v2 * u + 0.5 * a1 * u * u - j * u * u * u / 6.0;
}
/* duration and distance of the departure (va, aa) -> (vb, 0) */
static double gtrans_dist(double va, double aa, double vb, double A, double J, double *T) {
    const gtr g = gtrans(va, aa, vb, A, J);
    double p, v, a;
    *T = g.t1 + g.t2 + g.t3;
    gtrans_eval(va, aa, &g, J, *T, &p, &v, &a);
    return p;
}
/* distance covered by the two transitions v0 -> vc -> vf (an S-curve covers its mean velocity times its duration) */
static double two_trans(double v0, double a0, double vc, double vf, double aT, double A, double J, double *Ta, double *Td) {
    double Tj, apk, d;
    if (a0 == 0.0) { trans_times(v0, vc, A, J, Ta, &Tj, &apk); d = 0.5 * (v0 + vc) * *Ta; }
    else d = gtrans_dist(v0, a0, vc, A, J, Ta);
    if (aT == 0.0) { trans_times(vc, vf, A, J, Td, &Tj, &apk); d += 0.5 * (vc + vf) * *Td; }
    else d += gtrans_dist(vf, -aT, vc, A, J, Td);          /* the arrival is the time reverse of the departure (vf, -aT) -> (vc, 0): same distance */
    return d;
}
/* minimum-time profile of one joint */
static void prof_min(double dp, double v0, double a0, double vf, double aT, double V, double A, double J, jprof *o) {
    double Ta, Td, f;
    o->v0 = v0; o->vf = vf; o->A = A; o->J = J; o->quintic = 0; o->a0 = a0; o->aT = aT;
    f = two_trans(v0, a0, V, vf, aT, A, J, &Ta, &Td);
    if (dp >= f) { o->vc = V; o->Ta = Ta; o->Td = Td; o->Tc = (dp - f) / V; o->T = Ta + Td + o->Tc; return; }
    f = two_trans(v0, a0, -V, vf, aT, A, J, &Ta, &Td);
    if (dp <= f) { o->vc = -V; o->Ta = Ta; o->Td = Td; o->Tc = (dp - f) / (-V); o->T = Ta + Td + o->Tc; return; }
    double lo = -V, hi = V;
    for (int it = 0; it < 100; it++) {
        const double mid = 0.5 * (lo + hi);
        if (two_trans(v0, a0, mid, vf, aT, A, J, &Ta, &Td) < dp) lo = mid; else hi = mid;
    }
    o->vc = 0.5 * (lo + hi);
    two_trans(v0, a0, o->vc, vf, aT, A, J, &o->Ta, &o->Td);
    o->Tc = 0.0; o->T = o->Ta + o->Td;
}
/* duration of the profile with cruise velocity vc, or -1 when that profile does not exist (negative cruise time) */
static double dur_of(double dp, double v0, double a0, double vf, double aT, double vc, double A, double J) {
    double Ta, Td;
    if (fabs(vc) < 1e-9) return -1.0;
    const double f = two_trans(v0, a0, vc, vf, aT, A, J, &Ta, &Td), Tc = (dp - f) / vc;
    return Tc < 0.0 ? -1.0 : Ta + Td + Tc;
}
/* profile of duration T (T > minimum time): 1 when found */
static int prof_sync(double dp, double v0, double a0, double vf, double aT, double V, double A, double J, double T, jprof *o) {
    double lam = 1.0;
    for (int li = 0; li < 60; li++, lam *= 0.85) {
        const double Al = lam * A, Jl = lam * J;
        double pv = 0.0, pd = 0.0;
        int have = 0;
        for (int i = 0; i <= 64; i++) {
            const double vc = -V + (2.0 * V) * i / 64.0;
            const double t = dur_of(dp, v0, a0, vf, aT, vc, Al, Jl);
            if (t < 0.0) { have = 0; continue; }
            const double dd = t - T;
            if (have && ((pd <= 0.0) != (dd <= 0.0)) && !(pv < 0.0 && vc > 0.0)) {
                double lo = pv, dlo = pd, hi = vc;
                int ok = 1;
                for (int it = 0; it < 80; it++) {
                    const double mid = 0.5 * (lo + hi), tm = dur_of(dp, v0, a0, vf, aT, mid, Al, Jl);
                    if (tm < 0.0) { ok = 0; break; }
                    if (((tm - T) <= 0.0) == (dlo <= 0.0)) { lo = mid; dlo = tm - T; } else hi = mid;
                }
                if (ok) {
                    o->v0 = v0; o->vf = vf; o->A = Al; o->J = Jl; o->quintic = 0; o->a0 = a0; o->aT = aT; o->vc = 0.5 * (lo + hi);
                    const double f = two_trans(v0, a0, o->vc, vf, aT, Al, Jl, &o->Ta, &o->Td);
                    o->Tc = (dp - f) / o->vc; o->T = o->Ta + o->Td + o->Tc;
                    return 1;
                }
            }
            pv = vc; pd = dd; have = 1;
        }
    }
    return 0;
}
static void prof_eval(const jprof *o, double t, double *p, double *v, double *a) {
    if (o->quintic) {
        const double *c = o->c;
        *p = c[0] + t * (c[1] + t * (c[2] + t * (c[3] + t * (c[4] + t * c[5]))));
        *v = c[1] + t * (2 * c[2] + t * (3 * c[3] + t * (4 * c[4] + t * 5 * c[5])));
        *a = 2 * c[2] + t * (6 * c[3] + t * (12 * c[4] + t * 20 * c[5]));
        return;
    }
    double pa, q, va_, aa;
    /* departure (v0, a0) -> (vc, 0) */
    if (o->a0 == 0.0) {
        if (t <= o->Ta) { trans_eval(o->v0, o->vc, o->A, o->J, t, &q, v, a); *p = o->p0 + q; return; }
        trans_eval(o->v0, o->vc, o->A, o->J, o->Ta, &pa, &va_, &aa);
    } else {
        const gtr g = gtrans(o->v0, o->a0, o->vc, o->A, o->J);
        if (t <= o->Ta) { gtrans_eval(o->v0, o->a0, &g, o->J, t, &q, v, a); *p = o->p0 + q; return; }
        gtrans_eval(o->v0, o->a0, &g, o->J, o->Ta, &pa, &va_, &aa);
    }
    if (t <= o->Ta + o->Tc) { *p = o->p0 + pa + o->vc * (t - o->Ta); *v = o->vc; *a = 0.0; return; }
    /* arrival (vc, 0) -> (vf, aT) */
    if (o->aT == 0.0) { trans_eval(o->vc, o->vf, o->A, o->J, t - o->Ta - o->Tc, &q, v, a); }
    else {              /* time reverse of the departure (vf, -aT) -> (vc, 0): p(t) = D - P(Td - t), v(t) = V(Td - t), a(t) = -A(Td - t) */
        const gtr g = gtrans(o->vf, -o->aT, o->vc, o->A, o->J);
        double D, dv_, da_, pr_, ar_;
        gtrans_eval(o->vf, -o->aT, &g, o->J, o->Td, &D, &dv_, &da_);
        double u = o->Td - (t - o->Ta - o->Tc);
        if (u < 0.0) u = 0.0;
        gtrans_eval(o->vf, -o->aT, &g, o->J, u, &pr_, v, &ar_);
        q = D - pr_; *a = -ar_;
    }
    *p = o->p0 + pa + o->vc * o->Tc + q;
}
static void plan(const double *vmax, const double *amax, const double *jmax, const double *x0, const double *xf, const double *acc0, const double *accT,
                 jprof *pr, double *Tsync) {
    double T = 0.0;
    for (int j = 0; j < 7; j++) {
        const double a0 = acc0 ? acc0[j] : 0.0, aT = accT ? accT[j] : 0.0;
        prof_min(xf[j] - x0[j], x0[7 + j], a0, xf[7 + j], aT, vmax[j], amax[j], jmax[j], &pr[j]);
        pr[j].p0 = x0[j];
        if (pr[j].T > T) T = pr[j].T;
    }
    for (int j = 0; j < 7; j++) {
        if (pr[j].T >= T * (1.0 - 1e-12)) continue;
        const double a0 = acc0 ? acc0[j] : 0.0, aT = accT ? accT[j] : 0.0;
        jprof s;
        if (prof_sync(xf[j] - x0[j], x0[7 + j], a0, xf[7 + j], aT, vmax[j], amax[j], jmax[j], T, &s)) { s.p0 = x0[j]; pr[j] = s; continue; }
        /* fallback: quintic of the common duration through both boundary states */
        const double h = xf[j] - x0[j], v0 = x0[7 + j], v1 = xf[7 + j], T2 = T * T, T3 = T2 * T;
        pr[j].quintic = 1;
        pr[j].c[0] = x0[j]; pr[j].c[1] = v0; pr[j].c[2] = 0.5 * a0;
        pr[j].c[3] = (20.0 * h - (8.0 * v1 + 12.0 * v0) * T - (3.0 * a0 - aT) * T2) / (2.0 * T3);
        pr[j].c[4] = (-30.0 * h + (14.0 * v1 + 16.0 * v0) * T + (3.0 * a0 - 2.0 * aT) * T2) / (2.0 * T3 * T);
        pr[j].c[5] = (12.0 * h - 6.0 * (v1 + v0) * T - (a0 - aT) * T2) / (2.0 * T3 * T2);
        pr[j].T = T;
    }
    *Tsync = T;
}

/* warm start for the OCP: states/controls at the collocation nodes scaled by the duration (motionPlanner.cpp:151-174); acc0 / accT: boundary
 * accelerations [7] (NULL = zero: the arithmetic of the zero case is then exactly that of the zero-only restatement pinned by KAT-RK) */
void orc_warm_start_jerk_acc(int num_seg, const double *vmax, const double *amax, const double *jmax, const double *x0, const double *xf,
                             const double *acc0, const double *accT, double *xg, double *ug, double *Tg) {
    jprof pr[7];
    double T, tau[64];
    const int N = orc_num_nodes(num_seg);
    plan(vmax, amax, jmax, x0, xf, acc0, accT, pr, &T);
    orc_time_nodes(num_seg, tau);
    for (int k = 0; k < N; k++)
        for (int j = 0; j < 7; j++) {
            double q, v, a;
            prof_eval(&pr[j], tau[k] * T, &q, &v, &a);
            xg[14 * k + j] = q; xg[14 * k + 7 + j] = v; ug[7 * k + j] = a;
        }
    for (int r = 0; r < 14; r++) { xg[r] = x0[r]; xg[14 * (N - 1) + r] = xf[r]; }   /* motionPlanner.cpp:202-203 */
    *Tg = T;
}
void orc_warm_start_jerk(int num_seg, const double *vmax, const double *amax, const double *jmax, const double *x0, const double *xf,
                         double *xg, double *ug, double *Tg) {
    orc_warm_start_jerk_acc(num_seg, vmax, amax, jmax, x0, xf, 0, 0, xg, ug, Tg);
}

/* uniform samples of the same trajectory (get_ruckig_trajectory, motionPlanner.hpp:73-96): out (n_pts+1) x 22 = t, q, v, a */
void orc_jerk_trajectory_acc(const double *vmax, const double *amax, const double *jmax, const double *x0, const double *xf,
                             const double *acc0, const double *accT, int n_pts, double *out, double *T_out) {
    jprof pr[7];
    double T;
    plan(vmax, amax, jmax, x0, xf, acc0, accT, pr, &T);
    for (int i = 0; i <= n_pts; i++) {
        const double t = T * i / n_pts;
        double *o = out + (long)i * 22;
        o[0] = t;
        for (int j = 0; j < 7; j++) prof_eval(&pr[j], t, &o[1 + j], &o[8 + j], &o[15 + j]);
    }
    if (T_out) *T_out = T;
}
void orc_jerk_trajectory(const double *vmax, const double *amax, const double *jmax, const double *x0, const double *xf, int n_pts,
                         double *out, double *T_out) {
    orc_jerk_trajectory_acc(vmax, amax, jmax, x0, xf, 0, 0, n_pts, out, T_out);
}
