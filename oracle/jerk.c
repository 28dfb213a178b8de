/*
 * jerk.c — CPU restatement (TEST INFRASTRUCTURE ONLY) of the warm-start generator that stands in for Ruckig
 * (reference call sites: mpc_solver/motionPlanner.cpp:146-175 warm_start_RK, motionPlanner.hpp:73-96
 * get_ruckig_trajectory).  Ruckig itself (third-party, version unpinned, not installed) computes a jerk-limited,
 * time-optimal, time-synchronised trajectory between two states with zero boundary accelerations.  This file restates
 * that problem with the classical "double-S" construction:
 *
 *   per joint   : S-curve velocity transition v0 -> vc, cruise at vc, S-curve transition vc -> vf;
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

typedef struct { double v0, vc, vf, A, J, Ta, Tc, Td, T; int quintic; double c[6]; double p0; } jprof;

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
/* distance covered by the two transitions v0 -> vc -> vf (an S-curve covers its mean velocity times its duration) */
static double two_trans(double v0, double vc, double vf, double A, double J, double *Ta, double *Td) {
    double Tj, apk;
    trans_times(v0, vc, A, J, Ta, &Tj, &apk);
    trans_times(vc, vf, A, J, Td, &Tj, &apk);
    return 0.5 * (v0 + vc) * *Ta + 0.5 * (vc + vf) * *Td;
}
/* minimum-time profile of one joint */
static void prof_min(double dp, double v0, double vf, double V, double A, double J, jprof *o) {
    double Ta, Td, f;
    o->v0 = v0; o->vf = vf; o->A = A; o->J = J; o->quintic = 0;
    f = two_trans(v0, V, vf, A, J, &Ta, &Td);
    if (dp >= f) { o->vc = V; o->Ta = Ta; o->Td = Td; o->Tc = (dp - f) / V; o->T = Ta + Td + o->Tc; return; }
    f = two_trans(v0, -V, vf, A, J, &Ta, &Td);
    if (dp <= f) { o->vc = -V; o->Ta = Ta; o->Td = Td; o->Tc = (dp - f) / (-V); o->T = Ta + Td + o->Tc; return; }
    double lo = -V, hi = V;
    for (int it = 0; it < 100; it++) {
        const double mid = 0.5 * (lo + hi);
        if (two_trans(v0, mid, vf, A, J, &Ta, &Td) < dp) lo = mid; else hi = mid;
    }
    o->vc = 0.5 * (lo + hi);
    two_trans(v0, o->vc, vf, A, J, &o->Ta, &o->Td);
    o->Tc = 0.0; o->T = o->Ta + o->Td;
}
/* duration of the profile with cruise velocity vc, or -1 when that profile does not exist (negative cruise time) */
static double dur_of(double dp, double v0, double vf, double vc, double A, double J) {
    double Ta, Td;
    if (fabs(vc) < 1e-9) return -1.0;
    const double f = two_trans(v0, vc, vf, A, J, &Ta, &Td), Tc = (dp - f) / vc;
    return Tc < 0.0 ? -1.0 : Ta + Td + Tc;
}
/* profile of duration T (T > minimum time): 1 when found */
static int prof_sync(double dp, double v0, double vf, double V, double A, double J, double T, jprof *o) {
    double lam = 1.0;
    for (int li = 0; li < 60; li++, lam *= 0.85) {
        const double Al = lam * A, Jl = lam * J;
        double pv = 0.0, pd = 0.0;
        int have = 0;
        for (int i = 0; i <= 64; i++) {
            const double vc = -V + (2.0 * V) * i / 64.0;
            const double t = dur_of(dp, v0, vf, vc, Al, Jl);
            if (t < 0.0) { have = 0; continue; }
            const double dd = t - T;
            if (have && ((pd <= 0.0) != (dd <= 0.0)) && !(pv < 0.0 && vc > 0.0)) {
                double lo = pv, dlo = pd, hi = vc;
                int ok = 1;
                for (int it = 0; it < 80; it++) {
                    const double mid = 0.5 * (lo + hi), tm = dur_of(dp, v0, vf, mid, Al, Jl);
                    if (tm < 0.0) { ok = 0; break; }
                    if (((tm - T) <= 0.0) == (dlo <= 0.0)) { lo = mid; dlo = tm - T; } else hi = mid;
                }
                if (ok) {
                    o->v0 = v0; o->vf = vf; o->A = Al; o->J = Jl; o->quintic = 0; o->vc = 0.5 * (lo + hi);
                    const double f = two_trans(v0, o->vc, vf, Al, Jl, &o->Ta, &o->Td);
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
    if (t <= o->Ta) { trans_eval(o->v0, o->vc, o->A, o->J, t, &q, v, a); *p = o->p0 + q; return; }
    trans_eval(o->v0, o->vc, o->A, o->J, o->Ta, &pa, &va_, &aa);
    if (t <= o->Ta + o->Tc) { *p = o->p0 + pa + o->vc * (t - o->Ta); *v = o->vc; *a = 0.0; return; }
    trans_eval(o->vc, o->vf, o->A, o->J, t - o->Ta - o->Tc, &q, v, a);
    *p = o->p0 + pa + o->vc * o->Tc + q;
}
static void plan(const double *vmax, const double *amax, const double *jmax, const double *x0, const double *xf, jprof *pr, double *Tsync) {
    double T = 0.0;
    for (int j = 0; j < 7; j++) {
        prof_min(xf[j] - x0[j], x0[7 + j], xf[7 + j], vmax[j], amax[j], jmax[j], &pr[j]);
        pr[j].p0 = x0[j];
        if (pr[j].T > T) T = pr[j].T;
    }
    for (int j = 0; j < 7; j++) {
        if (pr[j].T >= T * (1.0 - 1e-12)) continue;
        jprof s;
        if (prof_sync(xf[j] - x0[j], x0[7 + j], xf[7 + j], vmax[j], amax[j], jmax[j], T, &s)) { s.p0 = x0[j]; pr[j] = s; continue; }
        /* fallback: quintic of the common duration (zero boundary accelerations) */
        const double h = xf[j] - x0[j], v0 = x0[7 + j], v1 = xf[7 + j], T2 = T * T, T3 = T2 * T;
        pr[j].quintic = 1;
        pr[j].c[0] = x0[j]; pr[j].c[1] = v0; pr[j].c[2] = 0.0;
        pr[j].c[3] = (20.0 * h - (8.0 * v1 + 12.0 * v0) * T) / (2.0 * T3);
        pr[j].c[4] = (-30.0 * h + (14.0 * v1 + 16.0 * v0) * T) / (2.0 * T3 * T);
        pr[j].c[5] = (12.0 * h - 6.0 * (v1 + v0) * T) / (2.0 * T3 * T2);
        pr[j].T = T;
    }
    *Tsync = T;
}

/* warm start for the OCP: states/controls at the collocation nodes scaled by the duration (motionPlanner.cpp:151-174) */
void orc_warm_start_jerk(int num_seg, const double *vmax, const double *amax, const double *jmax, const double *x0, const double *xf,
                         double *xg, double *ug, double *Tg) {
    jprof pr[7];
    double T, tau[64];
    const int N = orc_num_nodes(num_seg);
    plan(vmax, amax, jmax, x0, xf, pr, &T);
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

/* uniform samples of the same trajectory (get_ruckig_trajectory, motionPlanner.hpp:73-96): out (n_pts+1) x 22 = t, q, v, a */
void orc_jerk_trajectory(const double *vmax, const double *amax, const double *jmax, const double *x0, const double *xf, int n_pts,
                         double *out, double *T_out) {
    jprof pr[7];
    double T;
    plan(vmax, amax, jmax, x0, xf, pr, &T);
    for (int i = 0; i <= n_pts; i++) {
        const double t = T * i / n_pts;
        double *o = out + (long)i * 22;
        o[0] = t;
        for (int j = 0; j < 7; j++) prof_eval(&pr[j], t, &o[1 + j], &o[8 + j], &o[15 + j]);
    }
    if (T_out) *T_out = T;
}
