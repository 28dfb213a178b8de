// jerk_device.hpp — jerk-limited, time-synchronised state-to-state trajectories for the 7 joints of one problem: what the
// reference obtains from Ruckig as warm start and comparison trajectory (mpc_solver/motionPlanner.cpp:146-175 warm_start_RK,
// motionPlanner.hpp:73-96 get_ruckig_trajectory; boundary accelerations as set_current_state / set_target_state forward them,
// motionPlanner.cpp:36-38,50-52: zero in every example of the reference, and then exactly the zero-only arithmetic).
//
// Construction (the classical double-S profile): per joint an S-curve velocity transition (v0, a0) -> (vc, 0), a cruise at vc and an
// S-curve transition (vc, 0) -> (vf, aT) (a non-zero boundary acceleration: three-phase jerk profile from the given acceleration; the
// arrival is the time reverse of a departure (vf, -aT) -> (vc, 0)).  Minimum time = cruise at the velocity limit when the distance allows it, else the cruise-free
// profile whose two transitions cover the distance exactly (bisection on vc).  The common duration is that of the slowest
// joint; every other joint gets the profile of exactly that duration, found by scanning the cruise velocity (and, if needed,
// scaled-down acceleration/jerk limits) for a sign change of the duration error and bisecting.  On the reference's stored
// Ruckig trajectory this reproduces Ruckig's duration to 6 digits and all seven joint trajectories (tests/test_oracle_ocp.py
// for the CPU restatement, tests/test_gpu_parity.py for this code against it).
//
// One 64-thread workgroup (one wave) per problem: lanes 0..6 find the minimum-time profile of one joint each, the wave then
// re-plans the faster joints to the common duration one after the other (64 scan points at a time), then all lanes sample.
#pragma once
#include <hip/hip_runtime.h>
#include "rbd_device.hpp"

namespace mpcmp {

struct JerkLimits { double v[7], a[7], j[7]; };

struct JProf {
    double p0, v0, vc, vf, A, J, Ta, Tc, Td, T;
    double c[6];
    double a0, aT;
    int quintic;
};

__device__ __forceinline__ void jk_trans_times(double va, double vb, double A, double J, double &T, double &Tj, double &apk) {
    double dv = vb - va;
    const double s = dv >= 0 ? 1.0 : -1.0;
    dv = fabs(dv);
    if (dv >= A * A / J) { Tj = A / J; T = Tj + dv / A; apk = s * A; }
    else { Tj = sqrt(dv / J); T = 2.0 * Tj; apk = s * J * Tj; }
}
__device__ inline void jk_trans_eval(double va, double vb, double A, double J, double t, double &p, double &v, double &a) {
    double T, Tj, apk;
    jk_trans_times(va, vb, A, J, T, Tj, apk);
    if (!(T > 0.0)) { p = 0.0; v = va; a = 0.0; return; }
    if (t < 0.0) t = 0.0;
    if (t > T) t = T;
    const double j = apk >= 0 ? J : -J;
    if (t <= Tj) { a = j * t; v = va + 0.5 * j * t * t; p = va * t + j * t * t * t / 6.0; return; }
    const double p1 = va * Tj + j * Tj * Tj * Tj / 6.0, v1 = va + 0.5 * j * Tj * Tj, T2 = T - 2.0 * Tj;
    if (t <= Tj + T2) { const double u = t - Tj; a = apk; v = v1 + apk * u; p = p1 + v1 * u + 0.5 * apk * u * u; return; }
    const double p2 = p1 + v1 * T2 + 0.5 * apk * T2 * T2, v2 = v1 + apk * T2, u = t - Tj - T2;
    a = apk - j * u; v = v2 + apk * u - 0.5 * j * u * u; p = p2 + v2 * u + 0.5 * apk * u * u - j * u * u * u / 6.0;
}
// General departure: from (va, aa) to (vb, 0) in minimum time with |a| <= A, |j| <= J (oracle/jerk.c gtrans): jerk s J up to the peak
// acceleration (held for t2 if it reaches the limit), then jerk -s J down to zero.
struct JGtr { double s, ap, t1, t2, t3; };
__device__ __forceinline__ JGtr jk_gtrans(double va, double aa, double vb, double A, double J) {
    JGtr g;
    const double dv = vb - va, dv0 = aa * fabs(aa) / (2.0 * J);
    g.s = (dv - dv0) >= 0.0 ? 1.0 : -1.0;
    const double as = g.s * aa, dvs = g.s * dv;
    const double ap2 = J * dvs + 0.5 * as * as;
    double ap = ap2 > 0.0 ? sqrt(ap2) : 0.0;
    g.t2 = 0.0;
    if (A < fabs(aa)) A = fabs(aa);
    if (ap > A) { ap = A; g.t2 = (dvs - (2.0 * A * A - as * as) / (2.0 * J)) / A; if (g.t2 < 0.0) g.t2 = 0.0; }
    g.t1 = (ap - as) / J; if (g.t1 < 0.0) g.t1 = 0.0;
    g.t3 = ap / J;
    g.ap = g.s * ap;
    return g;
}
__device__ inline void jk_gtrans_eval(double va, double aa, const JGtr &g, double J, double t, double &p, double &v, double &a) {
    const double T = g.t1 + g.t2 + g.t3, j = g.s * J;
    if (t < 0.0) t = 0.0;
    if (t > T) t = T;
    if (t <= g.t1) { a = aa + j * t; v = va + aa * t + 0.5 * j * t * t; p = va * t + 0.5 * aa * t * t + j * t * t * t / 6.0; return; }
    const double t1 = g.t1, a1 = aa + j * t1, v1 = va + aa * t1 + 0.5 * j * t1 * t1, p1 = va * t1 + 0.5 * aa * t1 * t1 + j * t1 * t1 * t1 / 6.0;
    if (t <= t1 + g.t2) { const double u = t - t1; a = a1; v = v1 + a1 * u; p = p1 + v1 * u + 0.5 * a1 * u * u; return; }
    const double t2 = g.t2, v2 = v1 + a1 * t2, p2 = p1 + v1 * t2 + 0.5 * a1 * t2 * t2, u = t - t1 - t2;
    a = a1 - j * u; v = v2 + a1 * u - 0.5 * j * u * u; p = p2 + v2 * u + 0.5 * a1 * u * u - j * u * u * u / 6.0;
}
__device__ __forceinline__ double jk_gtrans_dist(double va, double aa, double vb, double A, double J, double &T) {
    const JGtr g = jk_gtrans(va, aa, vb, A, J);
    double p, v, a;
    T = g.t1 + g.t2 + g.t3;
    jk_gtrans_eval(va, aa, g, J, T, p, v, a);
    return p;
}
// distance of the two transitions v0 -> vc -> vf (an S-curve covers its mean velocity times its duration)
__device__ __forceinline__ double jk_two_trans(double v0, double a0, double vc, double vf, double aT, double A, double J, double &Ta, double &Td) {
    double Tj, apk, d;
    if (a0 == 0.0) { jk_trans_times(v0, vc, A, J, Ta, Tj, apk); d = 0.5 * (v0 + vc) * Ta; }
    else d = jk_gtrans_dist(v0, a0, vc, A, J, Ta);
    if (aT == 0.0) { jk_trans_times(vc, vf, A, J, Td, Tj, apk); d += 0.5 * (vc + vf) * Td; }
    else d += jk_gtrans_dist(vf, -aT, vc, A, J, Td);       // (the arrival is the time reverse of the departure (vf, -aT) -> (vc, 0): same distance)
    return d;
}
__device__ inline void jk_prof_min(double dp, double v0, double a0, double vf, double aT, double V, double A, double J, JProf &o) {
    double Ta, Td, f;
    o.v0 = v0; o.vf = vf; o.A = A; o.J = J; o.quintic = 0; o.a0 = a0; o.aT = aT;
    f = jk_two_trans(v0, a0, V, vf, aT, A, J, Ta, Td);
    if (dp >= f) { o.vc = V; o.Ta = Ta; o.Td = Td; o.Tc = (dp - f) / V; o.T = Ta + Td + o.Tc; return; }
    f = jk_two_trans(v0, a0, -V, vf, aT, A, J, Ta, Td);
    if (dp <= f) { o.vc = -V; o.Ta = Ta; o.Td = Td; o.Tc = (dp - f) / (-V); o.T = Ta + Td + o.Tc; return; }
    double lo = -V, hi = V;
    for (int it = 0; it < 100; it++) {
        const double mid = 0.5 * (lo + hi);
        if (jk_two_trans(v0, a0, mid, vf, aT, A, J, Ta, Td) < dp) lo = mid; else hi = mid;
    }
    o.vc = 0.5 * (lo + hi);
    jk_two_trans(v0, a0, o.vc, vf, aT, A, J, o.Ta, o.Td);
    o.Tc = 0.0; o.T = o.Ta + o.Td;
}
// duration of the profile with cruise velocity vc, or -1 when it does not exist (negative cruise time)
__device__ __forceinline__ double jk_dur_of(double dp, double v0, double a0, double vf, double aT, double vc, double A, double J) {
    double Ta, Td;
    if (fabs(vc) < 1e-9) return -1.0;
    const double f = jk_two_trans(v0, a0, vc, vf, aT, A, J, Ta, Td), Tc = (dp - f) / vc;
    return Tc < 0.0 ? -1.0 : Ta + Td + Tc;
}
// Profile of duration T (> the joint's minimum time), computed by the whole 64-lane wave: lane i tests the bracket between the
// grid points i and i+1 of the cruise-velocity scan, the first bracket (in scan order, as a serial scan would find it) is
// bisected.  Same result as the serial restatement in oracle/jerk.c.
__device__ inline bool jk_prof_sync(double dp, double v0, double a0, double vf, double aT, double V, double A, double J, double T, JProf &o) {
    const int lane = threadIdx.x & 63;
    double lam = 1.0;
    for (int li = 0; li < 60; li++, lam *= 0.85) {
        const double Al = lam * A, Jl = lam * J;
        const double va = -V + (2.0 * V) * lane / 64.0, vb = -V + (2.0 * V) * (lane + 1) / 64.0;
        const double ta = jk_dur_of(dp, v0, a0, vf, aT, va, Al, Jl), tb = jk_dur_of(dp, v0, a0, vf, aT, vb, Al, Jl);
        const bool cand = ta >= 0.0 && tb >= 0.0 && (((ta - T) <= 0.0) != ((tb - T) <= 0.0)) && !(va < 0.0 && vb > 0.0);
        unsigned long long mask = __ballot(cand);
        while (mask) {
            const int first = __ffsll((long long)mask) - 1;
            mask &= ~(1ull << first);
            double lo = __shfl(va, first), dlo = __shfl(ta, first) - T, hi = __shfl(vb, first);
            bool ok = true;
            for (int it = 0; it < 80; it++) {
                const double mid = 0.5 * (lo + hi), tm = jk_dur_of(dp, v0, a0, vf, aT, mid, Al, Jl);
                if (tm < 0.0) { ok = false; break; }
                if (((tm - T) <= 0.0) == (dlo <= 0.0)) { lo = mid; dlo = tm - T; } else hi = mid;
            }
            if (ok) {
                o.v0 = v0; o.vf = vf; o.A = Al; o.J = Jl; o.quintic = 0; o.a0 = a0; o.aT = aT; o.vc = 0.5 * (lo + hi);
                const double f = jk_two_trans(v0, a0, o.vc, vf, aT, Al, Jl, o.Ta, o.Td);
                o.Tc = (dp - f) / o.vc; o.T = o.Ta + o.Td + o.Tc;
                return true;
            }
        }
    }
    return false;
}
__device__ inline void jk_prof_eval(const JProf &o, double t, double &p, double &v, double &a) {
    if (o.quintic) {
        const double *c = o.c;
        p = c[0] + t * (c[1] + t * (c[2] + t * (c[3] + t * (c[4] + t * c[5]))));
        v = c[1] + t * (2 * c[2] + t * (3 * c[3] + t * (4 * c[4] + t * 5 * c[5])));
        a = 2 * c[2] + t * (6 * c[3] + t * (12 * c[4] + t * 20 * c[5]));
        return;
    }
    double pa, q, va_, aa;
    // departure (v0, a0) -> (vc, 0)
    if (o.a0 == 0.0) {
        if (t <= o.Ta) { jk_trans_eval(o.v0, o.vc, o.A, o.J, t, q, v, a); p = o.p0 + q; return; }
        jk_trans_eval(o.v0, o.vc, o.A, o.J, o.Ta, pa, va_, aa);
    } else {
        const JGtr g = jk_gtrans(o.v0, o.a0, o.vc, o.A, o.J);
        if (t <= o.Ta) { jk_gtrans_eval(o.v0, o.a0, g, o.J, t, q, v, a); p = o.p0 + q; return; }
        jk_gtrans_eval(o.v0, o.a0, g, o.J, o.Ta, pa, va_, aa);
    }
    if (t <= o.Ta + o.Tc) { p = o.p0 + pa + o.vc * (t - o.Ta); v = o.vc; a = 0.0; return; }
    // arrival (vc, 0) -> (vf, aT)
    if (o.aT == 0.0) { jk_trans_eval(o.vc, o.vf, o.A, o.J, t - o.Ta - o.Tc, q, v, a); }
    else {              // time reverse of the departure (vf, -aT) -> (vc, 0): p(t) = D - P(Td - t), v(t) = V(Td - t), a(t) = -A(Td - t)
        const JGtr g = jk_gtrans(o.vf, -o.aT, o.vc, o.A, o.J);
        double D, dv_, da_, pr_, ar_;
        jk_gtrans_eval(o.vf, -o.aT, g, o.J, o.Td, D, dv_, da_);
        double u = o.Td - (t - o.Ta - o.Tc);
        if (u < 0.0) u = 0.0;
        jk_gtrans_eval(o.vf, -o.aT, g, o.J, u, pr_, v, ar_);
        q = D - pr_; a = -ar_;
    }
    p = o.p0 + pa + o.vc * o.Tc + q;
}

// plans the seven joints of problem b into pr[7] (LDS) and returns the common duration; call with all 64 lanes
__device__ inline double jk_plan(const JerkLimits &lim, const double *x0, const double *xf, const double *acc0, const double *accT, JProf *pr, double *sT) {
    const int tid = threadIdx.x;
    if (tid < 7) {
        JProf o;
        jk_prof_min(xf[tid] - x0[tid], x0[7 + tid], acc0 ? acc0[tid] : 0.0, xf[7 + tid], accT ? accT[tid] : 0.0, lim.v[tid], lim.a[tid], lim.j[tid], o);
        o.p0 = x0[tid];
        pr[tid] = o;
    }
    __syncthreads();
    if (tid == 0) { double T = 0.0; for (int j = 0; j < 7; j++) T = pr[j].T > T ? pr[j].T : T; *sT = T; }
    __syncthreads();
    const double T = *sT;
    for (int j = 0; j < 7; j++) {                      // (workgroup-uniform: the profiles live in LDS)
        if (!(pr[j].T < T * (1.0 - 1e-12))) continue;
        JProf s;
        const double a0 = acc0 ? acc0[j] : 0.0, aT = accT ? accT[j] : 0.0;
        const bool found = jk_prof_sync(xf[j] - x0[j], x0[7 + j], a0, xf[7 + j], aT, lim.v[j], lim.a[j], lim.j[j], T, s);
        __syncthreads();
        if (tid == 0) {
            if (found) { s.p0 = x0[j]; pr[j] = s; }
            else {      // fallback: quintic of the common duration through both boundary states
                const double h = xf[j] - x0[j], v0 = x0[7 + j], v1 = xf[7 + j], T2 = T * T, T3 = T2 * T;
                JProf &q = pr[j];
                q.quintic = 1;
                q.c[0] = x0[j]; q.c[1] = v0; q.c[2] = 0.5 * a0;
                q.c[3] = (20.0 * h - (8.0 * v1 + 12.0 * v0) * T - (3.0 * a0 - aT) * T2) / (2.0 * T3);
                q.c[4] = (-30.0 * h + (14.0 * v1 + 16.0 * v0) * T + (3.0 * a0 - 2.0 * aT) * T2) / (2.0 * T3 * T);
                q.c[5] = (12.0 * h - 6.0 * (v1 + v0) * T - (a0 - aT) * T2) / (2.0 * T3 * T2);
                q.T = T;
            }
        }
    }
    __syncthreads();
    return T;
}

// warm start of the OCP: node states [N][14], node controls [N][7], duration (motionPlanner.cpp:151-174, 202-203)
// (need_status / need_mask / need_div: the receding-horizon driver computes the guess only for the instances that will use it — those whose previous solve is
//  no guess: status[b / need_div] & need_mask — every step but the first; null = every problem)
__global__ __launch_bounds__(64) void k_warm_jerk(int nseg, JerkLimits lim, const double *x0, const double *xf, const double *acc0, const double *accT, double *wx, double *wu, double *wT,
                                                  const int *need_status = nullptr, int need_mask = 0, int need_div = 1) {
    __shared__ JProf pr[7];
    __shared__ double sT;
    const int b = blockIdx.x, tid = threadIdx.x, N = 3 * nseg + 1;
    if (need_status && !(need_status[b / need_div] & need_mask)) return;
    const double *a0 = x0 + 14 * (size_t)b, *af = xf + 14 * (size_t)b;
    const double T = jk_plan(lim, a0, af, acc0 ? acc0 + 7 * (size_t)b : nullptr, accT ? accT + 7 * (size_t)b : nullptr, pr, &sT);
    for (int t = tid; t < N * 7; t += 64) {
        const int k = t / 7, j = t % 7, s = k / 3, i = k % 3;
        const double xi = (i == 0) ? -1.0 : (i == 1 ? -0.5 : 0.5);                    // ascending cubic CGL nodes of segment s
        const double tau = (k == N - 1) ? 1.0 : (s + 0.5 * (xi + 1.0)) / nseg;
        double q, v, a;
        jk_prof_eval(pr[j], tau * T, q, v, a);
        if (k == 0) { q = a0[j]; v = a0[7 + j]; }
        if (k == N - 1) { q = af[j]; v = af[7 + j]; }
        wx[((size_t)b * N + k) * 14 + j] = q; wx[((size_t)b * N + k) * 14 + 7 + j] = v; wu[((size_t)b * N + k) * 7 + j] = a;
    }
    if (tid == 0) wT[b] = T;
}

// uniform samples [n_pts+1][22] = t, q(7), v(7), a(7) of the same trajectory (get_ruckig_trajectory, motionPlanner.hpp:73-96)
__global__ __launch_bounds__(64) void k_jerk_traj(JerkLimits lim, const double *x0, const double *xf, const double *acc0, const double *accT, int n_pts, double *out, double *Tout) {
    __shared__ JProf pr[7];
    __shared__ double sT;
    const int b = blockIdx.x, tid = threadIdx.x;
    const double T = jk_plan(lim, x0 + 14 * (size_t)b, xf + 14 * (size_t)b, acc0 ? acc0 + 7 * (size_t)b : nullptr, accT ? accT + 7 * (size_t)b : nullptr, pr, &sT);
    double *o = out + (size_t)b * (n_pts + 1) * 22;
    for (int t = tid; t < (n_pts + 1) * 7; t += 64) {
        const int i = t / 7, j = t % 7;
        const double tt = T * i / n_pts;
        double q, v, a;
        jk_prof_eval(pr[j], tt, q, v, a);
        o[(size_t)i * 22 + 1 + j] = q; o[(size_t)i * 22 + 8 + j] = v; o[(size_t)i * 22 + 15 + j] = a;
        if (j == 0) o[(size_t)i * 22] = tt;
    }
    if (tid == 0 && Tout) Tout[b] = T;
}

// MotionPlanner::get_RK_point (motionPlanner.hpp:130-142): the same trajectory at one physical time per problem, clamped to its
// duration (`time = std::min(time, trajectory.get_duration())`), plus the RNEA torque.  out [B][28] = q(7), v(7), a(7), tau(7).
__global__ __launch_bounds__(64) void k_jerk_point(const mpcmp_model *mdl, JerkLimits lim, const double *x0, const double *xf, const double *acc0, const double *accT,
                                                   const double *time, double *out, double *Tout) {
    __shared__ JProf pr[7];
    __shared__ double sT;
    __shared__ double pt[21];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double T = jk_plan(lim, x0 + 14 * (size_t)b, xf + 14 * (size_t)b, acc0 ? acc0 + 7 * (size_t)b : nullptr, accT ? accT + 7 * (size_t)b : nullptr, pr, &sT);
    const double tt = time[b] < T ? time[b] : T;
    if (tid < 7) {
        double q, v, a;
        jk_prof_eval(pr[tid], tt, q, v, a);
        pt[tid] = q; pt[7 + tid] = v; pt[14 + tid] = a;
    }
    __syncthreads();
    if (tid == 0) {
        double sc[14], v[7], a[7], tau[7];
#pragma unroll
        for (int j = 0; j < 7; j++) { sincos(pt[j], &sc[2 * j], &sc[2 * j + 1]); v[j] = pt[7 + j]; a[j] = pt[14 + j]; }
        rnea_dir<false>(mdl, sc, v, a, 0, 0, tau, nullptr);
        double *o = out + (size_t)b * 28;
#pragma unroll
        for (int j = 0; j < 7; j++) { o[j] = pt[j]; o[7 + j] = v[j]; o[14 + j] = a[j]; o[21 + j] = tau[j]; }
        if (Tout) Tout[b] = T;
    }
}

}  // namespace mpcmp
