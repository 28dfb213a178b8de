// multi_kernels.hpp — k_init_m / k_step_m: warm start + first linearisation, and line search + update + re-linearisation
// (+ final report), for OCPs with NARM independent 7-joint arms that share only the final time T (BASELINE.json configs[3]:
// 14-DoF dual Panda = robot_ocp.hpp:31-213 with doubled sizes) and for N = 25.  Same arithmetic as k_init / k_step
// (solver_kernels.hpp); written with strided loops (512 threads, any n) and the robot models read through a pointer.
//
// Device layout of an OCP (arm-major): z = [arm 0: xs [N][14] | us [N][7]] [arm 1: ...] [T];  general rows / multipliers
// [arm 0: dynamics 14(N-1) | path 8N] [arm 1: ...], then the box rows in the order of z.  The C ABI keeps the reference's
// layout with doubled sizes: x = [q(7 NARM); qd(7 NARM)] per node, u = qdd(7 NARM) per node.
#pragma once
#include "qp_kernel_v3.hpp"

namespace mpcmp {

template <int NSEG, int NARM>
struct DimM {
    using A = Dim3<NSEG>;
    static constexpr int N = A::N, na = A::na, meq = A::meq, ma = A::ma;
    static constexpr int n = NARM * na + 1, m = NARM * ma, mn = m + n;
    static constexpr int NT = 512, NW = NT / 64;
    // LDS (doubles): z | p | per-arm scratch (sin/cos N*14, raw N*147, trial sin/cos 9*N*14, trial violations 9*N) | x0/xf per arm | red
    static constexpr int oZ = 0, oP = oZ + n + 1, oScr = oP + n + 1;
    static constexpr int scr = N * 14 + N * 147;
    static constexpr int oSct = oScr + scr, oPv = oSct + 9 * N * 14, oX0 = oPv + 9 * N, oRed = oX0 + 2 * 14 * NARM, size = oRed + NW * 12 + 8;
};

// index of component r in [0,14) of arm a in an external state vector [q(7 NARM); qd(7 NARM)]
template <int NARM>
__device__ __forceinline__ int ext_x(int a, int r) { return r < 7 ? 7 * a + r : 7 * NARM + 7 * a + (r - 7); }

// variable box of arm variable v (external arm order) — motionPlanner.cpp:33,47,66-79
template <int NSEG>
__device__ __forceinline__ void arm_box(const mpcmp_config &c, const double *x0a, const double *xfa, int v, double &lo, double &hi) {
    constexpr int N = 3 * NSEG + 1;
    if (v < 14 * N) {
        const int k = v / 14, r = v % 14;
        if (k == 0) { lo = hi = x0a[r]; }
        else if (k == N - 1) { lo = xfa[r] - c.eps_target; hi = xfa[r] + c.eps_target; }
        else { lo = c.lbx[r]; hi = c.ubx[r]; }
    } else {
        const int r = (v - 14 * N) % 7;
        lo = c.lbu[r]; hi = c.ubu[r];
    }
}

// Linearisation of one arm's nodes: zl = the arm's block of the iterate (LDS), T the final time.  Writes g [8N], Gk [N][8][22],
// ceq [14(N-1)] (global).  scr: N*14 + N*147 doubles of LDS.
template <int NSEG, int NT>
__device__ __forceinline__ void linearise_arm(const mpcmp_config &cfg, const mpcmp_model *__restrict__ mdl_, const double *zl, double T,
                                              double *scr, double *g_out, double *Gk_out, double *ceq_out, int tid) {
    constexpr int N = 3 * NSEG + 1;
    double *sc = scr, *raw = scr + N * 14;
    for (int t = tid; t < N * 7; t += NT) {
        double s, c;
        sincos(zl[14 * (t / 7) + (t % 7)], &s, &c);
        sc[2 * t] = s; sc[2 * t + 1] = c;
    }
    __syncthreads();
    // Item order: the 14 N directions in (q, v) — the generic tangent recursion —, then the 7 N directions in a (mass-matrix columns: rnea_mcol, a
    // quarter of the instructions), then the N tool rows.  N = 25 has 550 items for 512 lanes: the 38 of the second pass (short ones) go to the LAST
    // lanes, whose first-pass items are short as well; in node-major order wave 0 ran two generic streams while seven waves waited.
    constexpr int LH = 14 * N, LF0 = 21 * N, LTOT = 22 * N;
#pragma nounroll
    for (int u0 = 0; u0 < LTOT; u0 += NT) {
        const int t = u0 == 0 ? tid : u0 + (NT - 1 - tid);
        if (t >= LTOT) continue;
        // (an opaque copy of the model pointer per pass: the ~180 model constants of the recursion are otherwise hoisted out of the loop
        //  as invariants and spilled - 1.7 KB of scratch per lane; with it they are scalar loads inside the pass)
        const mpcmp_model *mdl = mdl_;
        asm volatile("" : "+s"(mdl));
        const int k = t < LH ? t / 14 : (t < LF0 ? (t - LH) / 7 : t - LF0);
        const int d = t < LH ? t % 14 : (t < LF0 ? 14 + (t - LH) % 7 : 21);
        const double *q_sc = sc + 14 * k;
        const double *v = zl + 14 * k + 7, *a = zl + 14 * N + 7 * k;
        if (d < 14) {
            double tau[7], dtau[7];
            rnea_dir<true, false>(mdl, q_sc, v, a, d / 7, d % 7, tau, dtau);
#pragma unroll
            for (int i = 0; i < 7; i++) raw[(k * 7 + i) * 21 + d] = dtau[i];
            if (d == 0) {
#pragma unroll
                for (int i = 0; i < 7; i++) g_out[8 * k + i] = tau[i];
            }
        } else if (d < 21) {
            double dtau[7];
            rnea_mcol<false, false>(mdl, q_sc, d - 14, dtau);
#pragma unroll
            for (int i = 0; i < 7; i++) raw[(k * 7 + i) * 21 + d] = dtau[i];
        } else {
            V3 pt; double Jz[7];
            fk_tool(mdl, q_sc, &pt, Jz, nullptr, nullptr);
            g_out[8 * k + 7] = pt.z;
            double *row = Gk_out + (k * 8 + 7) * 22;
#pragma unroll
            for (int c = 0; c < 7; c++) row[c] = Jz[c];
#pragma unroll
            for (int c = 7; c < 22; c++) row[c] = 0.0;
        }
    }
    __syncthreads();
    // rows 0..6 of every node: [dtau/dq | dtau/dqd | M symmetrised | quirk column]  (robot_ocp.hpp:129-142)
    // (the quirk column — a 14-term sum — has a pass of its own: inside the pass below one lane in 22 ran its loop and the other 21 waited, in every
    //  round of the pass: 9.5 k of k_step<4>'s 93 k cycles)
    for (int t = tid; t < N * 7 * 22; t += NT) {
        const int k = t / 154, i = (t % 154) / 22, c = t % 22;
        const double *rk = raw + k * 147;
        if (c == 21) continue;
        double val;
        if (c < 14) val = rk[i * 21 + c];
        else { const int j = c - 14; val = (i <= j) ? rk[i * 21 + 14 + j] : rk[j * 21 + 14 + i]; }
        Gk_out[(k * 8 + i) * 22 + c] = val;
    }
    for (int t = tid; t < N * 7; t += NT) {
        const int k = t / 7, i = t % 7;
        const double *rk = raw + k * 147;
        double val = 0.0;
        if (cfg.quirk_dtau_dT) {
            for (int j = 0; j < 7; j++) {
                val += rk[i * 21 + 7 + j] * zl[14 * k + 7 + j];
                if (j >= i) val += rk[i * 21 + 14 + j] * zl[14 * N + 7 * k + j];
            }
        }
        Gk_out[(k * 8 + i) * 22 + 21] = val;
    }
    const double ts = 1.0 / (2.0 * NSEG);
    for (int r = tid; r < 14 * (N - 1); r += NT) {
        const int k = r / 14, rr = r % 14, s = k / 3, i = k % 3;
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++) acc += c_D[4 * i + j] * zl[14 * (3 * s + j) + rr];
        const double f = (rr < 7) ? zl[14 * k + 7 + rr] : zl[14 * N + 7 * k + rr - 7];
        ceq_out[r] = acc - ts * T * f;
    }
    __syncthreads();
}

template <int NT, int K, bool MAX>
__device__ __forceinline__ void reduce_m(double (&v)[K], double *red, int tid) { block_reduce<NT / 64, K, MAX>(v, red, tid); }

// ------------------------------------------------------------------------------------------------
template <int NSEG, int NARM>
__global__ __launch_bounds__(512) void k_init_m(mpcmp_config cfg, const mpcmp_model *models, WS ws, Xch xch, const double *warm_x,
                                                const double *warm_u, const double *warm_T, int reguess) {
    using D = DimM<NSEG, NARM>;
    constexpr int N = D::N, na = D::na, n = D::n, NT = D::NT;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *zl = lds + D::oZ, *scr = lds + D::oScr;
    const int tid = threadIdx.x, b = blockIdx.x;
    if (MPCMP_RETIRED(ws, b)) {      // arrived (k_advance): nothing is re-solved; only the launch-order bookkeeping of the slot
        if (tid == 0) { ws.perm[b] = b; ws.okey[b] = 0; if (b == 0) *ws.done = 0; }
        return;
    }
    // (the slot's previous solve, as in k_init: a failed solve leaves neither a re-guess nor multipliers behind)
    const bool prev_bad = (ws.status[b] & (MPCMP_STATUS_NAN | MPCMP_STATUS_NOT_PD | MPCMP_STATUS_XCH_DEAD | MPCMP_STATUS_T_OUT_OF_BOX)) != 0;
    __syncthreads();
    if (tid == 0) { ws.perm[b] = b; ws.okey[b] = 0; if (b == 0) *ws.done = 0; ws.qp_total[b] = 0; ws.status[b] = 0; ws.alpha[b] = 0.0; }
    const double *x0 = ws.x0 + (size_t)b * 14 * NARM, *xf = ws.xf + (size_t)b * 14 * NARM;
    if (reguess && prev_bad) { warm_x = ws.alt_x; warm_u = ws.alt_u; warm_T = ws.alt_T; reguess = 0; }      // (as in k_init)
    if (warm_x) {
        for (int v = tid; v < n - 1; v += NT) {
            const int a = v / na, w = v % na;
            double val;
            if (w < 14 * N) {
                const int k = w / 14, r = w % 14;
                val = warm_x[((size_t)b * N + k) * 14 * NARM + ext_x<NARM>(a, r)];
                // re-guess from the previous solution: "Fix initial and final point at correct place" (motionPlanner.cpp:199-207)
                if (reguess && k == 0) val = x0[ext_x<NARM>(a, r)];
                if (reguess && k == N - 1) val = xf[ext_x<NARM>(a, r)];
            } else {
                const int k = (w - 14 * N) / 7, j = (w - 14 * N) % 7;
                val = warm_u[((size_t)b * N + k) * 7 * NARM + 7 * a + j];
            }
            zl[v] = val;
        }
        if (tid == 0) zl[n - 1] = warm_T[b];
    } else {
        // stand-in for the warm start when none is given: per-joint quintic, zero boundary accelerations, common duration on a
        // geometric grid (same rule as k_init / the oracle's orc_warm_start, over all 7 NARM joints)
        double T = 0.05;
        double *coef = scr;   // [7 NARM][6]
        constexpr int NJ = 7 * NARM;
        auto set_coef = [&]() {
            if (tid < NJ) {
                const int a = tid / 7, j = tid % 7;
                const double q0 = x0[ext_x<NARM>(a, j)], v0 = x0[ext_x<NARM>(a, 7 + j)], q1 = xf[ext_x<NARM>(a, j)], v1 = xf[ext_x<NARM>(a, 7 + j)];
                const double h = q1 - q0, T2 = T * T, T3 = T2 * T;
                double *c = coef + 6 * tid;
                c[0] = q0; c[1] = v0; c[2] = 0.0;
                c[3] = (20.0 * h - (8.0 * v1 + 12.0 * v0) * T) / (2.0 * T3);
                c[4] = (-30.0 * h + (14.0 * v1 + 16.0 * v0) * T) / (2.0 * T3 * T);
                c[5] = (12.0 * h - 6.0 * (v1 + v0) * T) / (2.0 * T3 * T2);
            }
        };
        for (int it = 0; it < 200; it++) {
            set_coef();
            __syncthreads();
            int bad = 0;
            for (int t = tid; t < NJ * 65; t += NT) {
                const int jj = t / 65, s = t % 65, j = jj % 7;
                const double *c = coef + 6 * jj;
                const double tt = T * s / 64.0;
                const double vv = c[1] + tt * (2 * c[2] + tt * (3 * c[3] + tt * (4 * c[4] + tt * 5 * c[5])));
                const double aa = 2 * c[2] + tt * (6 * c[3] + tt * (12 * c[4] + tt * 20 * c[5]));
                if (fabs(vv) > cfg.ubx[7 + j] || fabs(aa) > cfg.ubu[j]) bad = 1;
            }
            const int anybad = __syncthreads_or(bad);
            if (!anybad || T * 1.05 > cfg.ubT) break;
            T *= 1.05;
        }
        __syncthreads();
        set_coef();
        __syncthreads();
        const double xi[4] = {-1.0, -0.5, 0.5, 1.0};
        for (int t = tid; t < N * NJ; t += NT) {
            const int k = t / NJ, jj = t % NJ, a = jj / 7, j = jj % 7;
            const int s = (k == N - 1) ? NSEG - 1 : k / 3, lj = k - 3 * s;
            const double tau = (s + 0.5 * (xi[lj] + 1.0)) / NSEG;
            const double *c = coef + 6 * jj;
            const double tt = tau * T;
            double *za = zl + a * na;
            za[14 * k + j] = c[0] + tt * (c[1] + tt * (c[2] + tt * (c[3] + tt * (c[4] + tt * c[5]))));
            za[14 * k + 7 + j] = c[1] + tt * (2 * c[2] + tt * (3 * c[3] + tt * (4 * c[4] + tt * 5 * c[5])));
            za[14 * N + 7 * k + j] = 2 * c[2] + tt * (6 * c[3] + tt * (12 * c[4] + tt * 20 * c[5]));
        }
        __syncthreads();
        if (tid < 14 * NARM) {
            const int a = tid / 14, r = tid % 14;
            zl[a * na + r] = x0[ext_x<NARM>(a, r)]; zl[a * na + 14 * (N - 1) + r] = xf[ext_x<NARM>(a, r)];
        }
        if (tid == 0) zl[n - 1] = T;
    }
    __syncthreads();
    for (int v = tid; v < n; v += NT) ws.z[(size_t)b * n + v] = zl[v];
    if (!cfg.carry_multipliers || prev_bad) for (int i = tid; i < D::mn; i += NT) ws.lam[(size_t)b * D::mn + i] = 0.0;
    if (NARM == 2) for (int i = tid; i < 2 * MPCMP_XCH_STRIDE; i += NT) xch.buf[(size_t)b * 2 * MPCMP_XCH_STRIDE + i] = MPCMP_XCH_EMPTY;
    const double T = zl[n - 1];
#pragma nounroll
    for (int a = 0; a < NARM; a++)
        linearise_arm<NSEG, NT>(cfg, models + a, zl + a * na, T, scr, ws.g + ((size_t)b * NARM + a) * 8 * N,
                                ws.Gk + ((size_t)b * NARM + a) * N * 176, ws.ceq + ((size_t)b * NARM + a) * D::meq, tid);
}

// ------------------------------------------------------------------------------------------------
template <int NSEG, int NARM>
__global__ __launch_bounds__(512) void k_step_m(mpcmp_config cfg, const mpcmp_model *models, WS ws, Xch xch, int final_iter, int sqp_it,
                                                double *sol_x, double *sol_u, double *sol_T, mpcmp_info *info) {
    using D = DimM<NSEG, NARM>;
    constexpr int N = D::N, na = D::na, n = D::n, meq = D::meq, ma = D::ma, NT = D::NT;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    if (MPCMP_RETIRED(ws, b)) { if (tid == 0) ws.qpit[b] = 0; step_tail(ws, lds, tid, NT); return; }
    double *zl = lds + D::oZ, *pl = lds + D::oP, *scr = lds + D::oScr, *sct = lds + D::oSct, *pv = lds + D::oPv, *xa = lds + D::oX0,
           *red = lds + D::oRed;
    const double *x0 = ws.x0 + (size_t)b * 14 * NARM, *xf = ws.xf + (size_t)b * 14 * NARM;
    double *lam = ws.lam + (size_t)b * D::mn;
    const double *y = ws.y + (size_t)b * D::mn;
    const double ts = 1.0 / (2.0 * NSEG);
    for (int v = tid; v < n; v += NT) { zl[v] = ws.z[(size_t)b * n + v]; pl[v] = ws.p[(size_t)b * n + v]; }
    if (tid < 14 * NARM) {          // per-arm copies of the boundary states: xa[a][0..13] = x0 of arm a, xa[NARM + a][..] = target
        const int a = tid / 14, r = tid % 14;
        xa[14 * a + r] = x0[ext_x<NARM>(a, r)]; xa[14 * (NARM + a) + r] = xf[ext_x<NARM>(a, r)];
    }
    // mu = ||lambda||_inf (polympc_redef.hpp:86)
    double r2[1] = {0.0};
    for (int i = tid; i < D::mn; i += NT) r2[0] = fmax(r2[0], fabs(lam[i]));
    reduce_m<NT, 1, true>(r2, red, tid);      // (also publishes zl / pl / xa)
    const double mu = r2[0];
    auto box_of = [&](int v, double &lo, double &hi) {
        if (v == n - 1) { lo = cfg.lbT; hi = cfg.ubT; }
        else { const int a = v / na; arm_box<NSEG>(cfg, xa + 14 * a, xa + 14 * (NARM + a), v % na, lo, hi); }
    };
    // l1 violation at the current iterate (:79)
    double c0[1] = {0.0};
    for (int i = tid; i < NARM * meq; i += NT) c0[0] += fabs(ws.ceq[(size_t)b * NARM * meq + i]);
    for (int i = tid; i < NARM * 8 * N; i += NT) c0[0] += viol(ws.g[(size_t)b * NARM * 8 * N + i], cfg.lbg[i % 8], cfg.ubg[i % 8]);
    for (int v = tid; v < n; v += NT) { double lo, hi; box_of(v, lo, hi); c0[0] += viol(zl[v], lo, hi); }
    reduce_m<NT, 1, false>(c0, red, tid);
    const double constr = c0[0];
    const double Tcur = zl[n - 1], pT = pl[n - 1];
    const double phi = Tcur + mu * constr;           // :93  (cost = T)
    const double Dphi = pT - mu * constr;            // :94  (cost gradient = e_T)
    // trial points alpha_t = tau^t, t = 0..ls_iters-2 (at most 9), all evaluated in one pass per arm
    const int ntr = cfg.ls_iters - 1 < 9 ? cfg.ls_iters - 1 : 9;
    double acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++) acc[t] = 0.0;
#pragma nounroll
    for (int a = 0; a < NARM; a++) {
        const double *za = zl + a * na, *pa = pl + a * na;
        const mpcmp_model *mdl_a = models + a;
        for (int t = tid; t < ntr * N * 7; t += NT) {
            const int tr = t / (N * 7), kj = t % (N * 7), k = kj / 7, j = kj % 7;
            double al = 1.0;
            for (int q = 0; q < tr; q++) al *= cfg.ls_tau;
            double s, c;
            sincos(za[14 * k + j] + al * pa[14 * k + j], &s, &c);
            sct[2 * t] = s; sct[2 * t + 1] = c;
        }
        __syncthreads();
#pragma nounroll
        for (int t = tid; t < ntr * N; t += NT) {
            const mpcmp_model *mdl = mdl_a;                 // (opaque per pass: see linearise_arm)
            asm volatile("" : "+s"(mdl));
            const int tr = t / N, k = t % N;
            double al = 1.0;
            for (int q = 0; q < tr; q++) al *= cfg.ls_tau;
            double v[7], ac[7], tau[7];
#pragma unroll
            for (int j = 0; j < 7; j++) {
                v[j] = za[14 * k + 7 + j] + al * pa[14 * k + 7 + j];
                ac[j] = za[14 * N + 7 * k + j] + al * pa[14 * N + 7 * k + j];
            }
            rnea_dir<false>(mdl, sct + 14 * t, v, ac, 0, 0, tau, nullptr);
            V3 ptool;
            fk_tool(mdl, sct + 14 * t, &ptool, nullptr, nullptr, nullptr);
            double s = viol(ptool.z, cfg.lbg[7], cfg.ubg[7]);
#pragma unroll
            for (int j = 0; j < 7; j++) s += viol(tau[j], cfg.lbg[j], cfg.ubg[j]);
            pv[t] = s;
        }
        __syncthreads();
        double al = 1.0;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            if (t > 0) al *= cfg.ls_tau;
            if (t < ntr) {
                for (int r = tid; r < meq; r += NT) {
                    const int k = r / 14, rr = r % 14, s = k / 3, i = k % 3;
                    double d = 0.0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int ix = 14 * (3 * s + j) + rr;
                        d += c_D[4 * i + j] * (za[ix] + al * pa[ix]);
                    }
                    const int fc = (rr < 7) ? 14 * k + 7 + rr : 14 * N + 7 * k + rr - 7;
                    d -= ts * (Tcur + al * pT) * (za[fc] + al * pa[fc]);
                    acc[t] += fabs(d);
                }
                for (int k = tid; k < N; k += NT) acc[t] += pv[t * N + k];
            }
        }
        __syncthreads();
    }
    {
        double al = 1.0;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            if (t > 0) al *= cfg.ls_tau;
            if (t < ntr) for (int v = tid; v < n; v += NT) { double lo, hi; box_of(v, lo, hi); acc[t] += viol(zl[v] + al * pl[v], lo, hi); }
        }
    }
    reduce_m<NT, 9, false>(acc, red, tid);
    double alpha = 1.0;
    // polympc_redef.hpp:97-117: i = 1 .. line_search_max_iter-1
    for (int t = 0; t < cfg.ls_iters - 1; t++) {
        double l1 = 0.0;
#pragma unroll
        for (int q = 0; q < 9; q++) if (q == t) l1 = acc[q];
        const double phis = (Tcur + alpha * pT) + mu * l1;
        if (phis <= phi + alpha * cfg.ls_eta * Dphi) break;     // :108
        alpha *= cfg.ls_tau;
    }
    // update
    __syncthreads();
    for (int v = tid; v < n; v += NT) { zl[v] += alpha * pl[v]; ws.z[(size_t)b * n + v] = zl[v]; }
    for (int i = tid; i < D::mn; i += NT) lam[i] += alpha * (y[i] - lam[i]);
    if (NARM == 2) for (int i = tid; i < 2 * MPCMP_XCH_STRIDE; i += NT) xch.buf[(size_t)b * 2 * MPCMP_XCH_STRIDE + i] = MPCMP_XCH_EMPTY;
    __syncthreads();
    const double Tn = zl[n - 1];
    double *gout = ws.g + (size_t)b * NARM * 8 * N, *ceqo = ws.ceq + (size_t)b * NARM * meq;
#pragma nounroll
    for (int a = 0; a < NARM; a++)
        linearise_arm<NSEG, NT>(cfg, models + a, zl + a * na, Tn, scr, gout + a * 8 * N, ws.Gk + ((size_t)b * NARM + a) * N * 176,
                                ceqo + a * meq, tid);
    if (tid == 0) ws.alpha[b] = alpha;
    if (final_iter) {
        __threadfence_block();
        double s1[1] = {0.0}, mxs[3] = {0, 0, 0};
        int bad = 0;
        for (int i = tid; i < NARM * meq; i += NT) { const double c = ceqo[i]; s1[0] += fabs(c); mxs[0] = fmax(mxs[0], fabs(c)); }
        for (int i = tid; i < NARM * 8 * N; i += NT) { const double vv = viol(gout[i], cfg.lbg[i % 8], cfg.ubg[i % 8]); s1[0] += vv; mxs[1] = fmax(mxs[1], vv); }
        for (int v = tid; v < n; v += NT) { double lo, hi; box_of(v, lo, hi); s1[0] += viol(zl[v], lo, hi); if (!isfinite(zl[v])) bad = 1; }
        if (tid < 14 * NARM) { const int a = tid / 14, r = tid % 14; mxs[2] = fabs(zl[a * na + 14 * (N - 1) + r] - xa[14 * (NARM + a) + r]); }
        reduce_m<NT, 1, false>(s1, red, tid);
        reduce_m<NT, 3, true>(mxs, red, tid);
        const int anybad = __syncthreads_or(bad);
        // solution in the ABI's layout: sol_x [B][N][14 NARM] = [q(7 NARM); qd(7 NARM)], sol_u [B][N][7 NARM]
        for (int v = tid; v < n - 1; v += NT) {
            const int a = v / na, w = v % na;
            if (w < 14 * N) { const int k = w / 14, r = w % 14; sol_x[((size_t)b * N + k) * 14 * NARM + ext_x<NARM>(a, r)] = zl[v]; }
            else { const int k = (w - 14 * N) / 7, j = (w - 14 * N) % 7; sol_u[((size_t)b * N + k) * 7 * NARM + 7 * a + j] = zl[v]; }
        }
        if (tid == 0) sol_T[b] = zl[n - 1];
        if (tid == 0) {
            mpcmp_info o;
            o.T = zl[n - 1]; o.viol_l1 = s1[0]; o.defect_inf = mxs[0]; o.path_viol_inf = mxs[1]; o.term_err_inf = mxs[2];
            o.last_alpha = alpha; o.qp_iters_total = ws.qp_total[b]; o.sqp_iters = sqp_it + 1;
            report_status(cfg, ws.status[b], anybad, o);
            if (info) info[b] = o;
            ws.status[b] |= o.status & (MPCMP_STATUS_NAN | MPCMP_STATUS_T_OUT_OF_BOX);      // (hard bits of the final iterate, as in k_step)
        }
    }
    step_tail(ws, lds, tid, NT);      // launch order of the next QP launch (see k_step)
}

// warm start of a multi-arm OCP from the single-arm generator (k_warm_jerk run on B*NARM arm problems): the common duration
// is the slowest arm's and a faster arm's trajectory is played back uniformly slower (q(t) = q_a(s t), qd = s qd_a,
// qdd = s^2 qdd_a with s = T_a / T <= 1: every limit still holds); exact end states (motionPlanner.cpp:202-203).
// in: ax [B*NARM][N][14], au [B*NARM][N][7], aT [B*NARM]  (arm problem index = b*NARM + a);  out: ABI layout.
template <int NARM>
__global__ __launch_bounds__(256) void k_warm_merge(int N, int B, const double *ax, const double *au, const double *aT, const double *x0,
                                                    const double *xf, double *wx, double *wu, double *wT) {
    const int b = blockIdx.x, tid = threadIdx.x;
    double T = 0.0;
    for (int a = 0; a < NARM; a++) T = fmax(T, aT[b * NARM + a]);
    for (int t = tid; t < N * 21 * NARM; t += blockDim.x) {
        const int a = t / (N * 21), w = t % (N * 21), k = w / 21, c = w % 21;
        const double s = aT[b * NARM + a] / T;
        const size_t ap = (size_t)b * NARM + a;
        if (c < 14) {
            double val = ax[(ap * N + k) * 14 + c] * (c < 7 ? 1.0 : s);
            if (k == 0) val = x0[(size_t)b * 14 * NARM + ext_x<NARM>(a, c)];
            if (k == N - 1) val = xf[(size_t)b * 14 * NARM + ext_x<NARM>(a, c)];
            wx[((size_t)b * N + k) * 14 * NARM + ext_x<NARM>(a, c)] = val;
        } else {
            wu[((size_t)b * N + k) * 7 * NARM + 7 * a + (c - 14)] = au[(ap * N + k) * 7 + (c - 14)] * (s * s);
        }
    }
    if (tid == 0) wT[b] = T;
}

// split external states [B][14 NARM] into per-arm states [B*NARM][14] (input of k_warm_jerk)
template <int NARM>
__global__ void k_split_states(int B, const double *xe, double *xa) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * NARM * 14) return;
    const int b = i / (NARM * 14), a = (i / 14) % NARM, r = i % 14;
    xa[i] = xe[(size_t)b * 14 * NARM + ext_x<NARM>(a, r)];
}

}  // namespace mpcmp
