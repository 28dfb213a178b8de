// qp_kernel_v2.hpp — k_qp2: the QP kernel specialised for NUM_SEG <= 4 (N <= 13), one OCP per workgroup of 1024
// threads (= one CU: 16 waves, 4 per SIMD, <=128 VGPRs).
//
// Same arithmetic as k_qp (solver_kernels.hpp) — assemble K, nested-dissection factorisation with explicit block
// inverses, OSQP-form ADMM — re-mapped for the CDNA4 execution model:
//   * every factor matrix lives in VGPRs for the whole ADMM loop ([G_s; E_s^T], E_s, S^-1: ~28k doubles spread
//     over the 1024 threads); LDS carries only the exchanged vectors and the (padded) path Jacobians;
//   * each thread owns a 2-row x k-column register block, so one LDS operand read feeds two FMAs; the
//     k-way partial sums are combined with DPP (quad_perm / row_half_mirror), not through LDS;
//   * waves are role-specialised (group A = segment quads for the interior solves + path rows, group B =
//     interface solve, dynamics rows, variables), 5 workgroup barriers per ADMM iteration.
// Factorisation: segment blocks are processed two at a time in LDS (symmetric sweep), then loaded by their
// owner threads, so the LDS footprint stays at ~125 KB.
#pragma once
#include "solver_kernels.hpp"

namespace mpcmp {

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sum2(double x) { return x + dpp_mov<0xB1>(x); }                  // lanes i, i^1
__device__ __forceinline__ double sum4(double x) { x += dpp_mov<0xB1>(x); return x + dpp_mov<0x4E>(x); }
__device__ __forceinline__ double sum8(double x) { x = sum4(x); return x + dpp_mov<0x141>(x); }   // + row_half_mirror
__device__ __forceinline__ double wave_sum(double x) {
    x = sum8(x); x += dpp_mov<0x140>(x);                                                           // row_mirror: 16 lanes
    x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
    return x;
}

template <int NSEG>
struct Qp2 {
    using D = Dim<NSEG>;
    static constexpr int NT = 1024, NW = 16;
    static constexpr int QPS = 40;                 // quads per segment: 25 G row-pairs + 15 E^T row-pairs
    static constexpr int NA = 4 * QPS * NSEG;      // group A threads (640 at NSEG=4)
    static constexpr int NB = NT - NA;             // group B threads
    static constexpr int NPR = (D::nI + 1) / 2;    // interface row pairs (39)
    static constexpr int GS = 23;                  // padded row stride of the path Jacobians in LDS
    static constexpr int XS = 24;                  // node-major x~ stride: [x_k(14) u_k(7) T pad pad]
    static constexpr int HS = 2;                   // segments factorised concurrently
    static_assert(NA % 64 == 0 && 8 * NPR <= NB && D::m - D::meq <= 8 * D::N && D::n <= NB && D::meq <= NB, "mapping");
    static_assert(8 * D::N <= 60 * NSEG, "path-row lanes must fit the E^T quads");
    // LDS (doubles)
    static constexpr int oGk = 0;                              // [N][8][GS]
    static constexpr int oRhsJ = oGk + D::N * 8 * GS;          // [NSEG][56]   rhs, interior part (zero padded)
    static constexpr int oRhsI = oRhsJ + NSEG * 56;            // [80]         rhs, interface part
    static constexpr int oPart = oRhsI + 80;                   // [NSEG][32]   E_s^T b_Js
    static constexpr int oXC = oPart + NSEG * 32;              // [NSEG][32]   x_I restricted to C_s
    static constexpr int oXn = oXC + NSEG * 32;                // [N][XS]      x~ node-major
    static constexpr int oXx = oXn + D::N * XS;                // [N][XS]      x  node-major (termination tests)
    static constexpr int oWg = oXx + D::N * XS;                // [m]          w = rho z - y   (general rows)
    static constexpr int oYs = oWg + D::m;                     // [m]          y (termination tests)
    static constexpr int oTp = oYs + D::m;                     // [m]          coefT_r * w_r
    static constexpr int oMisc = oTp + D::m;                   // [8]          0: zero slot, 1: T-variable base
    static constexpr int oRed = oMisc + 8;                     // [NW*8]
    static constexpr int oS = oRed + NW * 8;                   // packed S (factorisation)
    static constexpr int oKJJ = oS + D::SP;                    // [HS][JP]
    static constexpr int oKJC = oKJJ + HS * D::JP;             // [HS][JC]
    static constexpr int oEh = oKJC + HS * D::JC;              // [HS][JC]
    static constexpr int oZ = oEh + HS * D::JC;                // [n]
    static constexpr int size = oZ + D::n;
};

// shared context of the two role groups
template <int NSEG>
struct Qp2Ctx {
    const mpcmp_config *cfg;
    WS ws;
    double *lds;
    int tid, b;
    double ts, tsT, rho_in, rho_eq, sigma, alpha;
};

#ifdef MPCMP_STAMPS
#define STAMP2(slot) do { if (c.tid == 0) { const unsigned long long now_ = clock64(); stamp_acc[slot] += now_ - stamp_t; stamp_t = now_; } } while (0)
#else
#define STAMP2(slot) do { } while (0)
#endif

// ---- group A: segment quads — interior solves (P1, P3) and path rows ------------------------------------
template <int NSEG>
__device__ __forceinline__ void qp2_group_a(const Qp2Ctx<NSEG> &c, const double (&m1)[2][14], const double (&e3)[2][8],
                                            unsigned long long *stamp_acc_out) {
    using D = Dim<NSEG>;
    using L = Qp2<NSEG>;
    constexpr int N = D::N, n = D::n, meq = D::meq, m = D::m, GS = L::GS, XS = L::XS;
    double *lds = c.lds;
    const mpcmp_config &cfg = *c.cfg;
    const int tid = c.tid, b = c.b;
    double *red = lds + L::oRed, *gkl = lds + L::oGk;
    double *rhsJ = lds + L::oRhsJ, *partl = lds + L::oPart, *xC = lds + L::oXC, *xn = lds + L::oXn, *xx = lds + L::oXx,
           *wg = lds + L::oWg, *ys = lds + L::oYs, *tpl = lds + L::oTp, *misc = lds + L::oMisc;
#ifdef MPCMP_STAMPS
    unsigned long long stamp_acc[16] = {0}, stamp_t = clock64();
#endif
    const int Q = tid >> 2, part = tid & 3, seg = Q / L::QPS, lp = Q % L::QPS;
    const bool isG = lp < 25;
    const int et = (seg * 15 + (lp - 25)) * 4 + part;
    const bool isPath = !isG && et < 8 * N;
    const int pk = et >> 3, prp = (et & 7) >> 1, phalf = et & 1;
    // path row owned by this lane (ADMM state), and the 2x11 coefficient block it helps to evaluate
    double lg = 0, ug = 0, rr_ = c.rho_in, coefT = 0, zg = 0, yg = 0;
    int myrow = 0;
    if (isPath) {
        const int q = 2 * prp + phalf;
        myrow = meq + 8 * pk + q;
        coefT = c.ws.Gk[((size_t)b * N + pk) * 176 + q * 22 + 21];
        const double gv = c.ws.g[(size_t)b * 8 * N + 8 * pk + q];
        lg = cfg.lbg[q] - gv; ug = cfg.ubg[q] - gv;
        rr_ = (ug - lg < 1e-4) ? c.rho_eq : c.rho_in;
    }
    const int groff = isPath ? (pk * 8 + 2 * prp) * GS + phalf * 11 : 0;
    const int xnoff = isPath ? pk * XS + phalf * 11 : 0;
    int jdst = -1;      // G quads: where row 2lp+part of x_J goes in the node-major x~
    if (isG && part < 2 && 2 * lp + part < 49) {
        const int v = c.ws.ext_of_int[49 * seg + 2 * lp + part];
        jdst = v < 14 * N ? (v / 14) * XS + v % 14 : ((v - 14 * N) / 7) * XS + 14 + (v - 14 * N) % 7;
    }
    auto row_dot_path = [&](const double *xe) -> double {
        const double *g0 = gkl + groff, *g1 = g0 + GS, *xv = xe + xnoff;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int cc = 0; cc < 11; cc++) { const double xc = xv[cc]; a0 += g0[cc] * xc; a1 += g1[cc] * xc; }
        a0 = sum2(a0); a1 = sum2(a1);
        return phalf ? a1 : a0;
    };
    const double alpha = c.alpha;
    int it = 0, done = 0;
    for (it = 1; it <= cfg.qp_iters; it++) {
        // ---- A (group A part): wave 0 sums the T column of A^T w ----
        if (tid < 64) {
            double sacc = 0.0;
            for (int r = tid; r < m; r += 64) sacc += tpl[r];
            sacc = wave_sum(sacc);
            if (tid == 0) misc[2] = sacc;
        }
        __syncthreads();
        STAMP2(3);
        // ---- P1: [t ; E^T b] = [G ; E^T] b_J ----
        double t0, t1;
        {
            const double *bj = rhsJ + 56 * seg + 14 * part;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int j = 0; j < 14; j++) { const double bv = bj[j]; a0 += m1[0][j] * bv; a1 += m1[1][j] * bv; }
            t0 = sum4(a0); t1 = sum4(a1);
            if (!isG && part < 2) {
                const int cc = 2 * (lp - 25) + part;
                if (cc < 29) partl[seg * 32 + cc] = part ? t1 : t0;
            }
        }
        __syncthreads();
        STAMP2(4);
        // ---- P2: (group B) ----
        __syncthreads();
        STAMP2(5);
        // ---- P3: x_J = t - E_s x_C(s) ----
        if (isG) {
            const double *xc = xC + 32 * seg + 8 * part;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int j = 0; j < 8; j++) { const double xv = xc[j]; a0 += e3[0][j] * xv; a1 += e3[1][j] * xv; }
            a0 = sum4(a0); a1 = sum4(a1);
            if (jdst >= 0) xn[jdst] = part ? (t1 - a1) : (t0 - a0);
        }
        __syncthreads();
        STAMP2(6);
        // ---- E: path rows ----
        const bool check = (it % cfg.check_every == 0);
        if (isPath) {
            const double zt = row_dot_path(xn);
            const double zr = alpha * zt + (1.0 - alpha) * zg;
            const double zn = clip(zr + yg / rr_, lg, ug);
            yg += rr_ * (zr - zn);
            zg = zn;
            const double w = rr_ * zg - yg;
            wg[myrow] = w;
            tpl[myrow] = coefT * w;
            if (check) ys[myrow] = yg;
        }
        __syncthreads();
        STAMP2(7);
        if (check) {
            double sums[2] = {isPath ? coefT * yg : 0.0, 0.0};
            block_reduce<L::NW, 2, false>(sums, red, tid);
            double mx[6] = {0, 0, 0, 0, 0, 0};
            if (isPath) {
                const double ax = row_dot_path(xx);
                mx[0] = fabs(ax - zg); mx[1] = fabs(ax); mx[2] = fabs(zg);
            }
            block_reduce<L::NW, 6, true>(mx, red, tid);
            const double ep = cfg.eps_abs + cfg.eps_rel * fmax(mx[1], mx[2]);
            const double ed = cfg.eps_abs + cfg.eps_rel * fmax(fmax(mx[4], mx[5]), 1.0);
            if (mx[0] <= ep && mx[3] <= ed) done = 1;
        }
        STAMP2(8);
        if (done) break;
    }
    if (it > cfg.qp_iters) it = cfg.qp_iters;
    if (isPath) c.ws.y[(size_t)b * D::mn + myrow] = yg;
    if (tid == 0) { c.ws.qpit[b] = it; c.ws.qp_total[b] += it; }
#ifdef MPCMP_STAMPS
    if (tid == 0) { for (int k = 3; k < 16; k++) stamp_acc_out[k] = stamp_acc[k]; stamp_acc_out[15] = it; }
#endif
    (void)n; (void)stamp_acc_out;
}

// ---- group B: interface solve (P2), dynamics rows, variables ---------------------------------------------
struct VarRole {
    double lb, ub, rb, hd, ha, qv, cf, dA[3], dB[3];
    int rA, rB, rf, pb, gcol, xpos, rpos;
    bool hasG;
};

template <int NSEG>
__device__ __forceinline__ VarRole make_var_role(const mpcmp_config &cfg, const WS &ws, int b, int u, bool isVar, double ts,
                                                 double tsT, double rho_in, double rho_eq) {
    using D = Dim<NSEG>;
    using L = Qp2<NSEG>;
    constexpr int N = D::N, n = D::n, meq = D::meq, nJ = D::nJ, GS = L::GS, XS = L::XS;
    const double *zg_ = ws.z + (size_t)b * n;
    VarRole vr;
    vr.lb = vr.ub = 0; vr.rb = rho_in; vr.hd = vr.ha = vr.qv = vr.cf = 0;
    vr.rA = vr.rB = vr.rf = 0; vr.pb = meq; vr.gcol = 0; vr.xpos = 0; vr.rpos = L::oMisc; vr.hasG = false;
#pragma unroll
    for (int i = 0; i < 3; i++) { vr.dA[i] = 0; vr.dB[i] = 0; }
    if (isVar) {
        const int v = u;
        const int ipos = int_of_ext(NSEG, v);
        vr.rpos = ipos < nJ ? L::oRhsJ + 56 * (ipos / 49) + ipos % 49 : L::oRhsI + (ipos - nJ);
        double lo, hi;
        var_box<NSEG>(cfg, ws.x0 + 14 * b, ws.xf + 14 * b, v, lo, hi);
        vr.rb = (hi - lo < 1e-4) ? rho_eq : rho_in;
        const double zv = zg_[v];
        vr.lb = lo - zv; vr.ub = hi - zv;
        if (v < 14 * N) {
            const int k = v / 14, cc = v % 14;
            vr.pb = meq + 8 * k; vr.xpos = k * XS + cc;
            if (k % 3 != 0) {
                vr.rA = 14 * 3 * (k / 3) + cc;
#pragma unroll
                for (int i = 0; i < 3; i++) vr.dA[i] = c_D[4 * i + k % 3];
            } else {
                if (k < N - 1) {
                    vr.rA = 14 * k + cc;
#pragma unroll
                    for (int i = 0; i < 3; i++) vr.dA[i] = c_D[4 * i + 0];
                }
                if (k > 0) {
                    vr.rB = 14 * (k - 3) + cc;
#pragma unroll
                    for (int i = 0; i < 3; i++) vr.dB[i] = c_D[4 * i + 3];
                }
            }
            if (cc >= 7 && k <= N - 2) { vr.rf = 14 * k + (cc - 7); vr.cf = -tsT; vr.ha = -ts * ws.lam[(size_t)b * D::mn + vr.rf]; }
            vr.gcol = k * 8 * GS + cc; vr.hasG = true;
        } else if (v < 21 * N) {
            const int k = (v - 14 * N) / 7, cc = (v - 14 * N) % 7;
            vr.pb = meq + 8 * k; vr.xpos = k * XS + 14 + cc;
            if (k <= N - 2) { vr.rf = 14 * k + 7 + cc; vr.cf = -tsT; vr.ha = -ts * ws.lam[(size_t)b * D::mn + vr.rf]; }
            vr.gcol = k * 8 * GS + 14 + cc; vr.hasG = true;
        } else {
            vr.qv = 1.0; vr.xpos = 21;      // T is replicated at slot 21 of every node row
        }
        vr.hd = fabs(vr.ha) + cfg.hess_reg;   // Gershgorin shift, polympc_redef.hpp:57-70
    }
    return vr;
}

template <int NSEG>
__device__ __forceinline__ void qp2_group_b(const Qp2Ctx<NSEG> &c, const double (&s2)[2][10], const int (&rof)[10]) {
    using D = Dim<NSEG>;
    using L = Qp2<NSEG>;
    constexpr int N = D::N, n = D::n, meq = D::meq, m = D::m, nI = D::nI, GS = L::GS, XS = L::XS;
    double *lds = c.lds;
    const mpcmp_config &cfg = *c.cfg;
    const int tid = c.tid, b = c.b, u = tid - L::NA;
    double *red = lds + L::oRed, *gkl = lds + L::oGk;
    double *rhsI = lds + L::oRhsI, *partl = lds + L::oPart, *xC = lds + L::oXC, *xn = lds + L::oXn, *xx = lds + L::oXx,
           *wg = lds + L::oWg, *ys = lds + L::oYs, *tpl = lds + L::oTp, *misc = lds + L::oMisc;
    const bool isP2 = (u >> 3) < L::NPR;
    const int rp2 = u >> 3, part2 = u & 7;
    const bool isDyn = u < meq, isVar = u < n, isT = u == n - 1;
    const double *zg_ = c.ws.z + (size_t)b * n;
    VarRole vr = make_var_role<NSEG>(cfg, c.ws, b, u, isVar, c.ts, c.tsT, c.rho_in, c.rho_eq);
    if (isT) { vr.hd = lds[L::oMisc + 3] + cfg.hess_reg; vr.ha = 0.0; }
    // dynamics row role
    double rcoef[6] = {0, 0, 0, 0, 0, 0}, lg = 0, ug = 0, zg = 0, yg = 0;
    int ix0 = 0, ixf = 0, ixT = 21;
    if (isDyn) {
        const int r = u, k = r / 14, rr = r % 14, s = k / 3, i = k % 3;
        ix0 = 3 * s * XS + rr;
        ixf = k * XS + ((rr < 7) ? 7 + rr : 14 + rr - 7);
        ixT = k * XS + 21;
#pragma unroll
        for (int j = 0; j < 4; j++) rcoef[j] = c_D[4 * i + j];
        rcoef[4] = -c.tsT;
        rcoef[5] = -c.ts * zg_[(rr < 7) ? 14 * k + 7 + rr : 14 * N + 7 * k + rr - 7];
        lg = ug = -c.ws.ceq[(size_t)b * meq + r];
    }
    const double rr_ = c.rho_eq, coefT = rcoef[5];
    // interface solve output: where x_I[row] goes
    int xdst = -1, cdst1 = -1, cdst2 = -1, myIrow = -1;
    if (isP2 && part2 < 2) {
        const int row = 2 * rp2 + part2;
        if (row < nI) {
            myIrow = row;
            if (row < 14 * (NSEG + 1)) {
                const int sb = row / 14, cc = row % 14;
                xdst = 3 * sb * XS + cc;
                if (sb < NSEG) cdst1 = L::oXC + sb * 32 + cc;
                if (sb > 0) cdst2 = L::oXC + (sb - 1) * 32 + 14 + cc;
            } else if (row < nI - 1) {
                xdst = (N - 1) * XS + 14 + (row - 14 * (NSEG + 1));
            }
        }
    }
    auto row_dot_dyn = [&](const double *xe) -> double {
        return rcoef[0] * xe[ix0] + rcoef[1] * xe[ix0 + XS] + rcoef[2] * xe[ix0 + 2 * XS] + rcoef[3] * xe[ix0 + 3 * XS] +
               rcoef[4] * xe[ixf] + rcoef[5] * xe[ixT];
    };
    auto col_gather = [&](const double *w) -> double {
        double s = vr.cf * w[vr.rf];
#pragma unroll
        for (int i = 0; i < 3; i++) s += vr.dA[i] * w[vr.rA + 14 * i];
#pragma unroll
        for (int i = 0; i < 3; i++) s += vr.dB[i] * w[vr.rB + 14 * i];
        if (vr.hasG) {
            const double *gc = gkl + vr.gcol;
#pragma unroll
            for (int q = 0; q < 8; q++) s += gc[q * GS] * w[vr.pb + q];
        }
        return s;
    };
    const double alpha = c.alpha, sigma = c.sigma;
    double x = 0, zb = 0, yb = 0;
    int it = 0, done = 0;
    for (it = 1; it <= cfg.qp_iters; it++) {
        // ---- A: rhs = sigma x - q + rho_b zb - yb + A^T w ----
        if (isVar) {
            const double base = sigma * x - vr.qv + (vr.rb * zb - yb);
            if (isT) misc[1] = base;
            else lds[vr.rpos] = base + col_gather(wg);
        }
        __syncthreads();
        // ---- P1: (group A) ----
        __syncthreads();
        // ---- P2: x_I = S^-1 (b_I - sum_s E_s^T b_Js) ----
        if (isP2) {
            double a0 = 0.0, a1 = 0.0;
            const double *bi = rhsI + part2 * 10;
#pragma unroll
            for (int j = 0; j < 10; j++) {
                double r = bi[j] - lds[rof[j] >> 16] - lds[rof[j] & 0xffff];
                if (part2 * 10 + j == nI - 1) {       // column T: b_T assembled here
                    r = misc[1] + misc[2];
#pragma unroll
                    for (int s = 0; s < NSEG; s++) r -= partl[s * 32 + 28];
                }
                a0 += s2[0][j] * r; a1 += s2[1][j] * r;
            }
            a0 = sum8(a0); a1 = sum8(a1);
            if (myIrow >= 0) {
                const double xi = part2 ? a1 : a0;
                if (myIrow == nI - 1) {
#pragma unroll
                    for (int k = 0; k < N; k++) xn[k * XS + 21] = xi;
#pragma unroll
                    for (int s = 0; s < NSEG; s++) xC[s * 32 + 28] = xi;
                } else {
                    xn[xdst] = xi;
                    if (cdst1 >= 0) lds[cdst1] = xi;
                    if (cdst2 >= 0) lds[cdst2] = xi;
                }
            }
        }
        __syncthreads();
        // ---- P3: (group A) ----
        __syncthreads();
        // ---- E: dynamics rows and variables ----
        const bool check = (it % cfg.check_every == 0);
        if (isDyn) {
            const double zt = row_dot_dyn(xn);
            const double zr = alpha * zt + (1.0 - alpha) * zg;
            const double zn = clip(zr + yg / rr_, lg, ug);
            yg += rr_ * (zr - zn);
            zg = zn;
            const double w = rr_ * zg - yg;
            wg[u] = w;
            tpl[u] = coefT * w;
            if (check) ys[u] = yg;
        }
        if (isVar) {
            const double xtv = xn[vr.xpos];
            x = alpha * xtv + (1.0 - alpha) * x;
            const double zr = alpha * xtv + (1.0 - alpha) * zb;
            const double zn = clip(zr + yb / vr.rb, vr.lb, vr.ub);
            yb += vr.rb * (zr - zn);
            zb = zn;
            if (check) {
                if (isT) { for (int k = 0; k < N; k++) xx[k * XS + 21] = x; }
                else xx[vr.xpos] = x;
            }
        }
        __syncthreads();
        if (check) {
            double sums[2] = {isDyn ? coefT * yg : 0.0, (isVar && !isT) ? vr.ha * x : 0.0};
            block_reduce<L::NW, 2, false>(sums, red, tid);
            double mx[6] = {0, 0, 0, 0, 0, 0};
            if (isDyn) {
                const double ax = row_dot_dyn(xx);
                mx[0] = fabs(ax - zg); mx[1] = fabs(ax); mx[2] = fabs(zg);
            }
            if (isVar) {
                mx[0] = fmax(mx[0], fabs(x - zb)); mx[1] = fmax(mx[1], fabs(x)); mx[2] = fmax(mx[2], fabs(zb));
                double hx, aty;
                if (isT) { hx = vr.hd * x + sums[1]; aty = sums[0] + yb; }
                else { hx = vr.hd * x + vr.ha * xx[21]; aty = col_gather(ys) + yb; }
                mx[3] = fabs(hx + aty + vr.qv); mx[4] = fabs(hx); mx[5] = fabs(aty);
            }
            block_reduce<L::NW, 6, true>(mx, red, tid);
            const double ep = cfg.eps_abs + cfg.eps_rel * fmax(mx[1], mx[2]);
            const double ed = cfg.eps_abs + cfg.eps_rel * fmax(fmax(mx[4], mx[5]), 1.0);
            if (mx[0] <= ep && mx[3] <= ed) done = 1;
        }
        if (done) break;
    }
    if (isVar) {
        c.ws.p[(size_t)b * n + u] = x;
        c.ws.y[(size_t)b * D::mn + m + u] = yb;
    }
    if (isDyn) c.ws.y[(size_t)b * D::mn + u] = yg;
}

template <int NSEG>
__global__ __launch_bounds__(1024) void k_qp2(mpcmp_config cfg, WS ws) {
    using D = Dim<NSEG>;
    using L = Qp2<NSEG>;
    constexpr int N = D::N, n = D::n, meq = D::meq, nJ = D::nJ, nI = D::nI, NT = L::NT;
    constexpr int GS = L::GS, XS = L::XS;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    double *red = lds + L::oRed, *gkl = lds + L::oGk;
    Qp2Ctx<NSEG> c;
    c.cfg = &cfg; c.ws = ws; c.lds = lds; c.tid = tid; c.b = b;
    c.ts = 1.0 / (2.0 * NSEG);
    c.rho_in = cfg.rho; c.rho_eq = cfg.rho * cfg.rho_eq_scale; c.sigma = cfg.sigma; c.alpha = cfg.alpha;
    const double ts = c.ts, rho_in = c.rho_in, rho_eq = c.rho_eq, sigma = c.sigma;
    const double *zg_ = ws.z + (size_t)b * n;
    const double *Gkg = ws.Gk + (size_t)b * N * 176;
    const double T = zg_[n - 1];
    const double tsT = ts * T;
    c.tsT = tsT;
    int status = 0;
#ifdef MPCMP_STAMPS
    unsigned long long stamp_acc[16] = {0}, stamp_t = clock64();
#endif
    const bool grpA = tid < L::NA;
    const int u = tid - L::NA;
    const bool isVar = !grpA && u < n, isT = !grpA && u == n - 1;

    for (int i = tid; i < L::oRed - L::oRhsJ; i += NT) lds[L::oRhsJ + i] = 0.0;     // exchanged vectors and their pads
    __syncthreads();
    // ---- variable role (group B): only the Hessian/rho entries needed by the assembly are kept live here;
    //      the full role is rebuilt inside qp2_group_b (keeps the factorisation's register footprint small) ----
    double v_diag = 0.0, v_ha = 0.0;
    int ipos = 0;
    {
        VarRole vr = make_var_role<NSEG>(cfg, ws, b, u, isVar, ts, tsT, rho_in, rho_eq);
        double sv[1] = {isVar && !isT ? fabs(vr.ha) : 0.0};
        block_reduce<L::NW, 1, false>(sv, red, tid);
        if (isT) { vr.hd = sv[0] + cfg.hess_reg; vr.ha = 0.0; }
        if (tid == 0) lds[L::oMisc + 3] = sv[0];
        v_diag = vr.hd + sigma + vr.rb; v_ha = vr.ha;
        if (isVar) ipos = int_of_ext(NSEG, u);
    }
    STAMP(0);
    // ---------------- assembly + factorisation ----------------
    double *S = lds + L::oS, *KJJ = lds + L::oKJJ, *KJC = lds + L::oKJC, *Eh = lds + L::oEh, *zl = lds + L::oZ;
    for (int i = tid; i < N * 176; i += NT) gkl[(i / 22) * GS + (i % 22)] = Gkg[i];
    for (int v = tid; v < n; v += NT) zl[v] = zg_[v];
    __syncthreads();
    auto term_val = [&](uint32_t t) -> double {
        const int r = t >> 16, a = (t >> 8) & 255, cc = t & 255;
        double va, vb, rho;
        if (r < meq) {
            const int k = r / 14, rr = r % 14, i = k % 3;
            const int fc = (rr < 7) ? 14 * k + 7 + rr : 14 * N + 7 * k + rr - 7;
            const double cT = -ts * zl[fc];
            va = a < 4 ? c_D[4 * i + a] : (a == 4 ? -tsT : cT);
            vb = cc < 4 ? c_D[4 * i + cc] : (cc == 4 ? -tsT : cT);
            rho = rho_eq;
        } else {
            const double *row = gkl + (r - meq) * GS;
            va = row[a]; vb = row[cc];
            rho = rho_in;
        }
        return rho * va * vb;
    };
    auto assemble = [&](int e0, int cnt, double *out) {
        for (int e = tid; e < cnt; e += NT) {
            double acc = 0.0;
            const int t1 = ws.entry_ptr[e0 + e + 1];
            for (int t = ws.entry_ptr[e0 + e]; t < t1; t++) acc += term_val(ws.terms[t]);
            out[e] = acc;
        }
    };
    auto tri_decode = [](int e, int &i, int &j) {
        i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= e) i++;
        while (i * (i + 1) / 2 > e) i--;
        j = e - i * (i + 1) / 2;
    };
    // symmetric sweep of `nblk` packed nb x nb SPD blocks (stride bstride) in LDS: A <- -(A^-1);
    // each thread owns up to EPT fixed entries (decoded once)
    auto sweep = [&](double *A, int nb, int cnt, int nblk, int bstride) {
        constexpr int EPT = 4;
        int ei[EPT], ej[EPT], eo[EPT], ek[EPT];
#pragma unroll
        for (int q = 0; q < EPT; q++) {
            const int ge = tid + q * NT;
            if (ge < cnt * nblk) {
                const int blk = ge / cnt, e = ge % cnt;
                tri_decode(e, ei[q], ej[q]);
                ek[q] = blk * bstride; eo[q] = blk * bstride + e;
            } else { ei[q] = -1; ej[q] = 0; eo[q] = 0; ek[q] = 0; }
        }
        const int myblk = tid / nb, myi = tid % nb;
        for (int k = 0; k < nb; k++) {
            const int kk = packed(k, k);
            double rdm = 0.0;
            if (tid < nb * nblk) {
                const double d = A[myblk * bstride + kk];
                if (!(d > 0.0)) status |= 2;
                rdm = 1.0 / d;
            }
#pragma unroll
            for (int q = 0; q < EPT; q++) {
                if (ei[q] >= 0 && ei[q] != k && ej[q] != k) {
                    const double rd = 1.0 / A[ek[q] + kk];
                    A[eo[q]] -= A[ek[q] + packed(ei[q], k)] * (A[ek[q] + packed(ej[q], k)] * rd);
                }
            }
            __syncthreads();
            if (tid < nb * nblk) {
                double *Ab = A + myblk * bstride;
                if (myi != k) Ab[packed(myi, k)] *= rdm;
                else Ab[kk] = -rdm;
            }
            __syncthreads();
        }
    };
    // interface block (base part)
    assemble(NSEG * (D::JP + D::JC), D::SP, S);
    __syncthreads();
    if (isVar && ipos >= nJ) {
        const int a = ipos - nJ;
        S[packed(a, a)] += v_diag;
        if (!isT) S[packed(nI - 1, a)] += v_ha;
    }
    __syncthreads();
    // register-resident factor blocks of group A
    const int Q = tid >> 2, part = tid & 3, seg = Q / L::QPS, lp = Q % L::QPS;
    const bool isG = grpA && lp < 25;
    double m1[2][14];   // rows (2lp, 2lp+1) of [G_s ; E_s^T], columns part*14 .. +13 (of 49, zero padded)
    double e3[2][8];    // G quads: rows (2lp, 2lp+1) of E_s, columns part*8 .. +7 (of 29)
#pragma unroll
    for (int a = 0; a < 2; a++) {
#pragma unroll
        for (int j = 0; j < 14; j++) m1[a][j] = 0.0;
#pragma unroll
        for (int j = 0; j < 8; j++) e3[a][j] = 0.0;
    }
    for (int s0 = 0; s0 < NSEG; s0 += L::HS) {
        const int nh = (NSEG - s0 < L::HS) ? NSEG - s0 : L::HS;
        for (int h = 0; h < nh; h++) {
            assemble((s0 + h) * (D::JP + D::JC), D::JP, KJJ + h * D::JP);
            assemble((s0 + h) * (D::JP + D::JC) + D::JP, D::JC, KJC + h * D::JC);
        }
        __syncthreads();
        if (isVar && ipos < nJ && ipos / 49 >= s0 && ipos / 49 < s0 + nh) {
            const int h = ipos / 49 - s0, li = ipos % 49;
            KJJ[h * D::JP + packed(li, li)] += v_diag;
            KJC[h * D::JC + li * 29 + 28] += v_ha;
        }
        __syncthreads();
        sweep(KJJ, 49, D::JP, nh, D::JP);           // KJJ <- -G
        // E_h = G K_JC
        for (int e = tid; e < nh * D::JC; e += NT) {
            const int h = e / D::JC, i = (e % D::JC) / 29, cc = e % 29;
            const double *Gn = KJJ + h * D::JP, *Kc = KJC + h * D::JC + cc;
            double acc = 0.0;
            for (int j = 0; j < 49; j++) acc -= Gn[packed(i, j)] * Kc[j * 29];
            Eh[e] = acc;
        }
        __syncthreads();
        // S -= K_CJ E on the lower triangle of each 29x29 coupled block
        for (int h = 0; h < nh; h++) {
            const int s = s0 + h;
            for (int e = tid; e < 29 * 30 / 2; e += NT) {
                int ca, cb;
                tri_decode(e, ca, cb);
                double acc = 0.0;
                for (int i = 0; i < 49; i++) acc += KJC[h * D::JC + i * 29 + ca] * Eh[h * D::JC + i * 29 + cb];
                const int ia = ca < 28 ? 14 * s + ca : nI - 1, ib = cb < 28 ? 14 * s + cb : nI - 1;
                S[packed(ia, ib)] -= acc;
            }
            __syncthreads();
        }
        // owners load their register blocks
        if (grpA && seg >= s0 && seg < s0 + nh) {
            const int h = seg - s0;
            const double *Gn = KJJ + h * D::JP, *Es = Eh + h * D::JC;
#pragma unroll
            for (int a = 0; a < 2; a++) {
                if (isG) {
                    const int row = 2 * lp + a;
#pragma unroll
                    for (int j = 0; j < 14; j++) {
                        const int col = part * 14 + j;
                        m1[a][j] = (row < 49 && col < 49) ? -Gn[packed(row, col)] : 0.0;
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int col = part * 8 + j;
                        e3[a][j] = (row < 49 && col < 29) ? Es[row * 29 + col] : 0.0;
                    }
                } else {
                    const int cc = 2 * (lp - 25) + a;       // row cc of E_s^T
#pragma unroll
                    for (int j = 0; j < 14; j++) {
                        const int i = part * 14 + j;
                        m1[a][j] = (cc < 29 && i < 49) ? Es[i * 29 + cc] : 0.0;
                    }
                }
            }
        }
        __syncthreads();
    }
    STAMP(1);
    sweep(S, nI, D::SP, 1, D::SP);                 // S <- -(S^-1)
    STAMP(2);
    {
        const int any = __syncthreads_or(status);
        if (tid == 0 && any) ws.status[b] |= any;
    }
#ifdef MPCMP_STAMPS
    unsigned long long *dbg = ws.dbg + (size_t)b * 16;
    if (tid == 0) { for (int k = 0; k < 3; k++) dbg[k] = stamp_acc[k]; }
#else
    unsigned long long *dbg = nullptr;
#endif
    if (grpA) {
        qp2_group_a<NSEG>(c, m1, e3, dbg);
    } else {
        // rows (2rp2, 2rp2+1) of S^-1, columns part2*10 .. +9; per column the LDS slots of the E^T b contributions
        double s2[2][10];
        int rof[10];
        const int rp2 = u >> 3, part2 = u & 7;
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const int col = part2 * 10 + j;
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const int row = 2 * rp2 + a;
                s2[a][j] = (row < nI && col < nI) ? -S[packed(row, col)] : 0.0;
            }
            int o1 = L::oMisc, o2 = L::oMisc;           // zero slot
            if (col < 14 * (NSEG + 1)) {
                const int sb = col / 14, cc = col % 14;
                if (sb < NSEG) o1 = L::oPart + sb * 32 + cc;
                if (sb > 0) o2 = L::oPart + (sb - 1) * 32 + 14 + cc;
            }
            rof[j] = (o1 << 16) | o2;
        }
        qp2_group_b<NSEG>(c, s2, rof);
    }
}

}  // namespace mpcmp
