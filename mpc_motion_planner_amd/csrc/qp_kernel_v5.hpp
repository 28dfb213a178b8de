// qp_kernel_v5.hpp — k_qp5<NSEG>: the ADMM loop of the N = 19 QP (the reference as shipped, robot_ocp.hpp:31-32) with EVERY dense
// factor of the solve resident in registers, loaded once.
//
// Same arithmetic as k_qp3 (T bordered out, one arm; structure3.hpp) with k_qp2's solve and k_qp2's row / variable scheme: the
// interior solve uses E_s = G_s K_JC,s (formed by k_qp3f on the matrix cores and handed over, LAY = 5) instead of a second product
// with G_s,
//     t = G b_J;   r_I = b_I - K_CJ t;   y_I = S^-1 r_I;   x_J = t - E y_C,
// so one solve is three barrier phases, each ONE product on a register-resident block.  768 threads = 12 waves, three per SIMD,
// 168 registers per lane; the factor (G 6 x 49 x 49, E 6 x 49 x 28, S^-1 98 x 98 = 32 k doubles, 35 k with padding) takes 104 - 112
// registers of every lane:
//     role G  waves 0..5, wave = segment: 4 x 13 block of G_s per lane (quad = four rows; lanes 56..62 of wave 5: G_u)   P1
//     role E  lanes 384..567: 4 x 14 block of E_s per lane (two lanes = four rows; "segment" 6 = E_u)                     P3
//     role S  lanes 568..767: 4 x 13 block of S^-1 per lane (eight lanes = four rows)                                     P2
// E^T is not kept (it does not fit): K_CJ t is taken by the G waves from their own t (K_CJ is sparse plus one dense 7 x 14 block per
// segment), inside the wave, as in k_qp3.  The rows and variables of the ADMM iteration are spread as in k_qp2: a variable and a
// dynamics row per lane on the G waves (the last variables and T on wave 6), the path rows on sixteen lanes per node on waves 7..11
// (four lanes per pair of rows, six columns each; the node's path-row part of A^T w is formed by the same lanes).  Five workgroup
// barriers per ADMM iteration:
//     A   rhs = sigma x - q + rho_b z_b - y_b + A^T w                      (variable lanes)
//     P1  t = G b_J, K_CJ t; x~_T of the T border                          (role G; one idle wave sums the border's partial sums)
//     P2  r_I = b_I - K_CJ t, y_I = S^-1 r_I, interface rows of x~         (role S)
//     P3  x_J = t - E y_C - w x~_T                                         (role E)
//     E   z~ = A x~, relaxation, projection, dual update                   (every lane: its row and / or its variable)
// The factor blocks are loaded ONCE (k_qp3 re-reads its blocks at every termination test: 2 GB per launch).
#pragma once
#include "qp_kernel_v3.hpp"

namespace mpcmp {

template <int NSEG>
struct Qp5 : Qp3<NSEG> {
    using Q3 = Qp3<NSEG>;
    using D = Dim3<NSEG>;
    using F = Qp5Fac<NSEG>;
    static constexpr int NT = 768, NWV = 12;
    static constexpr int NG = 64 * NSEG;                              // role G threads (wave = segment)
    static constexpr int NSL = F::NSL;                               // S lanes
    static constexpr int tS0 = NT - NSL;                             // first S lane (568)
    static constexpr int wS0 = tS0 / 64;                             // first wave with S lanes (8: mixed with E lanes)
    static constexpr int NE = tS0 - NG;                              // lanes of the E role incl. spares
    static constexpr int NEL = F::NEL;
    static_assert(NEL <= NE && NE <= F::ELS && NG < tS0, "role map");
    static constexpr int tP0 = NG + 64;                              // first path-row lane (448): sixteen lanes per node
    static constexpr int NPL = NT - tP0;                             // lanes of the path-row tables (320)
    static constexpr int tV0 = tP0 + 16 * D::N;                      // behind the path rows: the variables that have no lane on the G waves, then T (752..767)
    static constexpr int NVL = NG + (NT - tV0);                      // variable lanes: vi = tid on the G waves, NG + (tid - tV0) behind the path rows (400)
    static_assert(tV0 % 16 == 0 && tV0 / 64 == NWV - 1 && D::na == NVL - 1 && D::meq <= NG, "row / variable map");
    static constexpr int tidT = NT - 1;                              // the lane that carries the ADMM state of T (vi = na)
    static constexpr int GS = 24;                                    // row stride of the path Jacobians (22 + 2 zero pads: 16-byte reads of six columns)
    static constexpr int XS = 24;                                    // node stride of x~, w of the border, gp: [x_k (14) | u_k (7) | T | pad pad]
    static constexpr int NX = XS * D::N, NXP = NX + 8;               // (slots NX, NX + 1: pads for lanes without an output)
    static constexpr int JS = Q3::JS, RIW = Q3::RIW, NAP = Q3::NAP, MAP = Q3::MAP, TS = Q3::TS;
    static constexpr int e2(int x) { return (x + 1) / 2 * 2; }
    // common part (QP3_PROLOGUE_L): the path Jacobians with this kernel's stride
    static constexpr int oGk = 0;
    static constexpr int oMisc = oGk + D::N * 8 * GS;                // [32]
    static constexpr int oCD = oMisc + 32;                           // [16] differentiation matrix, [16] zeros
    static constexpr int oRedP = oCD + 32;                           // [160] workgroup reductions
    static constexpr int oCfg = oRedP + 160;                         // [64]
    static constexpr int oPat = oCfg + 64;                           // [54]
    static constexpr int oPE = oPat + 54;
    // LDS of the loop kernel behind the common part (doubles)
    static constexpr int vKT = oPE;                                  // [NAP] T column k (internal order), kappa
    static constexpr int vKCJ = vKT + NAP;                           // [NSEG][8][28] column form of the sparse K_JC (as k_qp3)
    static constexpr int vKUXT = vKCJ + NSEG * 224;                  // [NSEG + 1][14][8] dense blocks transposed
    static constexpr int vZR = vKUXT + (NSEG + 1) * 112;             // [16] zeros
    static constexpr int vDT = vZR + 16;                             // [5][3] columns of the differentiation matrix (column 4: zeros), [1] pad
    static constexpr int vVc = vDT + 16;                             // [6][NVL] variable constants: cf, lb, ub, ha, rho_b, 1 / rho_b
    static constexpr int vPc = vVc + 6 * NVL;                       // [4][NPL] path rows: lg, ug, rho, 1 / rho
    static constexpr int vRc = vPc + 4 * NPL;                        // [2][NG] dynamics rows: l = u = -c_eq, T coefficient -ts f
    static constexpr int vZ0 = vRc + 2 * NG;                         // ---- zero-initialised from here ----
    static constexpr int vRhsJ = vZ0;                                // [NSEG][JS]
    static constexpr int vRhsU = vRhsJ + NSEG * JS;                  // [JS]
    static constexpr int vRhsI = vRhsU + JS;                         // [RIW]
    static constexpr int vTJ = vRhsI + RIW;                          // [NSEG][TS] t of every segment (rows 49..51: zero), then [TS] of the U block
    static constexpr int vTU = vTJ + NSEG * TS;
    static constexpr int vPA = vTU + TS, vPB = vPA + RIW, vDP = vPB + RIW, vPD = vDP + RIW;      // K_CJ t in three parts (k_qp3), pad [2]
    static constexpr int vRIw = vPD + 2;                             // [RIW]
    static constexpr int vYI = vRIw + RIW;                           // [RIW] y_I (zero beyond nI, slot YPAD: pad)
    static constexpr int YPAD = 120;
    static_assert(14 * (NSEG + 2) <= YPAD && D::nI + 8 <= RIW, "operand ranges of the E lanes");
    static constexpr int vXn = vYI + RIW;                            // [NXP] x~ node-major
    static constexpr int vWv = vXn + NXP;                            // [NXP] w of the T border, node-major (at a fixed distance from x~)
    static constexpr int vXx = vWv + NXP;                            // [NXP] x node-major (termination tests)
    static constexpr int vGp = vXx + NXP;                            // [NXP] path-row part of A^T w, node-major
    static constexpr int vGpy = vGp + NXP;                           // [NXP] path-row part of A^T y (termination tests)
    static constexpr int vWg = vGpy + NXP;                           // [meq + 2] w = rho z - y of the dynamics rows (slot meq + 1: write-only pad)
    static constexpr int vYs = vWg + e2(D::meq + 2);                 // [meq + 2] y of the dynamics rows (termination tests)
    static constexpr int vRedB = vYs + e2(D::meq + 2);               // [64] partial sums of w^T rhs per 8 lanes
    static constexpr int vRedT = vRedB + 64;                         // [64] partial sums of the T column of A^T w (dynamics rows)
    static constexpr int vPadW = vRedT + 64;                         // [8] write-only pad (16-byte stores)
    static constexpr int vVst = vPadW + 8;                           // [3][16] ADMM state x, z_b, y_b of the variables behind the path rows
    static constexpr int vZ1 = vVst + 48;                            // ---- to here ----
    static constexpr int NF = 8;                                     // lane-constant table: [NF][NT] 32-bit words (q5_lc)
    static constexpr int vLCT = vZ1;
    static constexpr int size5 = vLCT + NF * NT / 2;
    static_assert(size5 * 8 <= 160 * 1024 - 512, "LDS budget");
    static_assert(vXn % 2 == 0 && vGp % 2 == 0 && vGpy % 2 == 0 && vXx % 2 == 0 && vPadW % 2 == 0 && oGk % 2 == 0, "16-byte accesses");
};

// 4 x 13 block of S^-1, eight lanes per group of four rows: quad reduce-scatter (g_blk), then the group's two quads are added
// (operand reads in two batches: the lane's register budget)
__device__ __forceinline__ double s_blk8(const double (&m)[52], const double *op) {
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
    {
        double o[7];
#pragma unroll
        for (int j = 0; j < 7; j++) o[j] = ldv(op + j);
#pragma unroll
        for (int j = 0; j < 7; j++) { p0 += m[j] * o[j]; p1 += m[13 + j] * o[j]; p2 += m[26 + j] * o[j]; p3 += m[39 + j] * o[j]; }
    }
    __builtin_amdgcn_sched_barrier(0);          // (the second batch is not fetched ahead of the first batch's products)
    {
        double o[6];
#pragma unroll
        for (int j = 0; j < 6; j++) o[j] = ldv(op + 7 + j);
#pragma unroll
        for (int j = 0; j < 6; j++) { p0 += m[7 + j] * o[j]; p1 += m[20 + j] * o[j]; p2 += m[33 + j] * o[j]; p3 += m[46 + j] * o[j]; }
    }
    const double q0 = p0 + dpp_mov<0x4E>(p2), q1 = p1 + dpp_mov<0x4E>(p3);
    const double x = q0 + dpp_mov<0xB1>(q1);
    return x + dpp_xor4(x);
}

// Workgroup reduction (sum or max) of K values per thread for the termination tests, with a small register footprint (the lanes carry
// their factor blocks): DPP inside the waves, one LDS slot per (wave, k), then the sixteen lanes of a DPP row combine the NW partials
// of one k (k_qp2's scheme); the result is valid in every thread.  Two barriers; `red` (>= (NW + 1) K doubles) must not be shared
// with a reduction issued right before or after.
template <int NW, int K, bool MAX>
__device__ __forceinline__ void q5_reduce(double (&v)[K], double *red, int tid) {
    static_assert(NW <= 16 && 16 * K <= 64 * NW, "one DPP row per value");
#pragma unroll
    for (int k = 0; k < K; k++) {
        double x = v[k];
        if (MAX) {
            x = fmax(x, dpp_mov<0xB1>(x)); x = fmax(x, dpp_mov<0x4E>(x)); x = fmax(x, dpp_mov<0x141>(x)); x = fmax(x, dpp_mov<0x140>(x));
            x = fmax(fmax(x, read_lane(x, 16)), fmax(read_lane(x, 32), read_lane(x, 48)));
        } else x = wave_sum(x);
        if ((tid & 63) == 0) red[(tid >> 6) * K + k] = x;
    }
    __syncthreads();
    if (tid < 16 * K) {
        const int w = tid & 15, k = tid >> 4;
        double a = w < NW ? red[w * K + k] : 0.0;             // (0: identity of both reductions, the maxima are of magnitudes)
        if (MAX) { a = fmax(a, dpp_mov<0xB1>(a)); a = fmax(a, dpp_mov<0x4E>(a)); a = fmax(a, dpp_mov<0x141>(a)); a = fmax(a, dpp_mov<0x140>(a)); }
        else { a += dpp_mov<0xB1>(a); a += dpp_mov<0x4E>(a); a += dpp_mov<0x141>(a); a += dpp_mov<0x140>(a); }
        if (w == 0) red[NW * K + k] = a;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = red[NW * K + k];
}

template <int NSEG>
struct Qp5Ctx {
    const mpcmp_config *cfg;
    WS ws;
    double *lds;
    const double *fa;
    int tid, b;
    double tsT, rho_in, rho_eq, sigma, alpha;
};

#ifdef MPCMP_STAMPS
#define Q5_STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_busy[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = clock64(), st_m = 0; (void)st_m
#define Q5_STAMP_RESET do { for (int k_ = 0; k_ < 8; k_++) st_acc[k_] = st_busy[k_] = 0; st_t = clock64(); } while (0)
#define Q5_STAMP_DUMP(it_) do { if ((c.tid & 63) == 0) { unsigned long long *o_ = c.ws.dbg + (size_t)c.b * MPCMP_DBG_WORDS; \
        for (int k_ = 0; k_ < 8; k_++) o_[16 + (c.tid >> 6) * 8 + k_] = st_busy[k_]; \
        if (c.tid == 0) { for (int k_ = 0; k_ < 7; k_++) o_[k_] = st_acc[k_]; o_[15] = (it_); } } } while (0)
#else
#define Q5_STAMP_DECL do { } while (0)
#define Q5_STAMP_RESET do { } while (0)
#define Q5_STAMP_DUMP(it_) do { } while (0)
#endif

// Lane-constant table [NF][NT] of 32-bit words in LDS: every per-lane loop-invariant integer of the ADMM loop (LDS addresses in doubles,
// two per word) lives here and is fetched right in front of the barrier that opens the phase that uses it, so that no such value
// occupies a register across the loop (the factor block leaves a lane ~ 55 registers for everything else).  Fields:
//   role G : 0..3 the solve (t slot | operand block;  K_CJ column | t + c % 14;  part slot | dense coefficients;  dense operand | dense slot),
//            4 pxr (x~ slot | rhs slot), 5 prf (row of the f term | first row of the own segment's column),
//            6 prb (first row of the previous segment's column | bit 16: rho_b = rho_eq | bits 17..18: row of D of the lane's dynamics row),
//            7 ixr (dynamics row: first operand slot in x~ | slot of its f operand)
//   S lanes: 0 operand block of S^-1 r_I | y_I slot, 1 x~ slot of the interface row;   E lanes: 0 y_C operand | t slot, 1 x~ slots of the two rows
//   path-row lanes: 2 gro (Jacobian operand | bit 16 parity | bit 17 publishes gp), 3 xno (x~ operand);   variable lanes behind them: 4..6 as role G
template <int NSEG>
__device__ __forceinline__ int q5_lc(const Qp5Ctx<NSEG> &c, int t, int f) {
    using L = Qp5<NSEG>;
    const int *lct = reinterpret_cast<const int *>(c.lds + L::vLCT);
    return *(const volatile __attribute__((address_space(3))) int *)(lct + f * L::NT + t);
}
__device__ __forceinline__ int lo16(int w) { return w & 0xFFFF; }
__device__ __forceinline__ int hi16(int w) { return (int)((unsigned)w >> 16); }

// ---- the three products of one solve with K_0 ----
// P1 (role G): t = G b_J (k_qp3's block product), then inside the wave part = K_CJ t
template <int NSEG>
__device__ __forceinline__ void q5_p1(const Qp5Ctx<NSEG> &c, const double (&fm)[52], int k0, int k1, int k2, int k3) {
    double *lds = c.lds;
    lds[lo16(k0)] = g_blk<false>(fm, lds + hi16(k0));
    wave_sync();
    const double *kc = lds + lo16(k1), *tc = lds + hi16(k1);
    double kq[7], tv[7];
#pragma unroll
    for (int d = 0; d < 7; d++) { kq[d] = ldv(kc + 28 * d); tv[d] = ldv(tc + 7 * (d - 1)); }      // rows c % 14 + 7 (d - 1) of the segment
    const double acc = ((kq[0] * tv[0] + kq[1] * tv[1]) + (kq[2] * tv[2] + kq[3] * tv[3])) + ((kq[4] * tv[4] + kq[5] * tv[5]) + kq[6] * tv[6]);
    lds[lo16(k2)] = acc;
    // dense blocks (K_XU t of the columns x_3s; for the last segment also the U block): half a column per lane
    double dk[4], dt[4];
#pragma unroll
    for (int d = 0; d < 4; d++) { dk[d] = ldv(lds + hi16(k2) + d); dt[d] = ldv(lds + lo16(k3) + d); }
    const double ad = (dk[0] * dt[0] + dk[1] * dt[1]) + (dk[2] * dt[2] + dk[3] * dt[3]);
    lds[hi16(k3)] = ad + dpp_mov<0xB1>(ad);                                   // (odd lanes, lanes without a column: pad slot)
}
// P1 (one idle wave): x~_T = (b_T - w^T b) / delta of the bordered solve; b_T = base + (T column of A^T w)
template <int NSEG>
__device__ __forceinline__ void q5_p1_xT(const Qp5Ctx<NSEG> &c, int t) {
    using L = Qp5<NSEG>;
    double *lds = c.lds;
    const int lane = t & 63;
    const double s = (ldv(lds + L::vRedT + lane) - ldv(lds + L::vRedB + lane)) + ldv(lds + L::vGp + (lane < L::D::N ? lane * L::XS + 21 : L::NX));
    const double sa = wave_sum(s);
    if (lane == 0) lds[L::oMisc + L::M_xtT] = (lds[L::oMisc + L::M_baseT] + sa) / lds[L::oMisc + L::M_delta];
}
// P2 (waves with S lanes): r_I = b_I - part (every wave its own, identical copy), y_I = S^-1 r_I, interface rows of x~
template <int NSEG>
__device__ __forceinline__ void q5_p2(const Qp5Ctx<NSEG> &c, const double (&fm)[52], int t, int k0, int k1, const bool laneS, const bool use_xT) {
    using L = Qp5<NSEG>;
    double *lds = c.lds;
    double *rIw = lds + L::vRIw;
    {
        const int l8 = t & 63;
        const double *src = lds + L::vRhsI + l8;
        double rv[8];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            rv[4 * u] = ldv(src + 64 * u); rv[4 * u + 1] = ldv(src + (L::vPA - L::vRhsI) + 64 * u);
            rv[4 * u + 2] = ldv(src + (L::vPB - L::vRhsI) + 64 * u); rv[4 * u + 3] = ldv(src + (L::vDP - L::vRhsI) + 64 * u);
        }
        rIw[l8] = ((rv[0] - rv[1]) - rv[2]) - rv[3];
        rIw[l8 + 64] = ((rv[4] - rv[5]) - rv[6]) - rv[7];
    }
    const int xds = laneS ? lo16(k1) : L::vXn + L::NX, ysl = laneS ? hi16(k0) : L::vYI + L::YPAD;      // (the E lanes of the mixed wave: pad slots)
    const double wds = ldv(lds + xds + (L::vWv - L::vXn));
    const double xT = use_xT ? ldv(lds + L::oMisc + L::M_xtT) : 0.0;
    wave_sync();
    const double yi = s_blk8(fm, lds + (laneS ? lo16(k0) : L::vRIw));
    lds[ysl] = yi;                                                            // (lanes without an output row: pad slots)
    lds[xds] = yi - wds * xT;
}
// P3 (waves with E lanes): x_J = t - E y_C - w x~_T
template <int NSEG>
__device__ __forceinline__ void q5_p3(const Qp5Ctx<NSEG> &c, const double (&fm)[56], int k0, int k1, const bool laneE, const bool use_xT) {
    using L = Qp5<NSEG>;
    double *lds = c.lds;
    constexpr int dW = L::vWv - L::vXn;
    const int xd0 = laneE ? lo16(k1) : L::vXn + L::NX, xd1 = laneE ? hi16(k1) : L::vXn + L::NX + 1;      // (the S lanes of the mixed wave: pad slots)
    const double *yc = lds + (laneE ? lo16(k0) : L::vYI), *ts = lds + (laneE ? hi16(k0) : L::vTU + 8);
    const double t0 = ldv(ts), t1 = ldv(ts + 1);
    const double w0 = ldv(lds + xd0 + dW), w1 = ldv(lds + xd1 + dW);
    const double xT = use_xT ? ldv(lds + L::oMisc + L::M_xtT) : 0.0;
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
    for (int j0 = 0; j0 < 14; j0 += 7) {            // (two batches of operand reads: register budget)
        double o[7];
#pragma unroll
        for (int j = 0; j < 7; j++) o[j] = ldv(yc + j0 + j);
#pragma unroll
        for (int j = 0; j < 7; j++) { p0 += fm[j0 + j] * o[j]; p1 += fm[14 + j0 + j] * o[j]; p2 += fm[28 + j0 + j] * o[j]; p3 += fm[42 + j0 + j] * o[j]; }
        __builtin_amdgcn_sched_barrier(0);
    }
    const double e0 = p0 + dpp_mov<0xB1>(p2), e1 = p1 + dpp_mov<0xB1>(p3);      // rows 4 g + 2 h, + 1: the own half + the partner's
    lds[xd0] = (t0 - e0) - w0 * xT;                                            // (rows 49..51, spare lanes: pad)
    lds[xd1] = (t1 - e1) - w1 * xT;
}
// P3 (one idle wave): x~_T, once per node: the 22nd operand of the path rows
template <int NSEG>
__device__ __forceinline__ void q5_p3_xT(const Qp5Ctx<NSEG> &c, int t, const bool use_xT) {
    using L = Qp5<NSEG>;
    const int lane = t & 63;
    const double xT = use_xT ? ldv(c.lds + L::oMisc + L::M_xtT) : 0.0;
    if (lane < L::D::N) c.lds[L::vXn + L::XS * lane + 21] = xT;
}

// after the solve of K_0 w = k: w and delta of the T border (every thread; three barriers)
template <int NSEG>
__device__ __forceinline__ void q5_finish_border(const Qp5Ctx<NSEG> &c) {
    using L = Qp5<NSEG>;
    using D = Dim3<NSEG>;
    double *lds = c.lds, *misc = lds + L::oMisc;
    const int tid = c.tid;
    auto node_slot = [&](int v) -> int { return v < 14 * D::N ? L::XS * (v / 14) + v % 14 : L::XS * ((v - 14 * D::N) / 7) + 14 + (v - 14 * D::N) % 7; };
    double sacc = 0.0;
    for (int v = tid; v < D::na; v += L::NT) sacc += lds[L::vKT + int3_of_ext(NSEG, v)] * lds[L::vXn + node_slot(v)];
    double sv[1] = {sacc};
    block_reduce<L::NWV, 1, false>(sv, lds + L::oRedP, tid);
    for (int v = tid; v < D::na; v += L::NT) lds[L::vWv + node_slot(v)] = lds[L::vXn + node_slot(v)];
    if (tid == 0) {
        const double hdT = misc[L::M_sumha] + c.cfg->hess_reg;
        misc[L::M_hdT] = hdT;
        misc[L::M_delta] = (lds[L::vKT + D::na] + (hdT + c.sigma + misc[L::M_rbT])) - sv[0];
        misc[L::M_baseT] = -1.0;                              // sigma x_T - q_T + rho_T z_T - y_T with x = z = y = 0, q_T = 1 (cost = T)
        if (!(misc[L::M_delta] > 0.0)) atomicOr(&c.ws.status[c.b], 2);
    }
    __syncthreads();
}

// termination test, common tail (every thread): combine the maxima, add the row / column of T, decide.  Two barriers.
template <int NSEG>
__device__ __forceinline__ int q5_check_tail(const Qp5Ctx<NSEG> &c, const double (&sums)[2], double (&mx)[6]) {
    using L = Qp5<NSEG>;
    const double *misc = c.lds + L::oMisc;
    q5_reduce<L::NWV, 6, true>(mx, c.lds + L::oRedP + 32, c.tid);
    const double xTv = misc[L::M_xT], zT = misc[L::M_zbT], yT = misc[L::M_ybT];
    const double hxT = misc[L::M_hdT] * xTv + sums[1], atyT = sums[0] + yT;
    mx[0] = fmax(mx[0], fabs(xTv - zT)); mx[1] = fmax(mx[1], fabs(xTv)); mx[2] = fmax(mx[2], fabs(zT));
    mx[3] = fmax(mx[3], fabs(hxT + atyT + 1.0)); mx[4] = fmax(mx[4], fabs(hxT)); mx[5] = fmax(mx[5], fabs(atyT));
    const double ep = c.cfg->eps_abs + c.cfg->eps_rel * fmax(mx[1], mx[2]);
    const double ed = c.cfg->eps_abs + c.cfg->eps_rel * fmax(fmax(mx[4], mx[5]), 1.0);      // ||q||_inf = 1
    return (mx[0] <= ep && mx[3] <= ed) ? 1 : 0;
}

// ---- a variable lane (k_qp2's role B): ADMM state of one variable, its constants lane-transposed in LDS ----
// (A^T w)[v] without the T row: the path-row part comes from the node's gp, the dynamics rows are gathered (coefficient 0 where there is none)
template <int NSEG>
__device__ __forceinline__ double q5_col_gather(const double *lds, int pxr, int prf, int prb, const double *vcl, const double *w, const double *gp) {
    using L = Qp5<NSEG>;
    const int xpos = lo16(pxr), rf = lo16(prf), rA = hi16(prf), rB = lo16(prb);
    const double *cA = lds + L::vDT + 3 * ((prb >> 19) & 7), *cB = lds + L::vDT + 3 * ((prb >> 22) & 7);      // columns of D (4: none)
    double wv[7], cv[7];
    const double s0 = ldv(gp + xpos);
    wv[0] = ldv(w + rf);
#pragma unroll
    for (int i = 0; i < 3; i++) { wv[1 + i] = ldv(w + rA + 14 * i); wv[4 + i] = ldv(w + rB + 14 * i); }
    cv[0] = ldv(vcl);
#pragma unroll
    for (int i = 0; i < 3; i++) { cv[1 + i] = ldv(cA + i); cv[4 + i] = ldv(cB + i); }
    double s = s0;
#pragma unroll
    for (int i = 0; i < 7; i++) s += cv[i] * wv[i];
    return s;
}

// ---- role G: waves 0 .. NSEG - 1.  P1; a variable per lane, a dynamics row per lane of the first meq lanes ----
template <int NSEG>
__device__ __forceinline__ void qp5_role_g(const Qp5Ctx<NSEG> &c) {
    using L = Qp5<NSEG>;
    using D = Dim3<NSEG>;
    constexpr int meq = D::meq, XS = L::XS;
    double *lds = c.lds;
    const mpcmp_config &cfg = *c.cfg;
    const int tid = c.tid, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    Q5_STAMP_DECL;
    double fm[52];
    {
        typedef const __attribute__((address_space(1))) double *gptr_t;
        gptr_t p = (gptr_t)(c.fa + L::oFG + (size_t)(wave * 52) * 64 + lane);
#pragma unroll
        for (int j = 0; j < 52; j++) fm[j] = p[j * 64];
    }
    // ADMM state of the lane's variable and of its dynamics row (z of an equality row is its bound l from the first update on)
    double vx = 0.0, vzb = 0.0, vyb = 0.0, ygd = 0.0;
    const bool isDyn = tid < meq;
    const bool waveDyn = (tid & ~63) < meq;
    const double mtsT = -c.tsT, alpha = c.alpha, sigma = c.sigma, rho_eq = c.rho_eq;
    auto row_dot_dyn = [&](const double *xe, int ixo, int cro, double rcT) -> double {
        const int ix0 = lo16(ixo), ixf = hi16(ixo);
        const double c0 = ldv(lds + cro), c1 = ldv(lds + cro + 1), c2 = ldv(lds + cro + 2), c3 = ldv(lds + cro + 3);
        const double x0 = ldv(xe + ix0), x1 = ldv(xe + ix0 + XS), x2 = ldv(xe + ix0 + 2 * XS), x3 = ldv(xe + ix0 + 3 * XS), xf = ldv(xe + ixf), xT = ldv(xe + 21);
        return ((c0 * x0 + c1 * x1) + (c2 * x2 + c3 * x3)) + (mtsT * xf + rcT * xT);
    };
    // ---- K_0 w = k (the T border) ----
    {
        int t = tid;
        asm volatile("" : "+v"(t));
        q5_p1<NSEG>(c, fm, q5_lc<NSEG>(c, t, 0), q5_lc<NSEG>(c, t, 1), q5_lc<NSEG>(c, t, 2), q5_lc<NSEG>(c, t, 3));
    }
    __syncthreads();
    __syncthreads();
    __syncthreads();
    q5_finish_border<NSEG>(c);
    Q5_STAMP_RESET;
    int it = 0, done = 0, until_check = cfg.check_every;
    int apx, apf, apb;          // constants of phase A, fetched ahead of the barrier that ends the previous iteration
    {
        int t = tid;
        asm volatile("" : "+v"(t));
        apx = q5_lc<NSEG>(c, t, 4); apf = q5_lc<NSEG>(c, t, 5); apb = q5_lc<NSEG>(c, t, 6);
    }
    for (it = 1; it <= cfg.qp_iters; it++) {
        int sio = tid;
        asm volatile("" : "+v"(sio));
        // ---- A: rhs = sigma x - q + rho_b z_b - y_b + A^T w; partial sums of w^T rhs ----
        {
            const double rb = ldv(lds + L::vVc + sio + 4 * L::NVL);
            const double wb = ldv(lds + L::vWv + lo16(apx));
            const double r = (sigma * vx + (rb * vzb - vyb)) + q5_col_gather<NSEG>(lds, apx, apf, apb, lds + L::vVc + sio, lds + L::vWg, lds + L::vGp);
            lds[hi16(apx)] = r;
            const double bp = sum8(wb * r);
            lds[L::vRedB + (sio >> 3)] = bp;              // (all eight lanes of a group hold the sum and store it)
        }
        const int k0 = q5_lc<NSEG>(c, sio, 0), k1 = q5_lc<NSEG>(c, sio, 1), k2 = q5_lc<NSEG>(c, sio, 2), k3 = q5_lc<NSEG>(c, sio, 3);
        QB(0); __syncthreads(); QS(0);
        q5_p1<NSEG>(c, fm, k0, k1, k2, k3);
        QB(1); __syncthreads(); QS(1);
        // ---- P2 (role S) ----
        QB(2); __syncthreads(); QS(2);
        // ---- P3 (role E); this role is idle: the constants of phase E ----
        const int epx = q5_lc<NSEG>(c, sio, 4), epb = q5_lc<NSEG>(c, sio, 6), eix = q5_lc<NSEG>(c, sio, 7);
        QB(3); __syncthreads(); QS(3);
        // ---- E: the variable and the dynamics row of the lane ----
        const bool check = (--until_check == 0);
        if (check) until_check = cfg.check_every;
        {
            const double *vcl = lds + L::vVc + sio, *rcl = lds + L::vRc + sio;
            const int xpos = lo16(epx);
            const double xtv = ldv(lds + L::vXn + xpos), vlb = ldv(vcl + 1 * L::NVL), vub = ldv(vcl + 2 * L::NVL);
            const double rb = ldv(vcl + 4 * L::NVL), rbi = ldv(vcl + 5 * L::NVL);
            if (waveDyn) {
                // (lanes past the last row compute on row 0's operands and store into the pad)
                const double lgd = ldv(rcl), rcT = ldv(rcl + L::NG);
                const double zt = row_dot_dyn(lds + L::vXn, eix, L::oCD + 4 * ((epb >> 17) & 3), rcT);
                const double zr = alpha * zt + (1.0 - alpha) * (it > 1 ? lgd : 0.0);
                ygd += rho_eq * (zr - lgd);                 // the row is an equality: the projection of anything onto [l, l] is l
                const double w = rho_eq * lgd - ygd;
                lds[L::vWg + (isDyn ? sio : meq + 1)] = w;
                const double tp = sum8(isDyn ? rcT * w : 0.0);
                lds[L::vRedT + (sio >> 3)] = tp;
                if (check) lds[L::vYs + (isDyn ? sio : meq + 1)] = ygd;
            }
            vx = alpha * xtv + (1.0 - alpha) * vx;
            const double zrv = alpha * xtv + (1.0 - alpha) * vzb;
            const double znv = clip(zrv + vyb * rbi, vlb, vub);
            vyb += rb * (zrv - znv);
            vzb = znv;
            if (check) lds[L::vXx + xpos] = vx;
        }
        apx = q5_lc<NSEG>(c, sio, 4); apf = q5_lc<NSEG>(c, sio, 5); apb = q5_lc<NSEG>(c, sio, 6);
        QB(4); __syncthreads(); QS(4);
#ifndef Q5_NOTEST
        if (__builtin_expect(check, 0)) {
            int t = tid;
            asm volatile("" : "+v"(t));
            const double *vcl = lds + L::vVc + t, *rcl = lds + L::vRc + t, *xx = lds + L::vXx;
            const int pxc = q5_lc<NSEG>(c, t, 4), prfc = q5_lc<NSEG>(c, t, 5), prbc = q5_lc<NSEG>(c, t, 6), ixc = q5_lc<NSEG>(c, t, 7);
            const double ha = ldv(vcl + 3 * L::NVL);
            const double rcT = ldv(rcl + L::NG), zgd = ldv(rcl);
            double sums[2] = {isDyn ? rcT * ygd : 0.0, ha * vx};
            q5_reduce<L::NWV, 2, false>(sums, lds + L::oRedP, tid);      // (its barriers publish gpy)
            double mx[6] = {0, 0, 0, 0, 0, 0};
            if (isDyn) {
                const double ax = row_dot_dyn(xx, ixc, L::oCD + 4 * ((prbc >> 17) & 3), rcT);
                mx[0] = fabs(ax - zgd); mx[1] = fabs(ax); mx[2] = fabs(zgd);
            }
            {
                const double hx = (fabs(ha) + cfg.hess_reg) * vx + ha * xx[21], aty = q5_col_gather<NSEG>(lds, pxc, prfc, prbc, vcl, lds + L::vYs, lds + L::vGpy) + vyb;
                mx[0] = fmax(mx[0], fabs(vx - vzb)); mx[1] = fmax(mx[1], fabs(vx)); mx[2] = fmax(mx[2], fabs(vzb));
                mx[3] = fabs(hx + aty); mx[4] = fabs(hx); mx[5] = fabs(aty);
            }
            done = q5_check_tail<NSEG>(c, sums, mx);
            QS(5);
        }
#endif
        if (done) break;
    }
    const bool capped = it > cfg.qp_iters;
    if (capped) it = cfg.qp_iters;
    Q5_STAMP_DUMP(it);
    if (tid == 0) { c.ws.qpit[c.b] = it; c.ws.qp_total[c.b] += it; if (capped) atomicAdd(&c.ws.status[c.b], MPCMP_ST_CAP_ONE); }
    c.ws.p[(size_t)c.b * (D::na + 1) + tid] = vx;
    c.ws.y[(size_t)c.b * (D::ma + D::na + 1) + D::ma + tid] = vyb;
    if (isDyn) c.ws.y[(size_t)c.b * (D::ma + D::na + 1) + tid] = ygd;
}

// ---- the path rows (k_qp2's scheme): four lanes per pair of rows (six columns each), sixteen lanes per node; lanes 0, 1 of a quad own
// the rows 2 prp, 2 prp + 1.  z~ of the owned row and, from the same Jacobian operands, this node's path-row part of A^T w: every lane
// forms its six columns of g_row0 w0 + g_row1 w1, the four row pairs of the node (lane bits 2, 3 of the DPP row) are summed with two
// row rotations, and the lanes of pair 0 publish the node's 24 padded columns.  The Jacobian rows are read twice (volatile reads in
// program order: x~, the own row, the other row; after the row update both rows again): the lane's factor block leaves ~ 50 registers.
template <int NSEG, class F>
__device__ __forceinline__ double q5_path_rows(double *lds, int gro, int xno, const double *xe, double *gdst, F &&row_update) {
    using L = Qp5<NSEG>;
    const int par = (gro >> 16) & 1, first = (gro >> 17) & 1, go = lo16(gro);
    const double *g0 = lds + go + par * L::GS, *g1 = lds + go + (1 - par) * L::GS;     // the row of the lane's parity first (quad_sum2)
    const double *xv = xe + xno;
    v2d x2[3], pa[3], pb[3];
#pragma unroll
    for (int j = 0; j < 3; j++) x2[j] = ldv2(xv + 2 * j);
#pragma unroll
    for (int j = 0; j < 3; j++) pa[j] = ldv2(g0 + 2 * j);
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = 0; j < 3; j++) { a0 += pa[j].x * x2[j].x; a0 += pa[j].y * x2[j].y; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 3; j++) pb[j] = ldv2(g1 + 2 * j);
#pragma unroll
    for (int j = 0; j < 3; j++) { a1 += pb[j].x * x2[j].x; a1 += pb[j].y * x2[j].y; }
    const double ax = quad_sum2(a0, a1);
    const double wq = row_update(ax);                                   // owners: the row's multiplier-like value
    const double w0 = dpp_mov<0x44>(wq), w1 = dpp_mov<0x11>(wq);        // quad broadcasts: owner of the own row, of the other row
    double *dst = first ? gdst + xno : lds + L::vPadW;                  // (pairs 1..3 of a node: pad)
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const v2d qa = ldv2(g0 + 2 * j), qb = ldv2(g1 + 2 * j);
        double cx = qa.x * w0 + qb.x * w1, cy = qa.y * w0 + qb.y * w1;
        cx += dpp_mov<0x128>(cx); cy += dpp_mov<0x128>(cy);             // row_ror:8
        cx += dpp_mov<0x124>(cx); cy += dpp_mov<0x124>(cy);             // row_ror:4
        D2 o; o.x = cx; o.y = cy;
        *reinterpret_cast<D2 *>(dst + 2 * j) = o;
    }
    return ax;
}
// one ADMM update of the path rows of this lane's quad (zg, yg: state of the owned row)
template <int NSEG>
__device__ __forceinline__ void q5_path_E(const Qp5Ctx<NSEG> &c, double &zg, double &yg, int gro, int xno, int et, bool ownsRow) {
    using L = Qp5<NSEG>;
    double *lds = c.lds;
    const double *pcl = lds + L::vPc + et;
    const double alpha = c.alpha;
    q5_path_rows<NSEG>(lds, gro, xno, lds + L::vXn, lds + L::vGp, [&](double zt) -> double {
        const double lgp = ldv(pcl), ugp = ldv(pcl + L::NPL), rr = ldv(pcl + 2 * L::NPL), rri = ldv(pcl + 3 * L::NPL);
        double w = 0.0;
        if (ownsRow) {
            const double zr = alpha * zt + (1.0 - alpha) * zg;
            const double zn = clip(zr + yg * rri, lgp, ugp);
            yg += rr * (zr - zn);
            zg = zn;
            w = rr * zg - yg;
        }
        return w;
    });
}
// termination test, path rows: A x of the owned row, the path-row part of A^T y (read by the variable lanes after the reduction's barriers)
template <int NSEG>
__device__ __forceinline__ void q5_path_check(const Qp5Ctx<NSEG> &c, double zg, double yg, int gro, int xno, bool ownsRow, double (&sums)[2], double (&mx)[6]) {
    using L = Qp5<NSEG>;
    double *lds = c.lds;
    const double ax = q5_path_rows<NSEG>(lds, gro, xno, lds + L::vXx, lds + L::vGpy, [&](double) -> double { return ownsRow ? yg : 0.0; });
    if (ownsRow) {
        sums[0] = lds[lo16(gro) - (xno % L::XS) + ((gro >> 16) & 1) * L::GS + 21] * yg;      // T coefficient of the owned row (column 21)
        mx[0] = fabs(ax - zg); mx[1] = fabs(ax); mx[2] = fabs(zg);
    }
}

// ---- role EP: waves NSEG .. wS0.  P3 (E lanes); the S lanes of the mixed wave wS0 take part in P2; path rows from lane tP0 on ----
template <int NSEG>
__device__ __forceinline__ void qp5_role_ep(const Qp5Ctx<NSEG> &c) {
    using L = Qp5<NSEG>;
    using D = Dim3<NSEG>;
    double *lds = c.lds;
    const mpcmp_config &cfg = *c.cfg;
    const int tid = c.tid, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Q5_STAMP_DECL;
    const bool laneS = tid >= L::tS0, laneE = !laneS;
    const bool waveS = wave >= L::wS0, waveP = wave > NSEG;
    double fm[56];
    {
        typedef const __attribute__((address_space(1))) double *gptr_t;
        gptr_t p = (gptr_t)(laneS ? c.fa + L::F::oFS + (tid - L::tS0) : c.fa + L::F::oFE + (tid - L::NG));
        const int sd = laneS ? L::F::NSL : L::F::ELS;
#pragma unroll
        for (int j = 0; j < 52; j++) fm[j] = p[j * sd];
#pragma unroll
        for (int j = 52; j < 56; j++) { const double q = p[laneE ? j * sd : 0]; fm[j] = laneE ? q : 0.0; }
    }
    const double (&fs)[52] = reinterpret_cast<const double (&)[52]>(fm);
    double zg = 0.0, yg = 0.0;
    const bool ownsRow = waveP && (tid & 3) < 2;
    // ---- K_0 w = k (the T border) ----
    {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int k0 = q5_lc<NSEG>(c, t, 0), k1 = q5_lc<NSEG>(c, t, 1);
        __syncthreads();
        if (waveS) q5_p2<NSEG>(c, fs, t, k0, k1, laneS, false);
        __syncthreads();
        q5_p3<NSEG>(c, fm, k0, k1, laneE, false);
        __syncthreads();
    }
    q5_finish_border<NSEG>(c);
    Q5_STAMP_RESET;
    int it = 0, done = 0, until_check = cfg.check_every;
    for (it = 1; it <= cfg.qp_iters; it++) {
        int sio = tid;
        asm volatile("" : "+v"(sio));
        // ---- A (variable lanes) ----
        QB(0); __syncthreads(); QS(0);
        // ---- P1 (role G) ----
        const int k0 = q5_lc<NSEG>(c, sio, 0), k1 = q5_lc<NSEG>(c, sio, 1);
        QB(1); __syncthreads(); QS(1);
        if (waveS) q5_p2<NSEG>(c, fs, sio, k0, k1, laneS, true);
        QB(2); __syncthreads(); QS(2);
        q5_p3<NSEG>(c, fm, k0, k1, laneE, true);
        const int gro = q5_lc<NSEG>(c, sio, 2), xno = q5_lc<NSEG>(c, sio, 3);
        QB(3); __syncthreads(); QS(3);
        // ---- E ----
        const bool check = (--until_check == 0);
        if (check) until_check = cfg.check_every;
#ifndef Q5_NOPATH
        if (waveP) q5_path_E<NSEG>(c, zg, yg, gro, xno, sio - L::tP0, ownsRow);
#endif
        QB(4); __syncthreads(); QS(4);
#ifndef Q5_NOTEST
        if (__builtin_expect(check, 0)) {
            int t = tid;
            asm volatile("" : "+v"(t));
            double sums[2] = {0.0, 0.0};
            double mx[6] = {0, 0, 0, 0, 0, 0};
            if (waveP) q5_path_check<NSEG>(c, zg, yg, q5_lc<NSEG>(c, t, 2), q5_lc<NSEG>(c, t, 3), ownsRow, sums, mx);
            q5_reduce<L::NWV, 2, false>(sums, lds + L::oRedP, tid);
            done = q5_check_tail<NSEG>(c, sums, mx);
            QS(5);
        }
#endif
        if (done) break;
    }
    Q5_STAMP_DUMP(it > cfg.qp_iters ? cfg.qp_iters : it);
    if (ownsRow) c.ws.y[(size_t)c.b * (D::ma + D::na + 1) + D::meq + 8 * ((tid - L::tP0) >> 4) + 2 * (((tid - L::tP0) & 15) >> 2) + (tid & 3)] = yg;
}

// ---- role SP: waves wS0 + 1 .. 11.  P2; path rows; the last wave also sums the T border's partial sums, replicates x~_T, and carries
// the variables that have no lane on the G waves (and T) in its lanes tV0 .. (their ADMM state lives in LDS: sixteen lanes) ----
template <int NSEG>
__device__ __forceinline__ void qp5_role_sp(const Qp5Ctx<NSEG> &c) {
    using L = Qp5<NSEG>;
    using D = Dim3<NSEG>;
    constexpr int N = D::N, na = D::na, XS = L::XS;
    double *lds = c.lds;
    const mpcmp_config &cfg = *c.cfg;
    const int tid = c.tid, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *misc = lds + L::oMisc;
    Q5_STAMP_DECL;
    const bool waveX = wave == L::NWV - 1;
    double fm[52];
    {
        typedef const __attribute__((address_space(1))) double *gptr_t;
        gptr_t p = (gptr_t)(c.fa + L::F::oFS + (tid - L::tS0));
#pragma unroll
        for (int j = 0; j < 52; j++) fm[j] = p[j * L::F::NSL];
    }
    const bool isPath = tid < L::tV0, laneV = !isPath;
    double zg = 0.0, yg = 0.0;
    const bool ownsRow = isPath && (tid & 3) < 2;
    const double alpha = c.alpha, sigma = c.sigma;
    // ---- K_0 w = k (the T border) ----
    {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int k0 = q5_lc<NSEG>(c, t, 0), k1 = q5_lc<NSEG>(c, t, 1);
        __syncthreads();
        q5_p2<NSEG>(c, fm, t, k0, k1, true, false);
        __syncthreads();
        if (waveX) q5_p3_xT<NSEG>(c, t, false);
        __syncthreads();
    }
    q5_finish_border<NSEG>(c);
    Q5_STAMP_RESET;
    int it = 0, done = 0, until_check = cfg.check_every;
    for (it = 1; it <= cfg.qp_iters; it++) {
        int sio = tid;
        asm volatile("" : "+v"(sio));
        // ---- A: the last variables ----
        if (waveX && laneV) {
            const int vi = L::NG + (sio - L::tV0);
            const int pxr = q5_lc<NSEG>(c, sio, 4), prf = q5_lc<NSEG>(c, sio, 5), prb = q5_lc<NSEG>(c, sio, 6);
            const double *vst = lds + L::vVst + (sio - L::tV0);
            const double vx = ldv(vst), vzb = ldv(vst + 16), vyb = ldv(vst + 32);
            const double rb = ldv(lds + L::vVc + vi + 4 * L::NVL);
            const double wb = ldv(lds + L::vWv + lo16(pxr));
            const double r = (sigma * vx + (rb * vzb - vyb)) + q5_col_gather<NSEG>(lds, pxr, prf, prb, lds + L::vVc + vi, lds + L::vWg, lds + L::vGp);
            lds[hi16(pxr)] = r;                             // (T: the pad slot; its w of the border is zero)
            const double bp = sum8(wb * r);
            lds[L::vRedB + (vi >> 3)] = bp;
        }
        QB(0); __syncthreads(); QS(0);
        // ---- P1 (role G); the last wave sums the border's partial sums ----
        if (waveX) q5_p1_xT<NSEG>(c, sio);
        const int k0 = q5_lc<NSEG>(c, sio, 0), k1 = q5_lc<NSEG>(c, sio, 1);
        QB(1); __syncthreads(); QS(1);
        q5_p2<NSEG>(c, fm, sio, k0, k1, true, true);
        QB(2); __syncthreads(); QS(2);
        if (waveX) q5_p3_xT<NSEG>(c, sio, true);
        const int gro = q5_lc<NSEG>(c, sio, 2), xno = q5_lc<NSEG>(c, sio, 3);
        QB(3); __syncthreads(); QS(3);
        // ---- E ----
        const bool check = (--until_check == 0);
        if (check) until_check = cfg.check_every;
#ifndef Q5_NOPATH
        if (isPath) q5_path_E<NSEG>(c, zg, yg, gro, xno, sio - L::tP0, ownsRow);
        else
#endif
        {
            const int vi = L::NG + (sio - L::tV0);
            const int pxr = q5_lc<NSEG>(c, sio, 4);
            double *vst = lds + L::vVst + (sio - L::tV0);
            const double *vcl = lds + L::vVc + vi;
            double vx = ldv(vst), vzb = ldv(vst + 16), vyb = ldv(vst + 32);
            const int xpos = lo16(pxr);
            const double xtv = ldv(lds + L::vXn + xpos), vlb = ldv(vcl + 1 * L::NVL), vub = ldv(vcl + 2 * L::NVL);
            const double rb = ldv(vcl + 4 * L::NVL), rbi = ldv(vcl + 5 * L::NVL);
            vx = alpha * xtv + (1.0 - alpha) * vx;
            const double zrv = alpha * xtv + (1.0 - alpha) * vzb;
            const double znv = clip(zrv + vyb * rbi, vlb, vub);
            vyb += rb * (zrv - znv);
            vzb = znv;
            vst[0] = vx; vst[16] = vzb; vst[32] = vyb;
            if (vi == na) {     // the shared variable T: its state is published for the border solve and the tests
                misc[L::M_xT] = vx; misc[L::M_zbT] = vzb; misc[L::M_ybT] = vyb;
                misc[L::M_baseT] = (sigma * vx - 1.0) + (rb * vzb - vyb);
                if (check) { for (int k = 0; k < N; k++) lds[L::vXx + k * XS + 21] = vx; }
            } else if (check) lds[L::vXx + xpos] = vx;
        }
        QB(4); __syncthreads(); QS(4);
#ifndef Q5_NOTEST
        if (__builtin_expect(check, 0)) {
            int t = tid;
            asm volatile("" : "+v"(t));
            double sums[2] = {0.0, 0.0};
            double mx[6] = {0, 0, 0, 0, 0, 0};
            if (isPath) q5_path_check<NSEG>(c, zg, yg, q5_lc<NSEG>(c, t, 2), q5_lc<NSEG>(c, t, 3), ownsRow, sums, mx);
            const int vi = laneV ? L::NG + (t - L::tV0) : 0;
            const bool isVar = laneV && vi < na;
            const double *vcl = lds + L::vVc + vi, *vst = lds + L::vVst + (laneV ? t - L::tV0 : 0);
            const double ha = isVar ? ldv(vcl + 3 * L::NVL) : 0.0, vx = ldv(vst), vzb = ldv(vst + 16), vyb = ldv(vst + 32);
            if (isVar) sums[1] = ha * vx;
            q5_reduce<L::NWV, 2, false>(sums, lds + L::oRedP, tid);
            if (isVar) {
                const double hx = (fabs(ha) + cfg.hess_reg) * vx + ha * lds[L::vXx + 21];
                const double aty = q5_col_gather<NSEG>(lds, q5_lc<NSEG>(c, t, 4), q5_lc<NSEG>(c, t, 5), q5_lc<NSEG>(c, t, 6), vcl, lds + L::vYs, lds + L::vGpy) + vyb;
                mx[0] = fabs(vx - vzb); mx[1] = fabs(vx); mx[2] = fabs(vzb);
                mx[3] = fabs(hx + aty); mx[4] = fabs(hx); mx[5] = fabs(aty);
            }
            done = q5_check_tail<NSEG>(c, sums, mx);
            QS(5);
        }
#endif
        if (done) break;
    }
    Q5_STAMP_DUMP(it > cfg.qp_iters ? cfg.qp_iters : it);
    if (laneV) {
        const int vi = L::NG + (tid - L::tV0);
        const double *vst = lds + L::vVst + (tid - L::tV0);
        c.ws.p[(size_t)c.b * (na + 1) + vi] = vst[0];
        c.ws.y[(size_t)c.b * (D::ma + na + 1) + D::ma + vi] = vst[32];
    }
    if (ownsRow) c.ws.y[(size_t)c.b * (D::ma + na + 1) + D::meq + 8 * ((tid - L::tP0) >> 4) + 2 * (((tid - L::tP0) & 15) >> 2) + (tid & 3)] = yg;
}

template <int NSEG>
__global__ __launch_bounds__(768) void k_qp5(mpcmp_config cfg, WS ws, const Qp3Pat *__restrict__ pat, Xch xch, int B, const double *__restrict__ fac) {
    constexpr int NARM = 1;
    QP3_PROLOGUE_L(Qp5<NSEG>, 768, false)
    (void)SC; (void)JS; (void)xown; (void)xpar; (void)dead; (void)status; (void)ts; (void)n_tot; (void)mn_tot; (void)lane; (void)redp; (void)ma;
    using F = Qp5Fac<NSEG>;
    constexpr int XS = L::XS, NX = L::NX, TS = L::TS;
    Qp5Ctx<NSEG> c;
    c.cfg = &cfg; c.ws = ws; c.lds = lds; c.fa = fac + (size_t)b * F::FAC; c.tid = tid; c.b = b;
    c.tsT = tsT; c.rho_in = rho_in; c.rho_eq = rho_eq; c.sigma = sigma; c.alpha = alpha;
    const double *fa = c.fa;
    // ---------------- the factor's LDS-resident parts, as k_qp3f<NSEG, 1, 5> left them ----------------
    for (int i = tid; i < L::NAP; i += NT) lds[L::vKT + i] = fa[L::oFT + i];
    for (int i = tid; i < NSEG * 224 + (NSEG + 1) * 112; i += NT) lds[L::vKCJ + i] = fa[L::oFD + i];      // column form of K_JC, transposed dense blocks
    if (tid < 16) lds[L::vZR + tid] = 0.0;
    if (tid == 0) misc[L::M_sumha] = fa[L::oFH];
    for (int i = tid; i < L::vZ1 - L::vZ0; i += NT) lds[L::vZ0 + i] = 0.0;          // exchanged vectors, pads, partial sums
    // ---------------- lane jobs ----------------
    auto rhs_slot = [&](int ip) -> int {
        return ip < nJ ? L::vRhsJ + L::JS * (ip / 49) + ip % 49 : (ip < nJ + 7 ? L::vRhsU + (ip - nJ) : L::vRhsI + (ip - nJ - 7));
    };
    auto node_slot = [&](int v) -> int { return v < 14 * N ? XS * (v / 14) + v % 14 : XS * ((v - 14 * N) / 7) + 14 + (v - 14 * N) % 7; };
    // lane jobs of the ADMM iteration: constants lane-transposed in LDS, addresses packed in the lane-constant table (q5_lc)
    int f[L::NF] = {0, 0, 0, 0, 0, 0, 0, 0};
    f[4] = NX | ((L::vXn + NX) << 16); f[6] = (4 << 19) | (4 << 22);      // (no variable: pad slots, rows with a zero coefficient)
    if (tid < L::NG || tid >= L::tV0) {          // the variable of this lane
        const int vi = tid < L::NG ? tid : L::NG + (tid - L::tV0);
        double cf = 0.0, lo = 0.0, hi = 0.0, ha = 0.0, rb = rho_in;
        int rf = 0, rA = 0, rB = 0, colA = 4, colB = 4;        // (column 4 of the table: zeros)
        if (vi < na) {
            const int v = vi;
            var_h(v, ha, rb, lo, hi);
            const double zv = zg_[v];
            lo -= zv; hi -= zv;
            if (v < 14 * N) {
                const int k = v / 14, cc = v % 14;
                if (k % 3 != 0) { rA = 14 * 3 * (k / 3) + cc; colA = k % 3; }
                else {
                    if (k < N - 1) { rA = 14 * k + cc; colA = 0; }
                    if (k > 0) { rB = 14 * (k - 3) + cc; colB = 3; }
                }
                if (cc >= 7 && k <= N - 2) { rf = 14 * k + (cc - 7); cf = -tsT; }
            } else {
                const int k = (v - 14 * N) / 7, cc = (v - 14 * N) % 7;
                if (k <= N - 2) { rf = 14 * k + 7 + cc; cf = -tsT; }
            }
            f[4] = node_slot(v) | (rhs_slot(int3_of_ext(NSEG, v)) << 16);
        } else {                                              // T
            lo = cfg.lbT - T; hi = cfg.ubT - T;
            rb = (cfg.ubT - cfg.lbT < 1e-4) ? rho_eq : rho_in;
            f[4] = 21 | ((L::vXn + NX) << 16);                // (x~_T is read from slot 21 of node 0; no rhs entry: b_T is formed by the border)
            misc[L::M_rbT] = rb;
        }
        f[5] = rf | (rA << 16);
        f[6] = rB | (colA << 19) | (colB << 22);
        double *vc = lds + L::vVc + vi;
        vc[0] = cf; vc[1 * L::NVL] = lo; vc[2 * L::NVL] = hi; vc[3 * L::NVL] = ha; vc[4 * L::NVL] = rb; vc[5 * L::NVL] = 1.0 / rb;
    }
    if (tid < 16) lds[L::vDT + tid] = tid < 12 ? c_D[4 * (tid % 3) + tid / 3] : 0.0;
    if (tid < L::NG) {          // the dynamics row of this lane (lanes past the last row: row 0's operands)
        const int r = tid < meq ? tid : 0, k = r / 14, rr = r % 14;
        lds[L::vRc + tid] = -ws.ceq[(size_t)b * meq + r];
        lds[L::vRc + L::NG + tid] = coef_T(r);
        f[6] |= (k % 3) << 17;
        f[7] = (3 * (k / 3) * XS + rr) | ((k * XS + (rr < 7 ? 7 + rr : 14 + rr - 7)) << 16);
    }
    if (tid >= L::tP0) {        // path rows: four lanes per pair of rows, sixteen lanes per node
        const int et = tid - L::tP0, pk = et >> 4, prp = (et & 15) >> 2, pq = et & 3, q = 2 * prp + pq;
        double lg = 0.0, ug = 0.0;
        if (tid < L::tV0 && pq < 2) {
            const double gv = ws.g[(size_t)b * 8 * N + 8 * pk + q];
            lg = c_lbg[q] - gv; ug = c_ubg[q] - gv;
        }
        lds[L::vPc + et] = lg; lds[L::vPc + L::NPL + et] = ug;
        { const double rr = (ug - lg < 1e-4) ? rho_eq : rho_in; lds[L::vPc + 2 * L::NPL + et] = rr; lds[L::vPc + 3 * L::NPL + et] = 1.0 / rr; }
        if (tid < L::tV0) {
            // bits 0..15: Jacobian operand (columns 6 pq .. 6 pq + 5 of the 24-wide padded rows 2 prp, 2 prp + 1 of node pk); 16: parity; 17: publishes gp
            f[2] = (L::oGk + (pk * 8 + 2 * prp) * L::GS + pq * 6) | ((pq & 1) << 16) | ((prp == 0 ? 1 : 0) << 17);
            f[3] = pk * XS + pq * 6;
        }
    }
    // constants of the solve
    if (wave < NSEG) {                              // P1 (k_qp3's LaneC1)
        const int ln = tid & 63, wv = wave;
        const bool g_row = ln < 49, gu_quad = wv == NSEG - 1 && ln >= 56, gu_row = gu_quad && ln < 63;
        const int tJ = L::vTJ + wv * TS;
        const int tslot = g_row ? tJ + ln : (gu_row ? L::vTU + ln - 56 : tJ + 56);                    // t slot of the own row
        const int op1 = (gu_quad ? L::vRhsU : L::vRhsJ + L::JS * wv) + 13 * (ln & 3);                  // operand block of G b_J
        const int cl = ln < 28 ? ln : 27;
        const int kcj = L::vKCJ + wv * 224 + cl;                                                       // column form of K_JC
        const int tcc = tJ + (cl < 14 ? cl : cl - 14);                                                 // t + c % 14
        const int partd = ln < 14 ? L::vPA + 14 * wv + ln : (ln < 28 ? L::vPB + 14 * wv + ln : L::vPD);
        const int c2 = ln >> 1, j = ln & 1;
        const bool lastw = wv == NSEG - 1, useg = c2 < 14, uU = lastw && c2 >= 14 && c2 < 28;
        const int p1k = useg ? L::vKUXT + (wv * 14 + c2) * 8 + 4 * j : (uU ? L::vKUXT + (NSEG * 14 + c2 - 14) * 8 + 4 * j : L::vZR);
        const int p1t = (uU ? L::vTU : tJ) + 4 * j;
        const int p1d = j ? L::vPD : (useg ? L::vDP + wv * 14 + c2 : (uU ? L::vPA + NSEG * 14 + c2 - 14 : L::vPD));
        f[0] = tslot | (op1 << 16); f[1] = kcj | (tcc << 16); f[2] = partd | (p1k << 16); f[3] = p1t | (p1d << 16);
    } else if (tid >= L::tS0) {                     // P2
        const int ls = tid - L::tS0, c8 = ls & 7, srow = 4 * (ls >> 3) + (c8 & 3);
        const bool out = c8 < 4 && srow < nI;
        f[0] = (L::vRIw + 13 * c8) | ((L::vYI + (out ? srow : L::YPAD)) << 16);                        // operand block of S^-1 r_I | y_I slot
        f[1] = L::vXn + (out ? XS * 3 * (srow / 14) + srow % 14 : NX);                                 // x~ slot of the interface row
    } else {                                        // P3
        const int le = tid - L::NG, sE = le / 26, rem = le % 26, g = rem >> 1, h = rem & 1, r0 = 4 * g + 2 * h;
        int yop = L::vYI, tsl = L::vTU + 8, xd0 = NX, xd1 = NX + 1;
        if (sE < NSEG) {
            yop = L::vYI + 14 * (sE + h); tsl = L::vTJ + sE * TS + r0;
            if (r0 < 49) xd0 = node_slot(ws.ext_of_int[49 * sE + r0]);
            if (r0 + 1 < 49) xd1 = node_slot(ws.ext_of_int[49 * sE + r0 + 1]);
        } else if (sE == NSEG && rem < 4) {
            yop = L::vYI + 14 * (NSEG + h); tsl = L::vTU + r0;
            if (r0 < 7) xd0 = node_slot(ws.ext_of_int[nJ + r0]);          // (the h = 1 lane of a pair finishes rows 4 g + 2, + 3: its own block is zero, its partner's is not)
            if (r0 + 1 < 7) xd1 = node_slot(ws.ext_of_int[nJ + r0 + 1]);
        }
        f[0] = yop | (tsl << 16); f[1] = (L::vXn + xd0) | ((L::vXn + xd1) << 16);
    }
    {
        int *lct = reinterpret_cast<int *>(lds + L::vLCT);
#pragma unroll
        for (int q = 0; q < L::NF; q++) lct[q * NT + tid] = f[q];
    }
    __syncthreads();
    for (int ip = tid; ip < na; ip += NT) lds[rhs_slot(ip)] = lds[L::vKT + ip];      // rhs of K_0 w = k
    __syncthreads();
    if (wave < NSEG) qp5_role_g<NSEG>(c);
    else if (wave <= L::wS0) qp5_role_ep<NSEG>(c);
    else qp5_role_sp<NSEG>(c);
}

}  // namespace mpcmp
