// qp_kernel_v5.hpp — k_qp5<NSEG>: the ADMM loop of the N = 19 QP (the reference as shipped, robot_ocp.hpp:31-32) with EVERY dense
// factor of the solve resident on the CU, loaded once.
//
// Same arithmetic as k_qp3 (T bordered out, one arm; structure3.hpp) with k_qp2's solve and k_qp2's row / variable scheme: the
// interior solve uses E_s = G_s K_JC,s (formed by k_qp3f on the matrix cores and handed over, LAY = 5) instead of a second product
// with G_s,
//     t = G b_J;   r_I = b_I - K_CJ t;   y_I = S^-1 r_I;   x_J = t - E y_C,
// so one solve is three barrier phases, each ONE product on a register-resident block.  768 threads = 12 waves, three per SIMD,
// 168 registers per lane; the factor (G 6 x 49 x 49, E 6 x 49 x 28, S^-1 98 x 98 = 32 k doubles, 35 k with padding) takes ~ 100
// registers of every lane (the tail column of each block lives in LDS, read with the operands of the product):
//     role G  waves 0..5, wave = segment: 4 x 13 block of G_s per lane (quad = four rows; lanes 56..62 of wave 5: G_u)   P1
//     role E  lanes 384..567: 4 x 14 block of E_s per lane (two lanes = four rows; "segment" 6 = E_u)                     P3
//     role S  lanes 568..767: 4 x 13 block of S^-1 per lane (eight lanes = four rows)                                     P2
// E^T is not kept (it does not fit): K_CJ t is taken by the G waves from their own t (K_CJ is sparse plus one dense 7 x 14 block per
// segment), inside the wave, as in k_qp3.  The rows and variables of the ADMM iteration are spread as in k_qp2, on the waves that have
// the registers and the idle phases for them: the path rows on sixteen lanes per node on the G waves 0..4 (four lanes per pair of
// rows, six columns each; the node's path-row part of A^T w is formed by the same lanes), a variable per lane on the E / S waves, a
// dynamics row per lane on the pure S waves 9..11; the last G wave carries what is left (15 variables, T, 60 dynamics rows).  Five
// workgroup barriers per ADMM iteration, three role loops (G; waves 6..8; waves 9..11):
//     A   rhs = sigma x - q + rho_b z_b - y_b + A^T w                      (variable lanes)
//     P1  t = G b_J, K_CJ t; x~_T of the T border                          (role G; the last wave sums the border's partial sums)
//     P2  r_I = b_I - K_CJ t, y_I = S^-1 r_I, interface rows of x~         (S lanes)
//     P3  x_J = t - E y_C - w x~_T                                         (E lanes)
//     E   z~ = A x~, relaxation, projection, dual update                   (every lane: its row and / or its variable)
// The factor is loaded ONCE (k_qp3 re-reads its blocks at every termination test: 2 GB per launch).  What makes that possible at 168
// registers per lane: no per-lane integer lives in a register across the loop (lane-constant table in LDS, fetched in the idle phase
// before its use), the termination test is a cold block with its own lean reductions, and the row / variable jobs sit where the
// block leaves room.  DESIGN.md has the measurements (tools/ablate5.py: what-if profile; tools/stamps5.py: phase stamps).
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
    static constexpr int NPN = 16 * D::N;                            // path-row lanes: sixteen per node, on the G waves (tid < 304)
    static constexpr int tV5 = 64 * (NSEG - 1);                      // the last G wave carries no path rows: its first sixteen lanes take the variables NG .. and T
    static constexpr int NPL = tV5;                                  // lanes of the path-row tables (320)
    static constexpr int NVL = NG + 16;                              // variable lanes: vi = vi_of(tid) on the E / S waves, NG + (tid - tV5) on the last G wave (400)
    static constexpr int tSP = 64 * (tS0 / 64 + 1);                  // first pure S wave (576): these waves have 8 registers more to spare than the E waves, so they take
    static constexpr int DR0 = NT - tSP;                             // dynamics rows 0 .. DR0 - 1 on the pure S lanes (row = vi), rows DR0 .. meq - 1 on the first lanes of the last G wave
    static constexpr int ND5 = D::meq - DR0;
    static_assert(ND5 >= 16 && ND5 <= 64 && DR0 % 8 == 0, "dynamics rows of the last G wave");
    __host__ __device__ static constexpr int vi_of(int tid) { return tid >= tSP ? tid - tSP : tid - NG + (NT - tSP); }      // vi = 0 .. 191 (all with a dynamics row); the E waves vi = 192 .. 383
    static_assert(NPN <= tV5 && D::na == NVL - 1, "row / variable map");
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
    static constexpr int NRC = (D::meq + 7) / 8 * 8;                 // rows of the dynamics-row tables
    static constexpr int vRc = vPc + 4 * NPL;                        // [2][NRC] dynamics rows: l = u = -c_eq, T coefficient -ts f
    static constexpr int vZ0 = vRc + 2 * NRC;                         // ---- zero-initialised from here ----
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
    static constexpr int NF = 6;                                     // lane-constant table: [NF][NT] 32-bit words (q5_lc)
    static constexpr int vLCT = vZ1;
    static constexpr int vL5 = vLCT + NF * NT / 2;                   // [4][64] 32-bit words: variable / dynamics-row constants of the last G wave (q5_l5)
    // The tail of every lane's factor block lives in LDS, lane-transposed (read with the operands of the product: same round trip): the
    // register budget of a lane (168) minus its block leaves too little for the rows and variables otherwise (scratch reloads cost ~ 500 cycles each)
    static constexpr int RG = 2, RE = 4, RS = 4;
    static constexpr int vTG = vL5 + 128;                            // [RG][NG]  entries 12, 25 of the 4 x 13 blocks of G (column 12 of rows 0, 1)
    static constexpr int vTE = vTG + RG * NG;                        // [RE][ELS] entries 13, 27, 41, 55 of the 4 x 14 blocks of E (column 13)
    static constexpr int vTS = vTE + RE * F::ELS;                    // [RS][NSL] entries 12, 25, 38, 51 of the 4 x 13 blocks of S^-1 (column 12)
    static constexpr int size5 = vTS + RS * NSL;
    static_assert(size5 * 8 <= 160 * 1024 - 512, "LDS budget");
    static_assert(vXn % 2 == 0 && vGp % 2 == 0 && vGpy % 2 == 0 && vXx % 2 == 0 && vPadW % 2 == 0 && oGk % 2 == 0, "16-byte accesses");
};

// Register-resident block products.  m: the lane's block without its tail entries, tl: the lane's tail in LDS (stride ts between entries).
// G: 4 x 13 block, four lanes per group of four rows (k_qp3's g_blk: lane (quad g, m) holds rows 4 g + (m ^ pos) x columns 13 m .. + 12; a
// reduce-scatter over the quad leaves row 4 g + m in lane 4 g + m); entries 12 and 25 (column 12 of pos 0, 1) come from LDS.
__device__ __forceinline__ double q5_gprod(const double (&m)[50], const double *tl, const int ts, const double *op) {
    double o[13];
#pragma unroll
    for (int j = 0; j < 13; j++) o[j] = ldv(op + j);
    const double t0 = ldv(tl), t1 = ldv(tl + ts);
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
    for (int j = 0; j < 12; j++) { p0 += m[j] * o[j]; p1 += m[12 + j] * o[j]; p2 += m[24 + j] * o[j]; p3 += m[37 + j] * o[j]; }
    p0 += t0 * o[12]; p1 += t1 * o[12]; p2 += m[36] * o[12]; p3 += m[49] * o[12];
    const double q0 = p0 + dpp_mov<0x4E>(p2), q1 = p1 + dpp_mov<0x4E>(p3);          // lanes m, m ^ 2
    return q0 + dpp_mov<0xB1>(q1);                                                  // lanes m, m ^ 1
}
// S^-1: 4 x 13 block, eight lanes per group of four rows: quad reduce-scatter as above, then the group's two quads are added; column 12 of
// every row (entries 12, 25, 38, 51) comes from LDS.
__device__ __forceinline__ double q5_sprod(const double (&m)[48], const double *tl, const int ts, const double *op) {
    double o[13], t[4];
#pragma unroll
    for (int j = 0; j < 13; j++) o[j] = ldv(op + j);
#pragma unroll
    for (int q = 0; q < 4; q++) t[q] = ldv(tl + q * ts);
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
    for (int j = 0; j < 12; j++) { p0 += m[j] * o[j]; p1 += m[12 + j] * o[j]; p2 += m[24 + j] * o[j]; p3 += m[36 + j] * o[j]; }
    p0 += t[0] * o[12]; p1 += t[1] * o[12]; p2 += t[2] * o[12]; p3 += t[3] * o[12];
    const double q0 = p0 + dpp_mov<0x4E>(p2), q1 = p1 + dpp_mov<0x4E>(p3);
    const double x = q0 + dpp_mov<0xB1>(q1);
    return x + dpp_xor4(x);
}

// what-if profiling (tools/ablate5.py): -DQ5_ABL=n removes one piece of the ADMM iteration (results are then wrong); the change in run time at a
// fixed iteration count is that piece's share of the critical path.  0 = product build.
#ifndef Q5_ABL
#define Q5_ABL 0
#endif
#define Q5_ON(n) (Q5_ABL != (n) && Q5_ABL < 10)        /* 10: every piece off (the bare loop: barriers, lane-constant fetches, the border's x~_T chain) */
#define Q5_XT (Q5_ABL < 11)                            /* 11: 10 and no x~_T chain;  12: 11 and no lane-constant fetches;  13: 12 with three barriers instead of five */
#define Q5_BAR(k) do { if (!(Q5_ABL == 13 && ((k) == 2 || (k) == 3))) __syncthreads(); } while (0)
template <int NSEG>
struct Qp5Ctx {
    const mpcmp_config *cfg;
    WS ws;
    double *lds;
    const double *fa;
    int tid, b;
    double tsT, rho_in, rho_eq, sigma, alpha, oma;     // (oma = 1 - alpha)
};
// a workgroup-uniform double that vector instructions produced, moved to scalar registers (it would otherwise occupy two vector registers of every role's loop)
__device__ __forceinline__ double q5_uniform(double x) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

// diagnostic builds (-DMPCMP_STAMPS, tools/stamps5.py): cycles per phase of the ADMM loop as wave 0 sees them (Q5S: after a barrier) and the busy
// part of each phase per wave (Q5B: in front of the barrier).  -DMPCMP_STAMPS_LIGHT: wave 0's phase stamps only (the per-wave counters cost ~ 20 %).
#ifdef MPCMP_STAMPS
#define Q5_STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_busy[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = clock64(); const bool st_on = c.tid < 64; (void)st_on; (void)st_busy
#define Q5_STAMP_RESET do { for (int k_ = 0; k_ < 8; k_++) st_acc[k_] = st_busy[k_] = 0; st_t = clock64(); } while (0)
#define Q5_STAMP_DUMP(it_) do { if ((c.tid & 63) == 0) { unsigned long long *o_ = c.ws.dbg + (size_t)c.b * MPCMP_DBG_WORDS; \
        for (int k_ = 0; k_ < 8; k_++) o_[16 + (c.tid >> 6) * 8 + k_] = st_busy[k_]; \
        if (c.tid == 0) { for (int k_ = 0; k_ < 7; k_++) o_[k_] = st_acc[k_]; o_[15] = (it_); } } } while (0)
#ifdef MPCMP_STAMPS_LIGHT
#define Q5S(k) do { if (st_on) { const unsigned long long n_ = clock64(); st_acc[k] += n_ - st_t; st_t = n_; } } while (0)
#define Q5B(k) do { } while (0)
#else
#define Q5S(k) do { const unsigned long long n_ = clock64(); st_acc[k] += n_ - st_t; st_t = n_; } while (0)
#define Q5B(k) do { st_busy[k] += clock64() - st_t; } while (0)
#endif
#else
#define Q5_STAMP_DECL do { } while (0)
#define Q5_STAMP_RESET do { } while (0)
#define Q5_STAMP_DUMP(it_) do { } while (0)
#define Q5S(k) do { } while (0)
#define Q5B(k) do { } while (0)
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
    if (Q5_ABL >= 12) return 0;
    return *(const volatile __attribute__((address_space(3))) int *)(lct + f * L::NT + t);
}
template <int NSEG>
__device__ __forceinline__ int q5_l5(const Qp5Ctx<NSEG> &c, int lane, int f) {       // last G wave: 0 pxr, 1 prf, 2 prb, 3 ixr of the lane's variable / dynamics row
    using L = Qp5<NSEG>;
    const int *t5 = reinterpret_cast<const int *>(c.lds + L::vL5);
    if (Q5_ABL >= 12) return 0;
    return *(const volatile __attribute__((address_space(3))) int *)(t5 + f * 64 + lane);
}
__device__ __forceinline__ int lo16(int w) { return w & 0xFFFF; }
__device__ __forceinline__ int hi16(int w) { return (int)((unsigned)w >> 16); }

// ---- the three products of one solve with K_0 ----
// P1 (role G): t = G b_J (k_qp3's block product), then inside the wave part = K_CJ t
template <int NSEG>
__device__ __forceinline__ void q5_p1(const Qp5Ctx<NSEG> &c, const double (&fm)[50], int t, int k0, int k1, int k2, int k3) {
    using L = Qp5<NSEG>;
    double *lds = c.lds;
    lds[lo16(k0)] = q5_gprod(fm, lds + L::vTG + t, L::NG, lds + hi16(k0));
    wave_sync();
    const double *kc = lds + lo16(k1), *tc = lds + hi16(k1);
    double kq[7], tv[7], dk[4], dt[4];
#pragma unroll
    for (int d = 0; d < 7; d++) { kq[d] = ldv(kc + 28 * d); tv[d] = ldv(tc + 7 * (d - 1)); }      // rows c % 14 + 7 (d - 1) of the segment
    // dense blocks (K_XU t of the columns x_3s; for the last segment also the U block): half a column per lane
    // (every read of this part is issued before its first store: one LDS round trip, not two)
#pragma unroll
    for (int d = 0; d < 4; d++) { dk[d] = ldv(lds + hi16(k2) + d); dt[d] = ldv(lds + lo16(k3) + d); }
    const double acc = ((kq[0] * tv[0] + kq[1] * tv[1]) + (kq[2] * tv[2] + kq[3] * tv[3])) + ((kq[4] * tv[4] + kq[5] * tv[5]) + kq[6] * tv[6]);
    const double ad = (dk[0] * dt[0] + dk[1] * dt[1]) + (dk[2] * dt[2] + dk[3] * dt[3]);
    lds[lo16(k2)] = acc;
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
__device__ __forceinline__ void q5_p2(const Qp5Ctx<NSEG> &c, const double (&fm)[48], int t, int k0, int k1, const bool laneS, const bool use_xT) {
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
    const double yi = q5_sprod(fm, lds + L::vTS + (laneS ? t - L::tS0 : 0), L::NSL, lds + (laneS ? lo16(k0) : L::vRIw));
    lds[ysl] = yi;                                                            // (lanes without an output row: pad slots)
    lds[xds] = yi - wds * xT;
}
// P3 (waves with E lanes): x_J = t - E y_C - w x~_T.  4 x 14 block, two lanes per group of four rows (lane h of a pair: rows 4 g + (a ^ 2 h),
// columns 14 h .. + 13); column 13 of every row comes from LDS.  Operand reads in two batches.
template <int NSEG>
__device__ __forceinline__ void q5_p3(const Qp5Ctx<NSEG> &c, const double (&fm)[52], int t, int k0, int k1, const bool laneE, const bool use_xT) {
    using L = Qp5<NSEG>;
    double *lds = c.lds;
    constexpr int dW = L::vWv - L::vXn;
    const int xd0 = laneE ? lo16(k1) : L::vXn + L::NX, xd1 = laneE ? hi16(k1) : L::vXn + L::NX + 1;      // (the S lanes of the mixed wave: pad slots)
    const double *yc = lds + (laneE ? lo16(k0) : L::vYI), *ts = lds + (laneE ? hi16(k0) : L::vTU + 8), *tl = lds + L::vTE + (laneE ? t - L::NG : 0);
    const double t0 = ldv(ts), t1 = ldv(ts + 1);
    const double w0 = ldv(lds + xd0 + dW), w1 = ldv(lds + xd1 + dW);
    const double xT = use_xT ? ldv(lds + L::oMisc + L::M_xtT) : 0.0;
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
    {
        double o[7];
#pragma unroll
        for (int j = 0; j < 7; j++) o[j] = ldv(yc + j);
#pragma unroll
        for (int j = 0; j < 7; j++) { p0 += fm[j] * o[j]; p1 += fm[13 + j] * o[j]; p2 += fm[26 + j] * o[j]; p3 += fm[39 + j] * o[j]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        double o[7], tq[4];
#pragma unroll
        for (int j = 0; j < 7; j++) o[j] = ldv(yc + 7 + j);
#pragma unroll
        for (int q = 0; q < 4; q++) tq[q] = ldv(tl + q * L::F::ELS);
#pragma unroll
        for (int j = 0; j < 6; j++) { p0 += fm[7 + j] * o[j]; p1 += fm[20 + j] * o[j]; p2 += fm[33 + j] * o[j]; p3 += fm[46 + j] * o[j]; }
        p0 += tq[0] * o[6]; p1 += tq[1] * o[6]; p2 += tq[2] * o[6]; p3 += tq[3] * o[6];
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

// termination test in two stages (every thread; the result is the conjunction the one-stage test formed): the PRIMAL residual first — it fails in
// nine tests of ten on the bench workloads — and the dual one, with everything it needs (A^T y of the path rows, the gathers, the sums of the T row /
// column), only if it passes.  Two barriers each; the T variable's own terms are added from its published state.
template <int NSEG>
__device__ __forceinline__ int q5_primal_ok(const Qp5Ctx<NSEG> &c, const double (&mp)[3]) {
    using L = Qp5<NSEG>;
    const double *misc = c.lds + L::oMisc;
    double v[3] = {mp[0], mp[1], mp[2]};
    block_reduce_lean<L::NWV, 3, 3>(v, c.lds + L::oRedP, c.tid);
    const double xTv = misc[L::M_xT], zT = misc[L::M_zbT];
    const double m0 = fmax(v[0], fabs(xTv - zT)), m1 = fmax(v[1], fabs(xTv)), m2 = fmax(v[2], fabs(zT));
    return m0 <= c.cfg->eps_abs + c.cfg->eps_rel * fmax(m1, m2) ? 1 : 0;
}
template <int NSEG>
__device__ __forceinline__ int q5_dual_ok(const Qp5Ctx<NSEG> &c, const double (&sums)[2], const double (&md)[3]) {
    using L = Qp5<NSEG>;
    const double *misc = c.lds + L::oMisc;
    double v[5] = {md[0], md[1], md[2], sums[0], sums[1]};
    block_reduce_lean<L::NWV, 5, 3>(v, c.lds + L::oRedP + 48, c.tid);      // (its own slots: the primal reduction's are still being read)
    const double xTv = misc[L::M_xT], yT = misc[L::M_ybT];
    const double hxT = misc[L::M_hdT] * xTv + v[4], atyT = v[3] + yT;
    const double m3 = fmax(v[0], fabs(hxT + atyT + 1.0)), m4 = fmax(v[1], fabs(hxT)), m5 = fmax(v[2], fabs(atyT));
    return m3 <= c.cfg->eps_abs + c.cfg->eps_rel * fmax(fmax(m4, m5), 1.0) ? 1 : 0;      // ||q||_inf = 1
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

// ---- the path rows (k_qp2's scheme): four lanes per pair of rows (six columns each), sixteen lanes per node; lanes 0, 1 of a quad own
// the rows 2 prp, 2 prp + 1.  z~ of the owned row and, from the same Jacobian operands, this node's path-row part of A^T w: every lane
// forms its six columns of g_row0 w0 + g_row1 w1, the four row pairs of the node (lane bits 2, 3 of the DPP row) are summed with two
// row rotations, and the lanes of pair 0 publish the node's 24 padded columns.
struct PathOp5 { v2d pa[3], pb[3]; double lgp, ugp, rr, rri; };      // Jacobian operands (the row of the lane's parity first) and constants of the owned row
template <int NSEG>
__device__ __forceinline__ PathOp5 q5_path_fetch(const double *lds, int gro, int t) {
    using L = Qp5<NSEG>;
    const int par = (gro >> 16) & 1, go = lo16(gro);
    const double *g0 = lds + go + par * L::GS, *g1 = lds + go + (1 - par) * L::GS, *pcl = lds + L::vPc + t;
    PathOp5 o;
#pragma unroll
    for (int j = 0; j < 3; j++) o.pa[j] = ldv2(g0 + 2 * j);
#pragma unroll
    for (int j = 0; j < 3; j++) o.pb[j] = ldv2(g1 + 2 * j);
    o.lgp = ldv(pcl); o.ugp = ldv(pcl + L::NPL); o.rr = ldv(pcl + 2 * L::NPL); o.rri = ldv(pcl + 3 * L::NPL);
    return o;
}
template <int NSEG, class F>
__device__ __forceinline__ double q5_path_rows(double *lds, int gro, int xno, int t, const double *xe, double *gdst, F &&row_update) {
    using L = Qp5<NSEG>;
    const int first = (gro >> 17) & 1;
    const double *xv = xe + xno;
    v2d x2[3];
#pragma unroll
    for (int j = 0; j < 3; j++) x2[j] = ldv2(xv + 2 * j);
    const PathOp5 o = q5_path_fetch<NSEG>(lds, gro, t);                  // (all operand reads of the phase in flight at once: one LDS round trip)
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        a0 += o.pa[j].x * x2[j].x; a1 += o.pb[j].x * x2[j].x;
        a0 += o.pa[j].y * x2[j].y; a1 += o.pb[j].y * x2[j].y;
    }
    const double ax = quad_sum2(a0, a1);
    const double wq = row_update(ax, o);                                // owners: the row's multiplier-like value
    const double w0 = dpp_mov<0x44>(wq), w1 = dpp_mov<0x11>(wq);        // quad broadcasts: owner of the own row, of the other row
    double *dst = first ? gdst + xno : lds + L::vPadW;                  // (pairs 1..3 of a node: pad)
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double cx = o.pa[j].x * w0 + o.pb[j].x * w1, cy = o.pa[j].y * w0 + o.pb[j].y * w1;
        cx += dpp_mov<0x128>(cx); cy += dpp_mov<0x128>(cy);             // row_ror:8
        cx += dpp_mov<0x124>(cx); cy += dpp_mov<0x124>(cy);             // row_ror:4
        D2 q; q.x = cx; q.y = cy;
        *reinterpret_cast<D2 *>(dst + 2 * j) = q;
    }
    return ax;
}

// A x of the owned path row alone (stage one of the termination test)
template <int NSEG>
__device__ __forceinline__ double q5_path_ax(const double *lds, int gro, int xno, int t, const double *xe) {
    const double *xv = xe + xno;
    v2d x2[3];
#pragma unroll
    for (int j = 0; j < 3; j++) x2[j] = ldv2(xv + 2 * j);
    const PathOp5 o = q5_path_fetch<NSEG>(lds, gro, t);
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        a0 += o.pa[j].x * x2[j].x; a1 += o.pb[j].x * x2[j].x;
        a0 += o.pa[j].y * x2[j].y; a1 += o.pb[j].y * x2[j].y;
    }
    return quad_sum2(a0, a1);
}

// ---- role G: waves 0 .. NSEG - 1.  P1 (wave = segment).  The path rows live here, sixteen lanes per node on the lanes below NPN: these
// waves are idle in P2 and P3 (their Jacobian operands and row constants are fetched then) and their factor block leaves the most
// registers.  The first sixteen lanes of the last G wave carry the variables that have no lane on the E / S waves, and T (state in LDS). ----
template <int NSEG>
__device__ __forceinline__ void qp5_role_g(const Qp5Ctx<NSEG> &c) {
    using L = Qp5<NSEG>;
    using D = Dim3<NSEG>;
    constexpr int N = D::N, na = D::na, XS = L::XS;
    double *lds = c.lds;
    const mpcmp_config &cfg = *c.cfg;
    const int tid = c.tid, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    double *misc = lds + L::oMisc;
    Q5_STAMP_DECL;
    double fm[50];              // the lane's 4 x 13 block of G_s without the entries 12, 25 (LDS: vTG)
    {
        typedef const __attribute__((address_space(1))) double *gptr_t;
        gptr_t p = (gptr_t)(c.fa + L::oFG + (size_t)(wave * 52) * 64 + lane);
#pragma unroll
        for (int e = 0; e < 52; e++) {
            const double q = p[e * 64];
            if (e == 12) lds[L::vTG + tid] = q;
            else if (e == 25) lds[L::vTG + L::NG + tid] = q;
            else fm[e < 12 ? e : (e < 25 ? e - 1 : e - 2)] = q;
        }
    }
    const bool isPath = tid < L::NPN, ownsRow = isPath && (tid & 3) < 2;
    const bool wavePath = (tid & ~63) < L::NPN;
    const bool laneV = tid >= L::tV5 && tid < L::tV5 + 16, laneD = tid >= L::tV5 && tid < L::tV5 + L::ND5;
    double zg = 0.0, yg = 0.0;          // ADMM state of the owned path row (last G wave: yg = the dual of the lane's dynamics row)
    const double alpha = c.alpha, sigma = c.sigma, rho_eq = c.rho_eq, mtsT = -c.tsT;
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
        q5_p1<NSEG>(c, fm, t, q5_lc<NSEG>(c, t, 0), q5_lc<NSEG>(c, t, 1), q5_lc<NSEG>(c, t, 2), q5_lc<NSEG>(c, t, 3));
    }
    __syncthreads();
    __syncthreads();
    __syncthreads();
    q5_finish_border<NSEG>(c);
    Q5_STAMP_RESET;
    int it = 0, done = 0;
    int pxr, prf, prb;          // last G wave: constants of phase A, fetched ahead of the barrier that ends the previous iteration
    {
        int t = tid;
        asm volatile("" : "+v"(t));
        pxr = q5_l5<NSEG>(c, t & 63, 0); prf = q5_l5<NSEG>(c, t & 63, 1); prb = q5_l5<NSEG>(c, t & 63, 2);
    }
    const bool warm = cfg.qp_warm_start != 0;
    if (warm) {     // warm duals (mpcmp_config.qp_warm_start): y_0 = lambda_k, x_0 = 0, z_0 = clip(0, l, u); w_0 = rho z_0 - y_0 published
        const double *lamb = c.ws.lam + (size_t)c.b * (D::ma + na + 1);
        int t = tid;
        asm volatile("" : "+v"(t));
        if (isPath) {
            const int gro = q5_lc<NSEG>(c, t, 4), xno = q5_lc<NSEG>(c, t, 5);
            if (ownsRow) yg = lamb[D::meq + 8 * (tid >> 4) + 2 * ((tid & 15) >> 2) + (tid & 3)];
            q5_path_rows<NSEG>(lds, gro, xno, t, lds + L::vXn, lds + L::vGp, [&](double, const PathOp5 &po) -> double {
                if (ownsRow) zg = clip(0.0, po.lgp, po.ugp);
                return ownsRow ? po.rr * zg - yg : 0.0;
            });
        }
        if (laneD) {
            const int r = L::DR0 + (t - L::tV5);
            const double lgd = lds[L::vRc + r], rcT = lds[L::vRc + L::NRC + r];
            yg = lamb[r];
            const double w = rho_eq * lgd - yg;
            lds[L::vWg + r] = w;
            const double tp = sum8(rcT * w);
            lds[L::vRedT + (r >> 3)] = tp;
        }
        if (laneV) {
            const int vi = L::NG + (t - L::tV5);
            double *vst = lds + L::vVst + (t - L::tV5);
            const double *vcl = lds + L::vVc + vi;
            const double vzb = clip(0.0, vcl[1 * L::NVL], vcl[2 * L::NVL]), vyb = lamb[D::ma + vi], rb = vcl[4 * L::NVL];
            vst[16] = vzb; vst[32] = vyb;
            if (vi == na) { misc[L::M_zbT] = vzb; misc[L::M_ybT] = vyb; misc[L::M_baseT] = -1.0 + (rb * vzb - vyb); }
        }
        __syncthreads();
    }
    // (periods of check_every iterations in an inner loop without test code, the test after it: solver_kernels.hpp MPCMP_PERIOD)
    while (it < cfg.qp_iters) {
      const int period = MPCMP_PERIOD(cfg, it);
#pragma nounroll
      for (int kk = 0; kk < period; kk++) {
        int sio = tid;
        asm volatile("" : "+v"(sio));
        const int k0 = q5_lc<NSEG>(c, sio, 0), k1 = q5_lc<NSEG>(c, sio, 1), k2 = q5_lc<NSEG>(c, sio, 2), k3 = q5_lc<NSEG>(c, sio, 3);
        // ---- A: the variables of the last G wave ----
        if (laneV && Q5_ON(1)) {
            const int vi = L::NG + (sio - L::tV5);
            const double *vst = lds + L::vVst + (sio - L::tV5), *vcl = lds + L::vVc + vi;
            const double vx = ldv(vst), vzb = ldv(vst + 16), vyb = ldv(vst + 32), rb = ldv(vcl + 4 * L::NVL);
            const double wb = ldv(lds + L::vWv + lo16(pxr));
            const double r = (sigma * vx + (rb * vzb - vyb)) + q5_col_gather<NSEG>(lds, pxr, prf, prb, vcl, lds + L::vWg, lds + L::vGp);
            lds[hi16(pxr)] = r;                             // (T: the pad slot; its w of the border is zero)
            const double bp = sum8(wb * r);
            lds[L::vRedB + (vi >> 3)] = bp;
        }
        Q5B(0); Q5_BAR(0); Q5S(0);
        if (Q5_ON(2)) q5_p1<NSEG>(c, fm, sio, k0, k1, k2, k3);
        Q5B(1); Q5_BAR(1); Q5S(1);
        // ---- P2 (role S); this role is idle: the lane constants of phase E ----
        const int gro = q5_lc<NSEG>(c, sio, 4), xno = q5_lc<NSEG>(c, sio, 5);
        Q5B(2); Q5_BAR(2); Q5S(2);
        // ---- P3 (role E) ----
        Q5B(3); Q5_BAR(3); Q5S(3);
        // ---- E: the path rows; the last G wave's variables ----
        if (wavePath) {
            if (isPath && Q5_ON(5)) {
                q5_path_rows<NSEG>(lds, gro, xno, sio, lds + L::vXn, lds + L::vGp, [&](double zt, const PathOp5 &po) -> double {
                    double w = 0.0;
                    if (ownsRow) {
                        const double zr = alpha * zt + c.oma * zg;
                        const double zn = clip(zr + yg * po.rri, po.lgp, po.ugp);
                        yg += po.rr * (zr - zn);
                        zg = zn;
                        w = po.rr * zg - yg;
                    }
                    return w;
                });
            }
        } else if (Q5_ON(6)) {
            // (this wave runs two jobs one after the other and is the last one to reach the phase's barrier: the operands of the second job — the
            //  sixteen variables — are read before the first job's arithmetic, one LDS round trip instead of two in a row)
            const int vi = L::NG + ((laneV ? sio : L::tV5) - L::tV5);
            double *vst = lds + L::vVst + ((laneV ? sio : L::tV5) - L::tV5);
            const double *vcl = lds + L::vVc + vi;
            double vx = 0.0, vzb = 0.0, vyb = 0.0, xtv = 0.0, vlb = 0.0, vub = 0.0, rb = 0.0, rbi = 0.0;
            if (laneV) {
                vx = ldv(vst); vzb = ldv(vst + 16); vyb = ldv(vst + 32);
                xtv = ldv(lds + L::vXn + lo16(pxr)); vlb = ldv(vcl + 1 * L::NVL); vub = ldv(vcl + 2 * L::NVL);
                rb = ldv(vcl + 4 * L::NVL); rbi = ldv(vcl + 5 * L::NVL);
            }
            if (laneD) {        // dynamics rows DR0 .. meq - 1
                const int r = L::DR0 + (sio - L::tV5);
                const double *rcl = lds + L::vRc + r;
                const double lgd = ldv(rcl), rcT = ldv(rcl + L::NRC);
                const double zt = row_dot_dyn(lds + L::vXn, q5_l5<NSEG>(c, sio & 63, 3), L::oCD + 4 * ((prb >> 17) & 3), rcT);
                const double zr = alpha * zt + c.oma * ((it + kk > 0 || warm) ? lgd : 0.0);
                yg += rho_eq * (zr - lgd);                  // the row is an equality: the projection of anything onto [l, l] is l
                const double w = rho_eq * lgd - yg;
                lds[L::vWg + r] = w;
                const double tp = sum8(rcT * w);            // (ND5 is not a multiple of 8: the lanes behind the last row are masked off, their partial sum slot gets the sum of the group's rows from a lane that is not)
                lds[L::vRedT + (r >> 3)] = tp;
            }
            if (laneV) {
                vx = alpha * xtv + c.oma * vx;
                const double zrv = alpha * xtv + c.oma * vzb;
                const double znv = clip(zrv + vyb * rbi, vlb, vub);
                vyb += rb * (zrv - znv);
                vzb = znv;
                vst[0] = vx; vst[16] = vzb; vst[32] = vyb;
                if (vi == na) {     // the shared variable T: its state is published for the border solve and the tests
                    misc[L::M_xT] = vx; misc[L::M_zbT] = vzb; misc[L::M_ybT] = vyb;
                    misc[L::M_baseT] = (sigma * vx - 1.0) + (rb * vzb - vyb);
                }
            }
        }
        pxr = q5_l5<NSEG>(c, sio & 63, 0); prf = q5_l5<NSEG>(c, sio & 63, 1); prb = q5_l5<NSEG>(c, sio & 63, 2);
        Q5B(4); Q5_BAR(4); Q5S(4);
      }
      it += period;
      if (MPCMP_NO_TEST(cfg, period)) break;
      {
            int t = tid;
            asm volatile("" : "+v"(t));
            const int vi = laneV ? L::NG + (t - L::tV5) : 0;
            const bool isVar = laneV && vi < na;
            const double *vcl = lds + L::vVc + vi, *vst = lds + L::vVst + (laneV ? t - L::tV5 : 0);
            const double ha = isVar ? ldv(vcl + 3 * L::NVL) : 0.0, vx = ldv(vst), vzb = ldv(vst + 16), vyb = ldv(vst + 32);
            // the relaxed iterate and the duals of this wave's dynamics rows, for every role's test
            if (laneD) lds[L::vYs + L::DR0 + (t - L::tV5)] = yg;
            if (laneV) {
                if (vi == na) { for (int k = 0; k < N; k++) lds[L::vXx + k * XS + 21] = vx; }
                else lds[L::vXx + lo16(q5_l5<NSEG>(c, t & 63, 0))] = vx;
            }
            __syncthreads();
            // stage one: the primal residual
            double mp[3] = {0, 0, 0};
            double rcT5 = 0.0;
            const int gro2 = isPath ? q5_lc<NSEG>(c, t, 4) : 0, xno2 = isPath ? q5_lc<NSEG>(c, t, 5) : 0;
            if (isPath) {
                const double ax = q5_path_ax<NSEG>(lds, gro2, xno2, t, lds + L::vXx);
                if (ownsRow) { mp[0] = fabs(ax - zg); mp[1] = fabs(ax); mp[2] = fabs(zg); }
            }
            if (laneD) {
                rcT5 = ldv(lds + L::vRc + L::NRC + L::DR0 + (t - L::tV5));
                const double zgd = ldv(lds + L::vRc + L::DR0 + (t - L::tV5));
                const double ax = row_dot_dyn(lds + L::vXx, q5_l5<NSEG>(c, t & 63, 3), L::oCD + 4 * ((q5_l5<NSEG>(c, t & 63, 2) >> 17) & 3), rcT5);
                mp[0] = fabs(ax - zgd); mp[1] = fabs(ax); mp[2] = fabs(zgd);
            }
            if (isVar) { mp[0] = fmax(mp[0], fabs(vx - vzb)); mp[1] = fmax(mp[1], fabs(vx)); mp[2] = fmax(mp[2], fabs(vzb)); }
            if (q5_primal_ok<NSEG>(c, mp)) {
                // stage two: the dual residual
                double sums[2] = {0.0, 0.0}, md[3] = {0, 0, 0};
                if (isPath) {       // the path-row part of A^T y (read by the variable lanes after the barrier below)
                    const double ygc = yg;
                    q5_path_rows<NSEG>(lds, gro2, xno2, t, lds + L::vXx, lds + L::vGpy, [&](double, const PathOp5 &) -> double { return ownsRow ? ygc : 0.0; });
                    if (ownsRow) sums[0] = lds[lo16(gro2) - (xno2 % XS) + ((gro2 >> 16) & 1) * L::GS + 21] * yg;      // T coefficient of the owned row (column 21)
                }
                if (laneD) sums[0] = rcT5 * yg;
                if (isVar) sums[1] = ha * vx;
                __syncthreads();                                      // (gpy published)
                if (isVar) {
                    const double hx = (fabs(ha) + cfg.hess_reg) * vx + ha * lds[L::vXx + 21];
                    const double aty = q5_col_gather<NSEG>(lds, q5_l5<NSEG>(c, t & 63, 0), q5_l5<NSEG>(c, t & 63, 1), q5_l5<NSEG>(c, t & 63, 2), vcl, lds + L::vYs, lds + L::vGpy) + vyb;
                    md[0] = fabs(hx + aty); md[1] = fabs(hx); md[2] = fabs(aty);
                }
                done = q5_dual_ok<NSEG>(c, sums, md);
            }
            Q5S(5);
      }
      if (done) break;
    }
    const bool capped = !done;
    Q5_STAMP_DUMP(it);
    if (tid == 0) { c.ws.qpit[c.b] = it; c.ws.qp_total[c.b] += it; if (capped) atomicAdd(&c.ws.status[c.b], MPCMP_ST_CAP_ONE); }
    if (laneV) {
        const int vi = L::NG + (tid - L::tV5);
        const double *vst = lds + L::vVst + (tid - L::tV5);
        c.ws.p[(size_t)c.b * (na + 1) + vi] = vst[0];
        c.ws.y[(size_t)c.b * (D::ma + na + 1) + D::ma + vi] = vst[32];
    }
    if (ownsRow) c.ws.y[(size_t)c.b * (D::ma + na + 1) + D::meq + 8 * (tid >> 4) + 2 * ((tid & 15) >> 2) + (tid & 3)] = yg;
    if (laneD) c.ws.y[(size_t)c.b * (D::ma + na + 1) + L::DR0 + (tid - L::tV5)] = yg;
}

// ---- roles E and S: waves NSEG .. 11.  EP = waves NSEG .. wS0: P3 on the E lanes (the S lanes of the mixed wave wS0 take part in P2);
// !EP = the pure S waves: P2, and the last wave sums the T border's partial sums and replicates x~_T.  Every lane carries the variable
// vi = tid - NG, the lanes vi < meq also the dynamics row vi (k_qp2's role B). ----
template <int NSEG, bool EP>
__device__ __forceinline__ void qp5_role_es(const Qp5Ctx<NSEG> &c) {
    using L = Qp5<NSEG>;
    using D = Dim3<NSEG>;
    constexpr int meq = D::meq, XS = L::XS, NFM = EP ? 52 : 48;
    double *lds = c.lds;
    const mpcmp_config &cfg = *c.cfg;
    const int tid = c.tid, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Q5_STAMP_DECL;
    const bool laneS = !EP || tid >= L::tS0, laneE = !laneS;
    const bool waveS = !EP || wave >= L::wS0, waveX = !EP && wave == L::NWV - 1;
    // the lane's block without its tail column (LDS: vTS / vTE): S lanes 4 x 12 of the 4 x 13 block of S^-1, E lanes 4 x 13 of the 4 x 14 block of E_s
    double fm[NFM];
    {
        typedef const __attribute__((address_space(1))) double *gptr_t;
        if (laneS) {
            gptr_t p = (gptr_t)(c.fa + L::F::oFS + (tid - L::tS0));
#pragma unroll
            for (int pos = 0; pos < 4; pos++) {
#pragma unroll
                for (int j = 0; j < 13; j++) {
                    const double q = p[(13 * pos + j) * L::F::NSL];
                    if (j < 12) fm[12 * pos + j] = q; else lds[L::vTS + pos * L::NSL + (tid - L::tS0)] = q;
                }
            }
#pragma unroll
            for (int j = 48; j < NFM; j++) fm[j] = 0.0;
        } else {
            gptr_t p = (gptr_t)(c.fa + L::F::oFE + (tid - L::NG));
#pragma unroll
            for (int a = 0; a < 4; a++) {
#pragma unroll
                for (int j = 0; j < 14; j++) {
                    const double q = p[(14 * a + j) * L::F::ELS];
                    if (j < 13) fm[(13 * a + j) < NFM ? 13 * a + j : 0] = q; else lds[L::vTE + a * L::F::ELS + (tid - L::NG)] = q;
                }
            }
        }
    }
    const double (&fs)[48] = reinterpret_cast<const double (&)[48]>(fm);
    // ADMM state of the lane's variable and of its dynamics row (z of an equality row is its bound l from the first update on)
    double vx = 0.0, vzb = 0.0, vyb = 0.0, ygd = 0.0;
    const int vi0 = L::vi_of(tid);
    constexpr bool isDyn = !EP;         // (every lane of the pure S waves carries the dynamics row vi; the E waves carry none)
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
        const int k0 = q5_lc<NSEG>(c, t, 0), k1 = q5_lc<NSEG>(c, t, 1);
        __syncthreads();
        if (waveS) q5_p2<NSEG>(c, fs, t, k0, k1, laneS, false);
        __syncthreads();
        if (EP) q5_p3<NSEG>(c, reinterpret_cast<const double (&)[52]>(fm), t, k0, k1, laneE, false);
        else if (waveX) q5_p3_xT<NSEG>(c, t, false);
        __syncthreads();
    }
    q5_finish_border<NSEG>(c);
    Q5_STAMP_RESET;
    int it = 0, done = 0;
    int apx, apf, apb;          // constants of phase A, fetched ahead of the barrier that ends the previous iteration
    {
        int t = tid;
        asm volatile("" : "+v"(t));
        apx = q5_lc<NSEG>(c, t, 2); apf = q5_lc<NSEG>(c, t, 3); apb = q5_lc<NSEG>(c, t, 4);
    }
    const bool warm = cfg.qp_warm_start != 0;
    if (warm) {     // warm duals (mpcmp_config.qp_warm_start): y_0 = lambda_k, x_0 = 0, z_0 = clip(0, l, u); w_0 of the dynamics rows published
        const double *lamb = c.ws.lam + (size_t)c.b * (D::ma + D::na + 1);
        const double *vcl = lds + L::vVc + vi0;
        vyb = lamb[D::ma + vi0]; vzb = clip(0.0, vcl[1 * L::NVL], vcl[2 * L::NVL]);
        if (isDyn) {
            const double lgd = lds[L::vRc + vi0], rcT = lds[L::vRc + L::NRC + vi0];
            ygd = lamb[vi0];
            const double w = rho_eq * lgd - ygd;
            lds[L::vWg + vi0] = w;
            const double tp = sum8(rcT * w);
            lds[L::vRedT + (vi0 >> 3)] = tp;
        }
        __syncthreads();
    }
    while (it < cfg.qp_iters) {
      const int period = MPCMP_PERIOD(cfg, it);
#pragma nounroll
      for (int kk = 0; kk < period; kk++) {
        int sio = tid;
        asm volatile("" : "+v"(sio));
        const int vi = L::vi_of(sio);
        // ---- A: rhs = sigma x - q + rho_b z_b - y_b + A^T w; partial sums of w^T rhs ----
        if (Q5_ON(1)) {
            const double *vcl = lds + L::vVc + vi;
            const double rb = ldv(vcl + 4 * L::NVL);
            const double wb = ldv(lds + L::vWv + lo16(apx));
            const double r = (sigma * vx + (rb * vzb - vyb)) + q5_col_gather<NSEG>(lds, apx, apf, apb, vcl, lds + L::vWg, lds + L::vGp);
            lds[hi16(apx)] = r;
            const double bp = sum8(wb * r);
            lds[L::vRedB + (vi >> 3)] = bp;               // (all eight lanes of a group hold the sum and store it)
        }
        Q5B(0); Q5_BAR(0); Q5S(0);
        // ---- P1 (role G); the last wave sums the border's partial sums ----
        if (waveX && Q5_XT) q5_p1_xT<NSEG>(c, sio);
        const int k0 = q5_lc<NSEG>(c, sio, 0), k1 = q5_lc<NSEG>(c, sio, 1);
        Q5B(1); Q5_BAR(1); Q5S(1);
        if (waveS && Q5_ON(3)) q5_p2<NSEG>(c, fs, sio, k0, k1, laneS, true);
        Q5B(2); Q5_BAR(2); Q5S(2);
        const int epx = q5_lc<NSEG>(c, sio, 2), epb = isDyn ? q5_lc<NSEG>(c, sio, 4) : 0, eix = isDyn ? q5_lc<NSEG>(c, sio, 5) : 0;      // (constants of phase E: in flight during the product)
        if (EP && Q5_ON(4)) q5_p3<NSEG>(c, reinterpret_cast<const double (&)[52]>(fm), sio, k0, k1, laneE, true);
        else if (waveX && Q5_XT) q5_p3_xT<NSEG>(c, sio, true);
        Q5B(3); Q5_BAR(3); Q5S(3);
        // ---- E: the variable and the dynamics row of the lane ----
        apx = q5_lc<NSEG>(c, sio, 2); apf = q5_lc<NSEG>(c, sio, 3); apb = q5_lc<NSEG>(c, sio, 4);      // (constants of the next phase A: not behind this phase's work)
        if (Q5_ON(6)) {
            const double *vcl = lds + L::vVc + vi, *rcl = lds + L::vRc + (isDyn ? vi : 0);
            const int xpos = lo16(epx);
            const double xtv = ldv(lds + L::vXn + xpos), vlb = ldv(vcl + 1 * L::NVL), vub = ldv(vcl + 2 * L::NVL);
            const double rb = ldv(vcl + 4 * L::NVL), rbi = ldv(vcl + 5 * L::NVL);
            if (isDyn) {
                const double lgd = ldv(rcl), rcT = ldv(rcl + L::NRC);
                const double zt = row_dot_dyn(lds + L::vXn, eix, L::oCD + 4 * ((epb >> 17) & 3), rcT);
                const double zr = alpha * zt + c.oma * ((it + kk > 0 || warm) ? lgd : 0.0);
                ygd += rho_eq * (zr - lgd);                 // the row is an equality: the projection of anything onto [l, l] is l
                const double w = rho_eq * lgd - ygd;
                lds[L::vWg + vi] = w;
                const double tp = sum8(rcT * w);
                lds[L::vRedT + (vi >> 3)] = tp;
            }
            vx = alpha * xtv + c.oma * vx;
            const double zrv = alpha * xtv + c.oma * vzb;
            const double znv = clip(zrv + vyb * rbi, vlb, vub);
            vyb += rb * (zrv - znv);
            vzb = znv;
        }
        Q5B(4); Q5_BAR(4); Q5S(4);
      }
      it += period;
      if (MPCMP_NO_TEST(cfg, period)) break;
      {
            int t = tid;
            asm volatile("" : "+v"(t));
            const int vc_ = L::vi_of(t);
            const double *vcl = lds + L::vVc + vc_, *rcl = lds + L::vRc + (isDyn ? vc_ : 0), *xx = lds + L::vXx;
            const int pxc = q5_lc<NSEG>(c, t, 2), prfc = q5_lc<NSEG>(c, t, 3), prbc = q5_lc<NSEG>(c, t, 4), ixc = q5_lc<NSEG>(c, t, 5);
            // the relaxed iterate and the dual of the lane's dynamics row, for every role's test
            if (isDyn) lds[L::vYs + vc_] = ygd;
            lds[L::vXx + lo16(pxc)] = vx;
            __syncthreads();
            const double ha = ldv(vcl + 3 * L::NVL);
            const double rcT = isDyn ? ldv(rcl + L::NRC) : 0.0, zgd = isDyn ? ldv(rcl) : 0.0;
            // stage one: the primal residual
            double mp[3] = {0, 0, 0};
            if (isDyn) {
                const double ax = row_dot_dyn(xx, ixc, L::oCD + 4 * ((prbc >> 17) & 3), rcT);
                mp[0] = fabs(ax - zgd); mp[1] = fabs(ax); mp[2] = fabs(zgd);
            }
            mp[0] = fmax(mp[0], fabs(vx - vzb)); mp[1] = fmax(mp[1], fabs(vx)); mp[2] = fmax(mp[2], fabs(vzb));
            if (q5_primal_ok<NSEG>(c, mp)) {
                // stage two: the dual residual
                double sums[2] = {isDyn ? rcT * ygd : 0.0, ha * vx}, md[3];
                __syncthreads();                                      // (gpy published)
                const double hx = (fabs(ha) + cfg.hess_reg) * vx + ha * xx[21], aty = q5_col_gather<NSEG>(lds, pxc, prfc, prbc, vcl, lds + L::vYs, lds + L::vGpy) + vyb;
                md[0] = fabs(hx + aty); md[1] = fabs(hx); md[2] = fabs(aty);
                done = q5_dual_ok<NSEG>(c, sums, md);
            }
            Q5S(5);
      }
      if (done) break;
    }
    Q5_STAMP_DUMP(it);
    c.ws.p[(size_t)c.b * (D::na + 1) + vi0] = vx;
    c.ws.y[(size_t)c.b * (D::ma + D::na + 1) + D::ma + vi0] = vyb;
    if (isDyn) c.ws.y[(size_t)c.b * (D::ma + D::na + 1) + vi0] = ygd;
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
    c.tsT = tsT; c.rho_in = rho_in; c.rho_eq = rho_eq; c.sigma = sigma; c.alpha = alpha; c.oma = q5_uniform(1.0 - alpha);
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
    int f[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // 0..3: the solve, 4..7: pxr, prf, prb, ixr of the lane's variable / dynamics row (G lanes: 4, 5 = gro, xno of the path rows)
    if (tid >= L::NG || (tid >= L::tV5 && tid < L::tV5 + 16)) {          // the variable of this lane
        const int vi = tid >= L::NG ? L::vi_of(tid) : L::NG + (tid - L::tV5);
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
    {                           // the dynamics row of this lane: rows 0 .. DR0 - 1 on the pure S lanes, the rest on the first lanes of the last G wave
        const int r = tid >= L::tSP ? tid - L::tSP : (tid >= L::tV5 && tid < L::tV5 + L::ND5 ? L::DR0 + (tid - L::tV5) : -1);
        if (r >= 0) {
            const int k = r / 14, rr = r % 14;
            lds[L::vRc + r] = -ws.ceq[(size_t)b * meq + r];
            lds[L::vRc + L::NRC + r] = coef_T(r);
            f[6] |= (k % 3) << 17;
            f[7] = (3 * (k / 3) * XS + rr) | ((k * XS + (rr < 7 ? 7 + rr : 14 + rr - 7)) << 16);
        }
    }
    if (tid < L::tV5) {         // path rows: four lanes per pair of rows, sixteen lanes per node
        const int pk = tid >> 4, prp = (tid & 15) >> 2, pq = tid & 3, q = 2 * prp + pq;
        double lg = 0.0, ug = 0.0;
        f[4] = L::oGk; f[5] = NX;                              // (no row: valid operands, nothing is stored)
        if (tid < L::NPN) {
            if (pq < 2) {
                const double gv = ws.g[(size_t)b * 8 * N + 8 * pk + q];
                lg = c_lbg[q] - gv; ug = c_ubg[q] - gv;
            }
            // bits 0..15: Jacobian operand (columns 6 pq .. 6 pq + 5 of the 24-wide padded rows 2 prp, 2 prp + 1 of node pk); 16: parity; 17: publishes gp
            f[4] = (L::oGk + (pk * 8 + 2 * prp) * L::GS + pq * 6) | ((pq & 1) << 16) | ((prp == 0 ? 1 : 0) << 17);
            f[5] = pk * XS + pq * 6;
        }
        const double rr = (ug - lg < 1e-4) ? rho_eq : rho_in;
        lds[L::vPc + tid] = lg; lds[L::vPc + L::NPL + tid] = ug; lds[L::vPc + 2 * L::NPL + tid] = rr; lds[L::vPc + 3 * L::NPL + tid] = 1.0 / rr;
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
        int *lct = reinterpret_cast<int *>(lds + L::vLCT), *t5 = reinterpret_cast<int *>(lds + L::vL5);
        if (tid >= L::tV5 && tid < L::NG) {             // last G wave: the variable / dynamics-row words go to their own small table
#pragma unroll
            for (int q = 0; q < 4; q++) { t5[q * 64 + (tid - L::tV5)] = f[4 + q]; f[4 + q] = q < 2 ? (q == 0 ? L::oGk : NX) : 0; }
        }
        if (tid >= L::NG) { f[2] = f[4]; f[3] = f[5]; f[4] = f[6]; f[5] = f[7]; }      // E / S lanes: 0, 1 the solve, 2..5 pxr, prf, prb, ixr
#pragma unroll
        for (int q = 0; q < L::NF; q++) lct[q * NT + tid] = f[q];
    }
    __syncthreads();
    for (int ip = tid; ip < na; ip += NT) lds[rhs_slot(ip)] = lds[L::vKT + ip];      // rhs of K_0 w = k
    __syncthreads();
    if (wave < NSEG) qp5_role_g<NSEG>(c);
    else if (wave <= L::wS0) qp5_role_es<NSEG, true>(c);
    else qp5_role_es<NSEG, false>(c);
}

}  // namespace mpcmp
