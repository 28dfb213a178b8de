// qp_kernel_v3.hpp — k_qp3f<NSEG, NARM> + k_qp3<NSEG, NARM>: the QP of the larger discretisations (N = 19, the reference as
// shipped, robot_ocp.hpp:31-32; N = 25) and of multi-arm robots (BASELINE.json configs[3]: 14-DoF dual Panda, N = 25).
//
// One workgroup per (OCP, arm).  Same arithmetic as k_qp / k_qp2 — OSQP-form ADMM on [A; I] with the reduced KKT system solved
// by nested dissection and explicit block inverses — with two structural changes that make the factor of an N = 25 arm fit one
// CU's registers (structure3.hpp):
//   * the final time T, the only variable the arms share, is bordered out: every arm factorises its own K_0 and the arms of one
//     OCP exchange ONE scalar per ADMM iteration (k_a^T K_0a^-1 b_a) through global memory (NARM = 2; nothing for NARM = 1);
//   * E_s = G_s K_JC is never formed: the interior solve applies G_s twice around the sparse K_JC / K_CJ products, so the only
//     dense factors are G_s (49 x 49 per segment) and S^-1 (nI x nI).
// k_qp3f (1024 threads) assembles and factorises K_0 and leaves the factor in a per-problem workspace, in the lane layout of
// k_qp3 (512 threads = 8 waves, two per SIMD, 256 registers per lane), which runs the ADMM loop: a lane keeps a 4 x 13 block of
// G_s (wave s = segment s; the four lanes of a quad share four rows, lanes 56..62 of the last segment's wave hold the 7 x 7 block
// of u_{N-1}), a 4 x SC block of S^-1 (sixteen lanes share four rows) and the ADMM state of one variable and one row in
// registers.  Every wave takes part in every phase; five workgroup barriers per iteration:
//     A  rhs = sigma x - q + rho z - y + A^T w    P1  t = G b_J,  K_CJ t    P3  y_I = S^-1 (b_I - K_CJ t)
//     P4 x_J = G (b_J - K_JC y_I) - w x_T         E   z~ = A x~, relaxation, projection, dual update
// DESIGN.md (section 4) has the measurements behind every choice made here.
#pragma once
#include "qp_kernel_v2.hpp"
#include "structure3.hpp"

namespace mpcmp {

// inter-workgroup exchange between the two arm workgroups of one OCP (NARM = 2): 8-byte data-tagged granules, agent-scope
// relaxed atomics on both sides (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility": valid
// form for <= 8-byte hand-offs, placement independent).  Every slot is written once per launch; k_init_m / k_step_m reset the
// slots of their problem to XCH_EMPTY before the next QP launch.  Every spin is bounded.
#define MPCMP_XCH_STRIDE 1024
#define MPCMP_XCH_EMPTY 0x7FF8DEADBEEF0001ull          /* a NaN payload no arithmetic produces */
struct Xch {
    unsigned long long *buf;        // [B][2][MPCMP_XCH_STRIDE]
};
__device__ __forceinline__ void xch_post(unsigned long long *slot, double v) {
    __hip_atomic_store(slot, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double xch_poll(const unsigned long long *slot, int &dead) {
    unsigned long long v = MPCMP_XCH_EMPTY;
    if (!dead) {
        for (int spin = 0; spin < (1 << 21); spin++) {
            v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != MPCMP_XCH_EMPTY) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (v == MPCMP_XCH_EMPTY) dead = 1;          // partner never arrived: give up on every later exchange as well
    }
    return dead ? 0.0 : __longlong_as_double((long long)v);
}

template <int NSEG>
struct Qp3 {
    using D = Dim3<NSEG>;
    static constexpr int NT = 1024;
    static constexpr int GS = 23;                                   // row stride of the path Jacobians in LDS: odd, so that the 32 lanes of an 8-byte read group (lane = row) hit 32 different bank pairs (22 gave a two-way conflict on every coefficient read)
    static constexpr int NS = GS;                                   // node stride of x~, w (border), w = rho z - y in the loop kernel: [x_k (14) | u_k (7) | x~_T] resp. [dynamics rows (14) | path rows (8)]
    static constexpr int NX = NS * D::N, NXP = (NX + 2 + 1) / 2 * 2; // (slot NX: pad for lanes without a job)
    static constexpr int JS = 52;                                   // stride of one segment's part of rhs (49 + zero pad: 4 x 13 column groups)
    static constexpr int SC = (D::nI + 15) / 16;                    // columns of S^-1 per lane (16 lanes per group of four rows)
    static constexpr int RIW = 128;                                 // padded length of r_I (>= 16 SC, two passes of a wave)
    static_assert(16 * SC <= RIW && D::nI <= RIW, "r_I padding");
    static constexpr int NAP = (D::na + 2 + 1) / 2 * 2;             // na + T slot, even
    static constexpr int MAP = (D::ma + 1) / 2 * 2;
    static constexpr int NB = NSEG + 1;                             // blocks of the interior sweep (segments + padded K_UU)
    static constexpr int CB = NB * 52 > 128 ? NB * 52 : 128;        // pivot-column buffer of the sweeps
    static constexpr int e2(int x) { return (x + 1) / 2 * 2; }
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    // ---- common part, both kernels (doubles) ----
    static constexpr int oGk = 0;                                   // [N][8][GS]    path Jacobians
    static constexpr int oMisc = oGk + D::N * 8 * GS;               // [32]  see M_* below
    static constexpr int oCD = oMisc + 32;                          // [16]  differentiation matrix, [16] zeros
    static constexpr int oRedP = oCD + 32;                          // [160] workgroup reductions
    static constexpr int oCfg = oRedP + 160;                        // [64] bound tables of the configuration: lbx 0, ubx 14, lbu 28, ubu 35, lbg 42, ubg 50
    static constexpr int oPat = oCfg + 64;                          // [49 + 28 + 28] ints: sparse K_JC pattern words (jc | cjl | cjh)
    static constexpr int oPE = oPat + 54;                           // end of the common part
    static constexpr int KJN = NSEG * 196 + 4;                      // sparse K_JC [NSEG][49][4] (canonical slots, structure3.hpp) + a zero row
    // ---- factorisation kernel ----
    static constexpr int oKJC = oPE;                                // [KJN]
    static constexpr int oKUX = oKJC + KJN;                         // [NSEG][7][14] dense block u_3s x x_3s
    static constexpr int oKuX = oKUX + NSEG * 98;                   // [7][14]       u_{N-1} x x_{N-1}
    static constexpr int oGu = oKuX + 98;                           // [28]          -(K_UU^-1) packed
    static constexpr int oKT = oGu + 32;                            // [na] T column k (internal order), [na] kappa_a
    static constexpr int oCT = oKT + NAP;                           // [meq]         T coefficient -ts*f of the dynamics rows
    static constexpr int fKJJ = oCT + e2(D::meq);                   // [NSEG][1225], later S packed [SP]
    static constexpr int fKUU = fKJJ + NSEG * D::JP;                // [28]
    static constexpr int fZ = fKUU + 32;                            // [na] iterate of this arm (assembly operands)
    static constexpr int fCol = fZ + NAP;                           // [2][CB]
    static constexpr int fRdv = fCol + 2 * CB;                      // [2][16]
    static constexpr int fEnd = fRdv + 32;
    static constexpr int fSW = fKJJ + e2(D::SP);                    // Schur phase scratch behind S: [8][64] column, [8][64] product
    static_assert(fSW + 512 + 2 * 416 <= fKUU, "Schur scratch must fit between S and the K_UU block");
    // ---- loop kernel ----
    static constexpr int lKJC = oPE;                                // [KJN] sparse K_JC, row form
    static constexpr int lKT = lKJC + KJN;                          // [na] T column k (internal order), [na] kappa_a
    static constexpr int oKCJ = lKT + NAP;                          // [NSEG][8][28] column form of the sparse K_JC: entry d of C-column c sits in row c % 14 + 7 (d - 1); d-major, so that lanes = columns read neighbouring words
    static constexpr int oKUXT = oKCJ + NSEG * 224;                 // [NSEG + 1][14][8] dense blocks transposed (column c: 7 entries + pad), last: u_{N-1} x x_{N-1}
    static constexpr int oKUXP = oKUXT + (NSEG + 1) * 112;          // [NSEG + 1][7][16] dense blocks, rows padded to 16
    static constexpr int oZR = oKUXP + (NSEG + 1) * 112;            // [16] zeros
    static constexpr int DER = NSEG * 224 + 2 * (NSEG + 1) * 112 + 16;   // (doubles of the derived copies)
    static constexpr int oLb = oKCJ + DER, oUb = oLb + NAP, oWv = oUb + NAP;     // variable constants (rho: a flag in the lane's descriptor word); w of the T border (node order)
    static constexpr int oLg = oWv + NXP, oUg = oLg + MAP, oCf = oUg + MAP;   // row constants
    static constexpr int oRpos = oCf + MAP;                         // [na] ints: LDS slot of the variable's rhs entry
    static constexpr int oRhsJ = oRpos + e2((D::na + 1) / 2);       // [NSEG][JS]; at the termination tests [oRhsJ, oRhsJ + NX): the duals of the rows in node order
    static_assert(NSEG * JS + JS + RIW >= NX, "dual vector of the termination test must fit over the rhs vectors");
    static constexpr int oRhsU = oRhsJ + NSEG * JS;                 // [JS] (7 used, rest zero)
    static constexpr int oRhsI = oRhsU + JS;                        // [RIW] (zero beyond nI)
    static constexpr int TS = 64;                                   // wave-private vector of a G wave: [0..51] operand / result, [56] dummy
    static constexpr int oTJ = oRhsI + RIW;                   // [8][TS], then the U block of the last segment's wave [TS]
    static constexpr int oTU = oTJ + 8 * TS;
    static constexpr int oDW = oTU + TS;                            // [8][16] wave-private: dense parts of the rows u_3s ([0..6]) and of the U block ([7..13])
    // K_CJ t, indexed like r_I (interface entry i = 14 node + component), so that r_I = b_I - PA - PB - DP is straight-line code:
    static constexpr int oPA = oDW + 128;                           // [RIW] sparse part from the segment the node opens (x_3s); node NSEG: the U block
    static constexpr int oPB = oPA + RIW;                           // [RIW] sparse part from the segment the node closes (x_3s+3)
    static constexpr int oDP = oPB + RIW;                           // [RIW] dense part (columns x_3s)
    static constexpr int oPD = oDP + RIW;                           // [2] pad slot for lanes without an entry
    static constexpr int oRIw = oPD + 2;                            // [RIW] r_I: ONE copy that every wave writes in full (identical values: no hand-off between waves)
    static constexpr int oYI = oRIw + RIW;                          // [nI] + pad slot
    static constexpr int oXt = oYI + e2(D::nI + 2);                 // [NXP] x~ in node order
    static constexpr int oWg = oXt + NXP;                           // [NXP] w = rho z - y in node order
    static constexpr bool P8 = NSEG < 8;                            // N = 19: partial sums per 8 lanes (half the DPP chain in A and E); wave 7, idle in P1, adds them up
    static constexpr int oRedB = oWg + NXP;                         // [64] partial sums of w^T rhs (per wave, or per 8 lanes)
    static constexpr int oRedT = oRedB + 64;                        // [64] partial sums of the T column of A^T w
    static constexpr int oS1 = oRedT + 64;                          // ADMM state of the second variable ([3][32]) / row ([2][96]) of the lanes that own two (N = 25)
    static constexpr bool LCT = NSEG < 8;                           // lane-constant table (N = 25: measured slower with it — the register allocator answers with copies and spills)
    static constexpr int NLC = 21;                                  // fields of LaneC1 + LaneC3 + LaneC4
    static constexpr int oLCT = oS1 + 96 + 2 * 96;                  // [NLC][512] 16-bit words   (oS1: [3][32] second variables, [2][96] second rows)
    static constexpr int oXdG = oLCT;                               // [8][64] ints, set-up only (before the table is filled): where the G lanes' x~ entries go
    static constexpr bool STL = false;                                    // (option: z_b, y_b, z_g, y_g of the lane's first variable / row in LDS; measured: no gain at N = 25)
    static constexpr int oSt = oLCT + (LCT ? NLC * 128 : 256);      // [4][512]
    static constexpr int oHa = oSt + (STL ? 4 * 512 : 0);           // [na] Hessian arrow entries h_a (termination test)
    static constexpr bool HAL = !STL;                               // (with the state in LDS there is no room for it: h_a is then re-derived at the tests)
    static constexpr int lEnd = oHa + (HAL ? NAP : 0);
    static constexpr int sizeF = fEnd, sizeL = lEnd;
    static_assert(sizeF * 8 <= 160 * 1024 - 512 && sizeL * 8 <= 160 * 1024 - 512, "LDS budget");
    // factor workspace (doubles per arm): the sparse K_JC [KJN], the T column [NAP], sum|ha| [8], the derived copies [DER], then
    // the 4 x 13 blocks of G [8][52][64], then the 4 x SC blocks of S^-1 [4 SC][512] (both in the loop kernel's lane layout,
    // see g_blk / s_blk)
    static constexpr int oFT = KJN, oFH = oFT + NAP, AUX = oFH + 8;
    static constexpr int oFD = AUX, oFG = oFD + DER, oFS = oFG + 8 * 52 * 64;
    static constexpr int FAC = oFS + 4 * SC * 512;
    // misc slots
    static constexpr int M_xT = 0, M_zbT = 1, M_ybT = 2, M_baseT = 3, M_delta = 4, M_hdT = 5, M_rbT = 6, M_lbT = 7, M_ubT = 8,
                         M_xtT = 9, M_sumha = 10, M_mtsT = 11 /* -ts T */, M_dl = 12, M_done = 13, M_s0 = 14, M_s1 = 15, M_c0 = 16 /* 16..31: check exchange */;
};

// factor workspace of k_qp3f<NSEG, 1, 4> -> k_qp4 (qp_kernel_v4.hpp), doubles per problem
template <int NSEG>
struct Qp4Fac {
    using Q3 = Qp3<NSEG>;
    static constexpr int SRS = 70;                                  // row stride of S^-1 (full, both triangles)
    static constexpr int KX = 44;                                   // row stride of the column form of K_JC
    static constexpr int fKJC = 0;                                  // [KJN] sparse K_JC, row form (canonical slots) + a zero row
    static constexpr int fKT = Q3::oFT, fH = Q3::oFH;               // T column k (internal order) + kappa; sum |ha|
    static constexpr int fKX = Q3::AUX;                             // [NSEG + 1][7][KX]
    static constexpr int fGu = fKX + (NSEG + 1) * 7 * KX;           // [7][8]  K_UU^-1
    static constexpr int fS = fGu + 56;                             // [nI][SRS]
    static constexpr int fG48 = fS + Q3::D::nI * SRS;               // [NSEG][56] row 48 of every G_s
    static constexpr int fG = fG48 + NSEG * 56;                     // [28][384]  2 x 14 blocks of rows 0..47, lane layout of k_qp4
    static constexpr int FAC = fG + 28 * 384;
};

// factor workspace of k_qp3f<NSEG, 1, 5> -> k_qp5 (qp_kernel_v5.hpp), doubles per problem: Qp3's up to the G blocks, then S^-1 in
// 4 x 13 blocks (eight lanes per group of four rows: lane ls = 8 gs + c8 keeps rows 4 gs + ((c8 & 3) ^ pos), columns 13 c8 .. + 12,
// entry 13 pos + j at [entry][NSL lanes]), then E_s = G_s K_JC,s in 4 x 14 blocks (two lanes per group of four rows: lane
// le = 26 s + 2 g + h keeps rows 4 g + (a ^ 2 h), a = 0..3, columns 14 h .. + 13 of E_s, entry 14 a + j at [entry][ELS lanes];
// "segment" NSEG is E_u = G_u K_UX, 7 x 14; unused lanes hold zeros)
template <int NSEG>
struct Qp5Fac {
    using Q3 = Qp3<NSEG>;
    static constexpr int NSL = 8 * ((Q3::D::nI + 3) / 4);           // S lanes (200 at N = 19)
    static constexpr int NEL = 26 * NSEG + 4;                       // E lanes in use (160)
    static constexpr int ELS = (768 - NSL - 64 * NSEG + 31) / 32 * 32;   // lane stride of the E blocks: every lane of k_qp5 between the G waves and the S lanes (192)
    static constexpr int oFS = Q3::oFS;                             // [52][NSL]
    static constexpr int oFE = Q3::FAC;                             // [56][ELS]
    static constexpr int FAC = oFE + 56 * ELS;
    static_assert(52 * NSL <= 4 * Q3::SC * 512, "the S^-1 blocks of k_qp5 must fit the area of k_qp3's");
};

__device__ __forceinline__ void wave_sync() {
    // LDS traffic of one wave is executed in order; this only keeps the compiler from moving accesses across the hand-off
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// row . operand for a G lane: broadcast LDS reads of the operand in chunks of three 16-byte pairs, so that the scheduler cannot
// hoist all 25 reads in front of the FMAs (the row itself already takes 98 of the 128 VGPRs)
// workgroup reduction (sum or max) of K values per thread by DPP inside the waves (no LDS permutes: 24 ds_bpermute per value in the
// generic block_reduce) and one LDS exchange between them; the result is valid in every thread.  Two barriers.
template <int NW, int K, bool MAX>
__device__ __forceinline__ void block_reduce_dpp(double (&v)[K], double *red, int tid) {
#pragma unroll
    for (int k = 0; k < K; k++) {
        double x = v[k];
        if (MAX) {
            x = fmax(x, dpp_mov<0xB1>(x)); x = fmax(x, dpp_mov<0x4E>(x)); x = fmax(x, dpp_mov<0x141>(x)); x = fmax(x, dpp_mov<0x140>(x));
            x = fmax(fmax(x, read_lane(x, 16)), fmax(read_lane(x, 32), read_lane(x, 48)));
        } else x = wave_sum(x);
        v[k] = x;                                   // valid in lanes 0..15
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[(tid >> 6) * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        double x = red[k];
#pragma unroll
        for (int w = 1; w < NW; w++) x = MAX ? fmax(x, red[w * K + k]) : x + red[w * K + k];
        v[k] = x;
    }
    __syncthreads();
}

// Workgroup reduction (sum or max) of K values per thread for the termination tests, with a small register footprint (the lanes carry
// their factor blocks; block_reduce_dpp's second stage reads NW x K partials per thread, which the compiler keeps all in flight): DPP inside the waves, one LDS slot per (wave, k), then the sixteen lanes of a DPP row combine the NW partials
// of one k (k_qp2's scheme); the result is valid in every thread.  Two barriers; `red` (>= (NW + 1) K doubles) must not be shared
// with a reduction issued right before or after.
template <int NW, int K, int KM>
__device__ __forceinline__ void block_reduce_lean(double (&v)[K], double *red, int tid) {      // values 0 .. KM - 1: maxima (of magnitudes), KM .. K - 1: sums
    static_assert(NW <= 16 && 16 * K <= 64 * NW, "one DPP row per value");
#pragma unroll
    for (int k = 0; k < K; k++) {
        double x = v[k];
        if (k < KM) {
            x = fmax(x, dpp_mov<0xB1>(x)); x = fmax(x, dpp_mov<0x4E>(x)); x = fmax(x, dpp_mov<0x141>(x)); x = fmax(x, dpp_mov<0x140>(x));
            x = fmax(fmax(x, read_lane(x, 16)), fmax(read_lane(x, 32), read_lane(x, 48)));
        } else x = wave_sum(x);
        if ((tid & 63) == 0) red[(tid >> 6) * K + k] = x;
    }
    __syncthreads();
    if (tid < 16 * K) {
        const int w = tid & 15, k = tid >> 4;
        double a = w < NW ? red[w * K + k] : 0.0;             // (0: identity of both reductions)
        if (k < KM) { a = fmax(a, dpp_mov<0xB1>(a)); a = fmax(a, dpp_mov<0x4E>(a)); a = fmax(a, dpp_mov<0x141>(a)); a = fmax(a, dpp_mov<0x140>(a)); }
        else { a += dpp_mov<0xB1>(a); a += dpp_mov<0x4E>(a); a += dpp_mov<0x141>(a); a += dpp_mov<0x140>(a); }
        if (w == 0) red[NW * K + k] = a;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = red[NW * K + k];
}

// diagnostic builds (-DMPCMP_STAMPS, tools/stamps3.py): cycles per phase of the ADMM loop (QS: after a barrier) and the busy part of
// each phase per wave (QB: in front of the barrier)
#ifdef MPCMP_STAMPS
#define QS(k) do { const unsigned long long n_ = clock64(); st_acc[k] += n_ - st_t; st_t = n_; } while (0)
#define QB(k) do { st_busy[k] += clock64() - st_t; } while (0)
#define QM(k) do { const unsigned long long n_ = clock64(); st_busy[k] += n_ - st_m; st_m = n_; } while (0)     /* sub-phase marks */
#define QM0() do { st_m = clock64(); } while (0)
#else
#define QM(k) do { } while (0)
#define QM0() do { } while (0)
#define QS(k) do { } while (0)
#define QB(k) do { } while (0)
#endif

// LDS reads of the hot loop.  On gfx950 a ds_read_b64 occupies the LDS array for 2 cycles and a ds_read_b128 for 4, but a
// ds_read2_b64 for 8 (MI355X_MICROARCH.md, LDS): the pairs the compiler forms out of neighbouring 8-byte reads cost twice what the
// two single reads do.  Volatile reads in the LDS address space are neither paired nor re-ordered among themselves: they are issued
// in program order, all in flight, and the compiler waits for each with a counted lgkmcnt just before its first use.
typedef const volatile __attribute__((address_space(3))) double *vlds_t;
typedef double v2d __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) v2d *vlds2_t;
__device__ __forceinline__ double ldv(const double *p) { return *(vlds_t)p; }
__device__ __forceinline__ v2d ldv2(const double *p) { return *(vlds2_t)p; }       // p: 16-byte aligned

// Block form of the same product for the loop kernel: the four lanes of a quad share four rows of G_s; lane (quad g, m) holds the
// 4 x 13 block of rows 4g + (m ^ pos), pos = 0..3, and columns 13m .. 13m + 12, reads only ITS 13 operand entries (a row per
// lane needs all 49 through LDS broadcasts), and a reduce-scatter over the quad (three DPP exchanges; the row order m ^ pos makes
// them select-free) leaves row 4g + m in lane 4g + m.
template <bool TWO_BATCHES = false>
__device__ __forceinline__ double g_blk(const double (&m)[52], const double *op) {
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
    if (!TWO_BATCHES) {
        double o[13];
#pragma unroll
        for (int j = 0; j < 13; j++) o[j] = ldv(op + j);
#pragma unroll
        for (int j = 0; j < 13; j++) { p0 += m[j] * o[j]; p1 += m[13 + j] * o[j]; p2 += m[26 + j] * o[j]; p3 += m[39 + j] * o[j]; }
    } else {                                        // (N = 25: every register counts)
        double o[7];
#pragma unroll
        for (int j = 0; j < 7; j++) o[j] = ldv(op + j);
#pragma unroll
        for (int j = 0; j < 7; j++) { p0 += m[j] * o[j]; p1 += m[13 + j] * o[j]; p2 += m[26 + j] * o[j]; p3 += m[39 + j] * o[j]; }
#pragma unroll
        for (int j = 0; j < 6; j++) o[j] = ldv(op + 7 + j);
#pragma unroll
        for (int j = 0; j < 6; j++) { p0 += m[7 + j] * o[j]; p1 += m[20 + j] * o[j]; p2 += m[33 + j] * o[j]; p3 += m[46 + j] * o[j]; }
    }
    const double q0 = p0 + dpp_mov<0x4E>(p2), q1 = p1 + dpp_mov<0x4E>(p3);          // lanes m, m ^ 2
    return q0 + dpp_mov<0xB1>(q1);                                                  // lanes m, m ^ 1
}
// S^-1: sixteen lanes share four rows; lane (group g, c = 4a + m) holds rows 4g + (m ^ pos) x columns SC c .. SC c + SC - 1.
// Quad reduce-scatter as above, then the four quads of the 16-lane DPP row are summed (row_ror 4, 8): every lane of the group
// ends with the total of row 4g + m.
template <int SC>
__device__ __forceinline__ double s_blk(const double (&m)[4 * SC], const double *op) {
    double o[SC];
#pragma unroll
    for (int j = 0; j < SC; j++) o[j] = ldv(op + j);
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
    for (int j = 0; j < SC; j++) { p0 += m[j] * o[j]; p1 += m[SC + j] * o[j]; p2 += m[2 * SC + j] * o[j]; p3 += m[3 * SC + j] * o[j]; }
    const double q0 = p0 + dpp_mov<0x4E>(p2), q1 = p1 + dpp_mov<0x4E>(p3);
    double x = q0 + dpp_mov<0xB1>(q1);
    x += dpp_mov<0x124>(x);                                                         // row_ror:4
    return x + dpp_mov<0x128>(x);                                                   // row_ror:8
}

#define QP3_PROLOGUE(NT_, FILL_CT_) QP3_PROLOGUE_L(Qp3<NSEG>, NT_, FILL_CT_)
// Logical thread index.  The 768-thread kernel (k_qp5) may run its logical waves on other hardware waves: hardware wave w runs on SIMD w % 4 and the roles
// are ranges of LOGICAL waves, so the permutation decides which roles share a SIMD's issue slots (nibble w of MPCMP_WPERM5 = logical wave of hardware wave w).
#ifndef MPCMP_WPERM5
#define MPCMP_WPERM5 0xBA9876543210ull
#endif
template <int NT_>
__device__ __forceinline__ int q3_logical_tid() {
    if (NT_ == 768 && MPCMP_WPERM5 != 0xBA9876543210ull)
        return (int)((MPCMP_WPERM5 >> (4 * (threadIdx.x >> 6))) & 15ull) * 64 + (int)(threadIdx.x & 63);
    return (int)threadIdx.x;
}
#define QP3_PROLOGUE_L(LT_, NT_, FILL_CT_) \
    using D = Dim3<NSEG>; \
    using L = LT_; \
    constexpr int N = D::N, na = D::na, meq = D::meq, ma = D::ma, nJ = D::nJ, nI = D::nI, NT = NT_, GS = L::GS, JS = L::JS, SC = L::SC; \
    constexpr int n_tot = NARM * na + 1, mn_tot = NARM * (ma + na) + 1; \
    extern __shared__ __attribute__((aligned(16))) double lds[]; \
    const int tid = q3_logical_tid<NT_>(), wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63; \
    int slot, arm; \
    if (NARM == 1) { slot = blockIdx.x; arm = 0; } \
    else { const int bi = blockIdx.x; slot = (bi >> 4) * 8 + (bi & 7); arm = (bi >> 3) & 1; } \
    if (slot >= B) return; \
    const int b = ws.perm[slot]; \
    /* receding horizon: an arrived instance is not re-solved (both arm workgroups leave).  Not at N = 25: k_qp3<8, 2> sits at 256 VGPRs and this early exit \
       costs it 20 B more scratch and 8 % of its speed (144 -> 164 B, 64.6 -> 70.0 ms per launch: profiles/r05_v2_dual14_*); there a retired instance keeps its \
       QP workgroups busy on its last linearisation (harmless: k_init_m, k_step_m and k_advance skip it, nothing it writes is read) */ \
    if (NSEG < 8 && MPCMP_RETIRED(ws, b)) return; \
    const double ts = 1.0 / (2.0 * NSEG); \
    const double rho_in = cfg.rho, rho_eq = cfg.rho * cfg.rho_eq_scale, sigma = cfg.sigma, alpha = cfg.alpha; \
    const double *zg_ = ws.z + (size_t)b * n_tot + arm * na; \
    const double T = ws.z[(size_t)b * n_tot + NARM * na]; \
    const double tsT = ts * T; \
    const double *Gkg = ws.Gk + ((size_t)b * NARM + arm) * N * 176; \
    const double *lam_rows = ws.lam + (size_t)b * mn_tot + arm * ma; \
    const double *x0e = ws.x0 + (size_t)b * 14 * NARM, *xfe = ws.xf + (size_t)b * 14 * NARM; \
    auto arm_x = [&](const double *xe, int r) -> double { return r < 7 ? xe[7 * arm + r] : xe[7 * NARM + 7 * arm + (r - 7)]; }; \
    double *gkl = lds + L::oGk, *misc = lds + L::oMisc, *cD = lds + L::oCD, *redp = lds + L::oRedP; \
    unsigned long long *xown = NARM == 2 ? xch.buf + ((size_t)b * 2 + arm) * MPCMP_XCH_STRIDE : nullptr; \
    const unsigned long long *xpar = NARM == 2 ? xch.buf + ((size_t)b * 2 + (1 - arm)) * MPCMP_XCH_STRIDE : nullptr; \
    int dead = 0, status = 0; \
    for (int i = tid; i < N * 8 * GS; i += NT) gkl[i] = (i % GS < 22) ? Gkg[(i / GS) * 22 + i % GS] : 0.0; \
    auto coef_T = [&](int r) -> double { const int k = r / 14, rr = r % 14; return -ts * zg_[(rr < 7) ? 14 * k + 7 + rr : 14 * N + 7 * k + rr - 7]; }; \
    if (FILL_CT_) for (int r = tid; r < meq; r += NT) lds[L::oCT + r] = coef_T(r); \
    if (tid < 32) cD[tid] = tid < 16 ? c_D[tid] : 0.0; \
    if (tid < 32) misc[tid] = tid == L::M_mtsT ? -tsT : 0.0; \
    double *cfl = lds + L::oCfg; \
    if (tid < 14) { cfl[tid] = cfg.lbx[tid]; cfl[14 + tid] = cfg.ubx[tid]; } \
    else if (tid < 21) { cfl[28 + tid - 14] = cfg.lbu[tid - 14]; cfl[35 + tid - 14] = cfg.ubu[tid - 14]; } \
    else if (tid < 29) { cfl[42 + tid - 21] = cfg.lbg[tid - 21]; cfl[50 + tid - 21] = cfg.ubg[tid - 21]; } \
    { \
        int *pt = reinterpret_cast<int *>(lds + L::oPat); \
        if (tid < 49) pt[tid] = (int)pat->jc[tid]; \
        else if (tid < 77) pt[tid] = (int)pat->cjl[tid - 49]; \
        else if (tid < 105) pt[tid] = (int)pat->cjh[tid - 77]; \
    } \
    __syncthreads(); \
    const double *c_lbx = cfl, *c_ubx = cfl + 14, *c_lbu = cfl + 28, *c_ubu = cfl + 35, *c_lbg = cfl + 42, *c_ubg = cfl + 50; \
    /* one term rho_r A[r][a] A[r][b] of an entry of A^T rho A.  Both factors are read from LDS through ONE computed address each \
       (a select between an LDS load and a register value makes the compiler select between POINTERS, one of them into scratch, \
       and load through a generic pointer: 0.8 M cycles per factorisation went into those flat loads) */ \
    auto term_val = [&](uint32_t t) -> double { \
        const int r = t >> 16, a = (t >> 8) & 255, c = t & 255; \
        const bool dyn = r < meq; \
        const int i4 = 4 * ((r / 14) % 3), gro = L::oGk + (r - meq) * GS; \
        const int ia = dyn ? (a < 4 ? L::oCD + i4 + a : (a == 4 ? L::oMisc + L::M_mtsT : L::oCT + r)) : gro + a; \
        const int ic = dyn ? (c < 4 ? L::oCD + i4 + c : (c == 4 ? L::oMisc + L::M_mtsT : L::oCT + r)) : gro + c; \
        const int iq = L::oCfg + 42 + (dyn ? 0 : ((r - meq) & 7)); \
        const double va = lds[ia], vb = lds[ic], lbq = lds[iq], ubq = lds[iq + 8]; \
        const double rho = (dyn || ubq - lbq < 1e-4) ? rho_eq : rho_in; \
        return rho * va * vb; \
    }; \
    auto dst_of = [&](int e) -> double * { \
        if (e < D::eKUU) return lds + L::fKJJ + e; \
        if (e < D::eKJC) return lds + L::fKUU + (e - D::eKUU); \
        if (e < D::eKUX) return lds + L::oKJC + (e - D::eKJC); \
        if (e < D::eKuX) return lds + L::oKUX + (e - D::eKUX); \
        if (e < D::eKT) return lds + L::oKuX + (e - D::eKuX); \
        if (e < D::EA) return lds + L::oKT + (e - D::eKT); \
        return lds + L::fKJJ + (e - D::eS); \
    }; \
    /* Assembly of the entries of K_0 in the slots [e0, e1).  Term lists in ELL form over SLOTS (the entries of the range sorted by list \
       length, longest first: mpcmp.hip): ws.entry_ptr[s] = number of terms | entry << 8, ws.terms[t * EP + s] (0xFFFFFFFF beyond a \
       list).  A thread owns up to NE slots and walks their lists together: per step the NE term words are loaded by independent, \
       coalesced requests (one round trip to L2 per step; entry after entry it was ~90 dependent round trips of 2 - 4 k cycles each \
       while every workgroup of the launch reads the same table), and the lanes of a wave hold lists of equal length. */ \
    auto assemble = [&](int e0, int e1, auto ne_tag) { \
        constexpr int NE = decltype(ne_tag)::value; \
        constexpr int EP = (D::E + 63) / 64 * 64; \
        int cnt[NE], mymax = 0; \
        double acc[NE]; \
        _Pragma("unroll") for (int i = 0; i < NE; i++) { \
            const int e = e0 + tid + i * NT; \
            cnt[i] = e < e1 ? (ws.entry_ptr[e] & 255) : 0; \
            acc[i] = 0.0; \
        } \
        _Pragma("unroll") for (int i = 0; i < NE; i++) mymax = cnt[i] > mymax ? cnt[i] : mymax; \
        uint32_t wn[NE];                /* the words of step t + 1 are in flight while step t is evaluated (one L2 round trip per step otherwise) */ \
        _Pragma("unroll") for (int i = 0; i < NE; i++) { const int e = e0 + tid + i * NT; wn[i] = ws.terms[e < e1 ? e : e0]; } \
        for (int t = 0; t < mymax; t++) { \
            uint32_t w[NE]; \
            _Pragma("unroll") for (int i = 0; i < NE; i++) w[i] = wn[i]; \
            const int tn = t + 1 < mymax ? t + 1 : t; \
            _Pragma("unroll") for (int i = 0; i < NE; i++) { const int e = e0 + tid + i * NT; wn[i] = ws.terms[tn * EP + (e < e1 ? e : e0)]; } \
            _Pragma("unroll") for (int i = 0; i < NE; i++) if (t < cnt[i]) acc[i] += term_val(w[i]); \
        } \
        _Pragma("unroll") for (int i = 0; i < NE; i++) { const int e = e0 + tid + i * NT; if (e < e1) *dst_of(ws.entry_ptr[e] >> 8) = acc[i]; } \
    }; \
    auto var_h = [&](int v, double &ha, double &rb, double &lo, double &hi) { \
        ha = 0.0; \
        if (v < 14 * N) { \
            const int k = v / 14, c = v % 14; \
            if (k == 0) { lo = hi = arm_x(x0e, c); } \
            else if (k == N - 1) { const double t = arm_x(xfe, c); lo = t - cfg.eps_target; hi = t + cfg.eps_target; } \
            else { lo = c_lbx[c]; hi = c_ubx[c]; } \
            if (c >= 7 && k <= N - 2) ha = -ts * lam_rows[14 * k + (c - 7)]; \
        } else { \
            const int k = (v - 14 * N) / 7, c = (v - 14 * N) % 7; \
            lo = c_lbu[c]; hi = c_ubu[c]; \
            if (k <= N - 2) ha = -ts * lam_rows[14 * k + 7 + c]; \
        } \
        rb = (hi - lo < 1e-4) ? rho_eq : rho_in; \
    };

// Factorisation half of the QP: assemble K_0 of one arm, invert the interior blocks and the interface Schur complement, and
// leave the factor (one row of G_s per G lane, a quarter row of S^-1 per S lane, the sparse coupling blocks, the T column) in
// the factor workspace `fac` [B][NARM][Qp3::FAC].  A kernel of its own so that the ADMM loop kernel's register allocation is
// not entangled with the sweeps' (with both in one kernel the compiler kept the factor rows in scratch: 45 serialised scratch
// reloads per matrix-vector product).
template <int NSEG, int NARM, int LAY = 3>
__global__ __launch_bounds__(1024) void k_qp3f(mpcmp_config cfg, WS ws, const Qp3Pat *__restrict__ pat, Xch xch, int B, double *__restrict__ fac) {
    QP3_PROLOGUE(1024, true)
    using L4 = Qp4Fac<NSEG>;                         // LAY == 4: hand-over in the layout of k_qp4 (qp_kernel_v4.hpp)
    using L5 = Qp5Fac<NSEG>;                        // LAY == 5: hand-over in the layout of k_qp5 (qp_kernel_v5.hpp): + E_s, S^-1 in 4 x 13 blocks
    constexpr int FACSZ = LAY == 4 ? L4::FAC : (LAY == 5 ? L5::FAC : L::FAC);
#ifdef MPCMP_STAMPS
    unsigned long long fst_t = clock64();
#define FST(k) do { if (tid == 0 && arm == 0) { const unsigned long long n_ = clock64(); ws.dbg[(size_t)b * MPCMP_DBG_WORDS + 128 + (k)] = n_ - fst_t; fst_t = n_; } } while (0)
#else
#define FST(k) do { } while (0)
#endif
    FST(0);
    assemble(0, D::EA, std::integral_constant<int, (D::EA + 1023) / 1024>());
    {   // sum |ha| of this arm (Gershgorin row of T, polympc_redef.hpp:57-70); kappa_a = sum_r rho_r A[r][T]^2 (one term per row:
        // a reduction, not an entry of the term table: walked by one thread its 400+ terms took 0.7 M cycles)
        double s = 0.0, kq = 0.0;
        for (int v = tid; v < na; v += NT) { double ha, rb, lo, hi; var_h(v, ha, rb, lo, hi); s += fabs(ha); }
        for (int r = tid; r < ma; r += NT) {
            const double cf = r < meq ? lds[L::oCT + r] : gkl[(r - meq) * GS + 21];
            const int q = (r - meq) & 7;
            const double rho = (r < meq || c_ubg[q] - c_lbg[q] < 1e-4) ? rho_eq : rho_in;
            kq += rho * cf * cf;
        }
        double sv[2] = {s, kq};
        block_reduce<16, 2, false>(sv, redp, tid);        // (its barriers also publish the assembled entries)
        if (tid == 0) { misc[L::M_sumha] = sv[0]; lds[L::oKT + na] = sv[1]; }
    }
    // diagonal H + sigma I + rho_box of the interior and U blocks; Hessian arrow into the T column
    for (int v = tid; v < na; v += NT) {
        double ha, rb, lo, hi;
        var_h(v, ha, rb, lo, hi);
        const double d = (fabs(ha) + cfg.hess_reg) + sigma + rb;      // Gershgorin shift: a_ii = 0 -> |ha| + hess_reg
        const int ip = int3_of_ext(NSEG, v);
        if (ip < nJ) lds[L::fKJJ + (ip / 49) * D::JP + packed(ip % 49, ip % 49)] += d;
        else if (ip < nJ + 7) lds[L::fKUU + packed(ip - nJ, ip - nJ)] += d;
        lds[L::oKT + ip] += ha;
    }
    __syncthreads();

    FST(1);
    // ---------------- factorisation ----------------
    auto tri_decode = [](int e, int &i, int &j) {
        i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= e) i++;
        while (i * (i + 1) / 2 > e) i--;
        j = e - i * (i + 1) / 2;
    };
    // Symmetric sweep of `nblk` SPD nb x nb blocks, all pivots: A <- -(A^-1).  A thread owns one 4x4 tile of the lower block
    // triangle in registers; per step only the pivot column and the pivot reciprocal travel through LDS (double buffered):
    // one barrier per step (same scheme as k_qp2's sweep).
    auto sweep = [&](int nb, int nblk, int cst, auto &&ld, auto &&st_) {
        constexpr int CB = L::CB;
        double *rdv = lds + L::fRdv;
        const int nt4 = (nb + 3) >> 2, ntile = nt4 * (nt4 + 1) / 2;
        // one tile per thread (every sweep of this kernel has at most NT tiles: checked at the call sites)
        constexpr int NH = 1;
        bool live[NH], diag[NH];
        int blk[NH], Ib[NH], Jb[NH];
        double *cb0[NH];
        double v[NH][4][4];
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const int t = tid + h * NT;
            live[h] = t < ntile * nblk;
            blk[h] = 0; Ib[h] = 0; Jb[h] = 0;
            if (live[h]) { blk[h] = t / ntile; tri_decode(t % ntile, Ib[h], Jb[h]); }
            diag[h] = live[h] && Ib[h] == Jb[h];
            cb0[h] = lds + L::fCol + blk[h] * cst;
#pragma unroll
            for (int a = 0; a < 4; a++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = 4 * Ib[h] + a, j = 4 * Jb[h] + q;
                    v[h][a][q] = (live[h] && i < nb && j < nb) ? (i >= j ? ld(blk[h], i, j) : ld(blk[h], j, i)) : ((live[h] && i == j) ? 1.0 : 0.0);
                }
            }
        }
        if (tid == 0) rdv[15] = 0.0;                  // "a pivot was not positive" (a flag in LDS: `status` as a register live across the steps was spilled and re-read in every one of them)
        __syncthreads();                              // (every tile is in registers before the first column is published)
#pragma unroll
        for (int h = 0; h < NH; h++) {
            if (live[h] && Jb[h] == 0) {
#pragma unroll
                for (int a = 0; a < 4; a++) cb0[h][4 * Ib[h] + a] = v[h][a][0];
                if (Ib[h] == 0) { if (!(v[h][0][0] > 0.0)) rdv[15] = 1.0; rdv[blk[h]] = pivot_rcp(v[h][0][0]); }
            }
        }
        __syncthreads();
        for (int kb = 0; 4 * kb < nb; kb++) {          // (pivots nb .. 4 nt4 - 1 are the identity padding: nothing to sweep)
#pragma unroll
            for (int ka = 0; ka < 4; ka++) {
                const int k = 4 * kb + ka;
                if (k < nb) {
                    const int k1a = (ka + 1) & 3, k1b = kb + (ka == 3 ? 1 : 0);
#pragma unroll
                    for (int h = 0; h < NH; h++) {
                        if (live[h]) {
                            const double *cur = cb0[h] + (k & 1) * CB;
                            double *nxt = cb0[h] + ((k + 1) & 1) * CB;
                            const D2 ci0 = lds2(cur + 4 * Ib[h]), ci1 = lds2(cur + 4 * Ib[h] + 2);
                            const D2 cj0 = lds2(cur + 4 * Jb[h]), cj1 = lds2(cur + 4 * Jb[h] + 2);
                            const double rd = rdv[(k & 1) * 16 + blk[h]];
                            const double cI[4] = {ci0.x, ci0.y, ci1.x, ci1.y};
                            const double rJ[4] = {cj0.x * rd, cj0.y * rd, cj1.x * rd, cj1.y * rd};
#pragma unroll
                            for (int a = 0; a < 4; a++) {
#pragma unroll
                                for (int q = 0; q < 4; q++) v[h][a][q] = v[h][a][q] - cI[a] * rJ[q];
                            }
                            if (Ib[h] == kb) {
#pragma unroll
                                for (int q = 0; q < 4; q++) v[h][ka][q] = rJ[q];
                            }
                            if (Jb[h] == kb) {
#pragma unroll
                                for (int a = 0; a < 4; a++) v[h][a][ka] = cI[a] * rd;
                                if (Ib[h] == kb) v[h][ka][ka] = -rd;
                            }
                            if (k + 1 < nb) {
                                if (Jb[h] == k1b) {
#pragma unroll
                                    for (int a = 0; a < 4; a++) nxt[4 * Ib[h] + a] = (diag[h] && a < k1a) ? v[h][k1a][a] : v[h][a][k1a];
                                    if (diag[h]) {
                                        const double pv = v[h][k1a][k1a];
                                        if (!(pv > 0.0)) rdv[15] = 1.0;
                                        rdv[((k + 1) & 1) * 16 + blk[h]] = pivot_rcp(pv);
                                    }
                                } else if (Ib[h] == k1b) {
#pragma unroll
                                    for (int q = 0; q < 4; q++) nxt[4 * Jb[h] + q] = v[h][k1a][q];
                                }
                            }
                        }
                    }
                    __syncthreads();
                }
            }
        }
#pragma unroll
        for (int h = 0; h < NH; h++) {
            if (live[h]) {
#pragma unroll
                for (int a = 0; a < 4; a++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int i = 4 * Ib[h] + a, j = 4 * Jb[h] + q;
                        if (i < nb && j <= i) st_(blk[h], i, j, v[h][a][q]);
                    }
                }
            }
        }
        if (tid == 0 && rdv[15] != 0.0) status |= 2;
        __syncthreads();
    };
    // interior blocks K_JJ,s (49 x 49) and K_UU (7 x 7, padded with an identity to 49 x 49), all at once
    static_assert(91 * (NSEG + 1) <= NT && ((nI + 3) / 4) * ((nI + 3) / 4 + 1) / 2 <= NT, "one 4 x 4 tile per thread in both sweeps");
    sweep(49, NSEG + 1, 52,
          [&](int blk, int i, int j) -> double {
              if (blk < NSEG) return lds[L::fKJJ + blk * D::JP + packed(i, j)];
              return i < 7 ? lds[L::fKUU + packed(i, j)] : (i == j ? 1.0 : 0.0);
          },
          [&](int blk, int i, int j, double val) {
              if (blk < NSEG) lds[L::fKJJ + blk * D::JP + packed(i, j)] = val;
              else if (i < 7) lds[L::fKUU + packed(i, j)] = val;
          });
    FST(2);
    // ---------------- Schur complement S = K_II - sum_s K_CJ G_s K_JC - K_XU G_u K_UX ----------------
    // The segment products E_s = G_s K_JC,s (49 x 49 by 49 x 28) and K_CJ,s E_s (28 x 49 by 49 x 28) are the block GEMMs of the QP
    // and run on the matrix cores (v_mfma_f64_16x16x4_f64; lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15], and row
    // (l >> 4) + 4 r, column l & 15 of the result in register r).  Two waves per segment (sw and sw + 8), one 16-column tile of E_s
    // each.  Register r of row tile mt of E_s IS the B operand of k-step 4 mt + r of the second product (same lane map), and a
    // fragment of K_JC serves as B operand of the first product and as A operand (K_CJ = K_JC^T) of the second: no data movement.
    // G_s is read where the sweep left it; the products are taken BEFORE K_II is assembled over it, and subtracted afterwards.
    using V4 = __attribute__((ext_vector_type(4))) double;
    const int sw = wave & 7, hf = wave >> 3;
    double *fa = fac + ((size_t)b * NARM + arm) * FACSZ;
    if (LAY == 4) {
        // k_qp4 (384 lanes): lane Lq = 4 Q + part (Q = 24 seg + lp) keeps rows 2 lp, 2 lp + 1 (the row of its own parity first) x columns
        // 14 part .. + 13 of G_seg, entry e = 14 a + j at [e][384 lanes]; row 48 of every segment goes out whole ([NSEG][56], zero padded).
        for (int w = tid; w < 2 * 384; w += NT) {
            const int Lq = w % 384, a = w / 384, Q = Lq >> 2, part = Lq & 3, seg = Q / 24, lp = Q % 24;
            const double *Gs = lds + L::fKJJ + (seg < NSEG ? seg : 0) * D::JP;
            const int row = 2 * lp + (a ^ (part & 1));
#pragma unroll 1
            for (int j = 0; j < 14; j++) {
                const int col = 14 * part + j;
                const bool in = seg < NSEG && col < 49;
                const double g = -Gs[packed(row, in ? col : 0)];
                fa[L4::fG + (14 * a + j) * 384 + Lq] = in ? g : 0.0;
            }
        }
        for (int i = tid; i < NSEG * 56; i += NT) { const int sg = i / 56, c = i % 56; fa[L4::fG48 + i] = c < 49 ? -lds[L::fKJJ + sg * D::JP + packed(48, c)] : 0.0; }
        if (tid < 56) { const int r = tid >> 3, c = tid & 7; fa[L4::fGu + tid] = (r < 7 && c < 7) ? -lds[L::fKUU + packed(r, c)] : 0.0; }
    } else {
        // G_s leaves for the factor workspace in the block layout of g_blk ([segment][52 entries][64 lanes]: lane L keeps row L, entry
        // ((L & 3) ^ mcol) * 13 + c % 13 of the lane (L & ~3) + mcol holds G[L][c], mcol = c / 13; lanes 56..62 of the last segment: G_u).
        // Written entry by entry, 64 consecutive doubles per store (by row, a store touched 64 different cache lines: 90 k cycles);
        // each of the segment's two waves writes one half of the entries.
        const double *Gs = lds + L::fKJJ + (sw < NSEG ? sw : 0) * D::JP, *Gu = lds + L::fKUU;
        double *fg = fa + L::oFG + (size_t)(sw * 52) * 64 + lane;
        const int mcol = lane & 3, qb = lane & ~3;
#pragma unroll 1
        for (int e = 26 * hf; e < 26 * hf + 26; e++) {       // (a rolled loop: unrolled, its independent loads and index chains were all hoisted and spilled)
            const int eq = e / 13, row = qb + (eq ^ mcol), c = mcol * 13 + e % 13;
            const bool inG = sw < NSEG && row < 49 && c < 49, inU = sw == NSEG - 1 && row >= 56 && row < 63 && c < 7;
            const double g = -Gs[packed(inG ? row : 0, inG ? c : 0)], u = -Gu[packed(inU ? row - 56 : 0, inU ? c : 0)];
            fg[e * 64] = inG ? g : (inU ? u : 0.0);
        }
    }
    V4 sacc[2];
    sacc[0] = V4{0.0, 0.0, 0.0, 0.0}; sacc[1] = V4{0.0, 0.0, 0.0, 0.0};
    if (sw < NSEG) {
        const double *Gs = lds + L::fKJJ + sw * D::JP;
        const double *kjc = lds + L::oKJC + sw * 196, *kux = lds + L::oKUX + sw * 98;
        const int li = lane & 15, lk = lane >> 4;
        // K_JC fragment (ks, t) = K_JC[4 ks + lk][16 t + li] (rows >= 49, columns >= 28: zero), recomputed where it is used (a table of the
        // 26 fragments was spilled while it was built): loads with safe indices, then selects
        uint32_t jw[13];
#pragma unroll
        for (int ks = 0; ks < 13; ks++) { const int row = 4 * ks + lk; jw[ks] = pat->jc[row < 49 ? row : 0]; }
        auto frag = [&](int ks, int t, int lkk, int lii) -> double {
            const int row = 4 * ks + lkk, rs = row < 49 ? row : 0, col = 16 * t + lii;
            double kq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) kq[q] = kjc[rs * 4 + q];
            const double dv = kux[(rs < 7 ? rs : 0) * 14 + (lii < 14 ? lii : 0)];
            double val = 0.0;
#pragma unroll
            for (int q = 0; q < 4; q++) if ((int)((jw[ks] >> (8 * q)) & 255u) == col) val = kq[q];
            if (t == 0 && rs < 7 && lii < 14) val = dv;
            return (row < 49 && col < 28) ? val : 0.0;
        };
        auto products = [&](auto NTc) {
            constexpr int nt = decltype(NTc)::value;
            V4 e[4];
#pragma unroll
            for (int mt = 0; mt < 4; mt++) e[mt] = V4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 13; ks++) {
                int lkk = lk, lii = li;
                asm volatile("" : "+v"(lkk), "+v"(lii) :: "memory");     // (opaque copies: the operand addresses are not computed ahead of their k-step, where they spilled)
                const int k = 4 * ks + lkk;
                double av[4];
#pragma unroll
                for (int mt = 0; mt < 4; mt++) {
                    const int i = 16 * mt + lii;
                    const double g = -Gs[packed(i < 49 ? i : 0, k < 49 ? k : 0)];
                    av[mt] = (i < 49 && k < 49) ? g : 0.0;
                }
                const double bf = frag(ks, nt, lkk, lii);
#pragma unroll
                for (int mt = 0; mt < 4; mt++) e[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[mt], bf, e[mt], 0, 0, 0);
            }
            if (LAY == 5) {
                // E_s leaves for the loop kernel straight from the accumulators: register r of row tile mt holds E[16 mt + (lane >> 4) + 4 r][16 nt + (lane & 15)]
                // (rows 49..51 of the last row group are exact zeros: their A operands were)
#pragma unroll
                for (int mt = 0; mt < 4; mt++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int R = 16 * mt + lk + 4 * r, Cc = 16 * nt + li;
                        if (R < 52 && Cc < 28) {
                            const int g = R >> 2, h = Cc >= 14 ? 1 : 0, a = (R & 3) ^ (2 * h), j = Cc - 14 * h;
                            fa[L5::oFE + (14 * a + j) * L5::ELS + 26 * sw + 2 * g + h] = e[mt][r];
                        }
                    }
                }
            }
#pragma unroll
            for (int ks = 0; ks < 13; ks++) {
                int lkk = lk, lii = li;
                asm volatile("" : "+v"(lkk), "+v"(lii) :: "memory");
#pragma unroll
                for (int mt2 = nt; mt2 < 2; mt2++)
                    sacc[mt2] = __builtin_amdgcn_mfma_f64_16x16x4f64(frag(ks, mt2, lkk, lii), e[ks >> 2][ks & 3], sacc[mt2], 0, 0, 0);
            }
        };
        if (hf == 0) products(std::integral_constant<int, 0>());
        else products(std::integral_constant<int, 1>());
    }
    if (tid < 28) lds[L::oGu + tid] = lds[L::fKUU + tid];           // -(K_UU^-1), for the Schur complement
    __syncthreads();
    // interface block K_II + its diagonal
    assemble(D::eS, D::E, std::integral_constant<int, (D::SP + 1023) / 1024>());
    __syncthreads();
    for (int v = tid; v < na; v += NT) {
        const int ip = int3_of_ext(NSEG, v);
        if (ip >= nJ + 7) {
            double ha, rb, lo, hi;
            var_h(v, ha, rb, lo, hi);
            const int ia = ip - nJ - 7;
            lds[L::fKJJ + packed(ia, ia)] += (fabs(ha) + cfg.hess_reg) + sigma + rb;
        }
    }
    __syncthreads();
    FST(3);
    // the segment products leave the accumulators: tile (mt2, nt = hf) of the segment's 28 x 28 block, lower triangle; even and odd
    // segments in turn (neighbours share the diagonal block of their common interface node)
    double *S = lds + L::fKJJ;
    for (int ph = 0; ph < 2; ph++) {
        if (sw < NSEG && (sw & 1) == ph) {
#pragma unroll
            for (int mt2 = 0; mt2 < 2; mt2++) {
                if (mt2 >= hf) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int R = 16 * mt2 + (lane >> 4) + 4 * r, C = 16 * hf + (lane & 15);
                        if (R < 28 && C < 28 && R >= C) S[packed(14 * sw + R, 14 * sw + C)] -= sacc[mt2][r];
                    }
                }
            }
        }
        __syncthreads();
    }
    if (tid < 105) {
        int a, c;
        tri_decode(tid, a, c);
        const double *kx = lds + L::oKuX, *gu = lds + L::oGu;
        double acc = 0.0;
        for (int r = 0; r < 7; r++)
            for (int q = 0; q < 7; q++) acc += kx[r * 14 + a] * gu[packed(r, q)] * kx[q * 14 + c];
        S[packed(14 * NSEG + a, 14 * NSEG + c)] += acc;             // gu holds -(K_UU^-1)
    }
    __syncthreads();
    FST(4);
    sweep(nI, 1, L::CB,
          [&](int, int i, int j) -> double { return S[packed(i, j)]; },
          [&](int, int i, int j, double val) { S[packed(i, j)] = val; });        // S <- -(S^-1)
    FST(5);
    // ---------------- derived copies of the coupling blocks in the access order of the loop kernel ----------------
    if (LAY == 4) {
        // column form [NSEG + 1][7 d][KX]: columns 0..27: entry of C-column c in row c % 14 + 7 (d - 1) of the segment (sparse part);
        // columns 28..41: the dense block u_3s x x_3s, row d (block NSEG: u_{N-1} x x_{N-1}); columns 42, 43: zero
        for (int i = tid; i < (NSEG + 1) * 7 * L4::KX; i += NT) {
            const int sg = i / (7 * L4::KX), d = (i % (7 * L4::KX)) / L4::KX, col = i % L4::KX;
            double val = 0.0;
            if (col < 28) {
                const int r = col % 14 + 7 * (d - 1);
                if (sg < NSEG && r >= 0 && r < 49) {
                    const uint32_t w = pat->jc[r];
#pragma unroll
                    for (int q = 0; q < 4; q++) if ((int)((w >> (8 * q)) & 255u) == col) val = lds[L::oKJC + sg * 196 + r * 4 + q];
                }
            } else if (col < 42) {
                val = sg < NSEG ? lds[L::oKUX + sg * 98 + d * 14 + (col - 28)] : lds[L::oKuX + d * 14 + (col - 28)];
            }
            fa[L4::fKX + i] = val;
        }
    } else {
    for (int i = tid; i < NSEG * 224; i += NT) {
        const int sg = i / 224, d = (i % 224) / 28, c = i % 28, r = c % 14 + 7 * (d - 1);
        double val = 0.0;
        if (d < 7 && r >= 0 && r < 49) {
            const uint32_t w = pat->jc[r];
#pragma unroll
            for (int q = 0; q < 4; q++) if ((int)((w >> (8 * q)) & 255u) == c) val = lds[L::oKJC + sg * 196 + r * 4 + q];
        }
        fa[L::oFD + i] = val;
    }
    for (int i = tid; i < (NSEG + 1) * 112; i += NT) {
        const int sg = i / 112;
        const double *src = sg < NSEG ? lds + L::oKUX + sg * 98 : lds + L::oKuX;
        { const int c = (i % 112) >> 3, r = i & 7; fa[L::oFD + (L::oKUXT - L::oKCJ) + i] = r < 7 ? src[r * 14 + c] : 0.0; }
        { const int r = (i % 112) >> 4, c = i & 15; fa[L::oFD + (L::oKUXP - L::oKCJ) + i] = c < 14 ? src[r * 14 + c] : 0.0; }
    }
    if (tid < 16) fa[L::oFD + (L::oZR - L::oKCJ) + tid] = 0.0;
    }
    if (tid < 4) lds[L::oKJC + NSEG * 196 + tid] = 0.0;
    __syncthreads();
    // ---------------- hand the factor to the loop kernel (once per QP: ~45k doubles per arm) ----------------
    {
        const int any = __syncthreads_or(status);
        if (tid == 0 && any) atomicOr(&ws.status[b], any);
    }
    for (int i = tid; i < L::KJN; i += NT) fa[i] = lds[L::oKJC + i];                // sparse K_JC + zero row
    for (int i = tid; i < L::NAP; i += NT) fa[L::oFT + i] = lds[L::oKT + i];        // T column, kappa
    if (tid == 0) fa[L::oFH] = misc[L::M_sumha];
    if (LAY == 4) {
        for (int i = tid; i < nI * L4::SRS; i += NT) { const int r = i / L4::SRS, c = i % L4::SRS; fa[L4::fS + i] = c < nI ? -S[packed(r, c)] : 0.0; }
    } else if (LAY == 5) {
        for (int i = tid; i < 52 * L5::NSL; i += NT) {
            const int e = i / L5::NSL, ls = i % L5::NSL, c8 = ls & 7;
            const int row = 4 * (ls >> 3) + ((c8 & 3) ^ (e / 13)), col = 13 * c8 + e % 13;
            fa[L5::oFS + i] = (row < nI && col < nI) ? -S[packed(row, col)] : 0.0;
        }
        // E_u = G_u K_UX (7 x 14) as "segment" NSEG of the E blocks (columns 14 .. 27: zero); zeros in the lanes nobody owns
        constexpr int NU = L5::ELS - 26 * NSEG;
        for (int i = tid; i < 56 * NU; i += NT) {
            const int e = i / NU, lu = i % NU, g = lu >> 1, a = e / 14, c = e % 14, r = 4 * g + a;
            double val = 0.0;
            if (lu < 4 && (lu & 1) == 0 && r < 7) {
                for (int q = 0; q < 7; q++) val -= lds[L::oGu + packed(r, q)] * lds[L::oKuX + q * 14 + c];      // (oGu holds -(K_UU^-1))
            }
            fa[L5::oFE + e * L5::ELS + 26 * NSEG + lu] = val;
        }
    } else if (wave >= 8) {
        const int si = tid - 512, row0 = 4 * (si >> 4), mpos = si & 3, col0 = SC * (si & 15);
        for (int e = 0; e < 4 * SC; e++) {
            const int row = row0 + (mpos ^ (e / SC)), col = col0 + e % SC;
            fa[L::oFS + e * 512 + si] = (row < nI && col < nI) ? -S[packed(row, col)] : 0.0;
        }
    }
    __syncthreads();
    FST(6);
}

// ADMM half of the QP (see the header comment): 512 threads = 8 waves, two per SIMD, so every lane may hold 256 registers — its
// 4 x 13 block of G_s (104), its 4 x SC block of S^-1 (56..64) and its ADMM state stay in VGPRs for the whole loop.  (With 1024
// threads and 128 registers the compiler kept part of the factor in scratch; one scratch reload costs ~500 cycles and they are
// serialised, so the seven reloads of a 26-term dot product made it take 3,500 cycles.)  Every wave takes part in every phase:
//     A   rhs = sigma x - q + rho_b z_b - y_b + A^T w                      (lane = variable)
//     P1  t = G b_J, part = K_CJ t                                         (wave = segment, quad = four rows of G_s)
//     P3  r_I = b_I - part, y_I = S^-1 r_I                                 (sixteen lanes per four rows of S^-1)
//     P4  x_J = G (b_J - K_JC y_I), x~ = y - w x~_T                        (wave = segment; interface rows by the S lanes)
//     E   z~ = A x~, relaxation, projection, dual update                   (lane = row / variable)
// The loop is bound by instruction issue (a wave64 VALU or LDS instruction occupies its unit for >= 4 clocks, and there are only
// two waves per SIMD to overlap anything), so the data layout is chosen to make every hot-loop access "lane base + immediate":
// x~, the border vector w and w = rho z - y are kept in NODE order ([x_k | u_k | x~_T], [dynamics rows | path rows], stride 22 =
// the row stride of the path Jacobians), and what a lane needs to know about its variable / row is packed in one register.
template <int NSEG, int NARM>
__global__ __launch_bounds__(512) void k_qp3(mpcmp_config cfg, WS ws, const Qp3Pat *__restrict__ pat, Xch xch, int B, const double *__restrict__ fac) {
    QP3_PROLOGUE(512, false)
    constexpr int NS = L::NS, NX = L::NX;
    constexpr bool SMALL = NSEG >= 8;                // N = 25: smaller batches of LDS reads in flight (register budget)
#ifdef MPCMP_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_busy[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = clock64(), st_m = 0;
#endif
    // ---------------- the factor, as the factorisation kernel left it ----------------
    const double *fa = fac + ((size_t)b * NARM + arm) * L::FAC;
    for (int i = tid; i < L::KJN + L::NAP; i += NT) lds[L::lKJC + i] = fa[i];
    for (int i = tid; i < L::DER; i += NT) lds[L::oKCJ + i] = fa[L::oFD + i];
    if (tid == 0) misc[L::M_sumha] = fa[L::oFH];
    // ---------------- loop set-up ----------------
    int *rpos = reinterpret_cast<int *>(lds + L::oRpos), *xdgt = reinterpret_cast<int *>(lds + L::oXdG);
    auto rhs_slot = [&](int ip) -> int {
        return ip < nJ ? L::oRhsJ + JS * (ip / 49) + ip % 49 : (ip < nJ + 7 ? L::oRhsU + (ip - nJ) : L::oRhsI + (ip - nJ - 7));
    };
    auto node_slot = [&](int v) -> int {            // external arm variable -> slot in the node-ordered vectors
        return v < 14 * N ? NS * (v / 14) + v % 14 : NS * ((v - 14 * N) / 7) + 14 + (v - 14 * N) % 7;
    };
    double *xt = lds + L::oXt,
           *wg = lds + L::oWg, *wvv = lds + L::oWv, *redB = lds + L::oRedB, *redT = lds + L::oRedT;
    {
        // where the G lanes' entries of x~ go
        const int wv = tid >> 6, ln = tid & 63;
        int xd = NX;
        if (wv < NSEG && ln < 49) xd = node_slot(ws.ext_of_int[49 * wv + ln]);
        if (wv == NSEG - 1 && ln >= 56 && ln < 63) xd = node_slot(ws.ext_of_int[nJ + (ln - 56)]);
        xdgt[tid] = xd;
        // loop-resident constants and the rhs of K_0 w = k
        for (int i = tid; i < L::oLCT - L::oRpos; i += NT) lds[L::oRpos + i] = 0.0;
        if (L::STL) for (int i = tid; i < 4 * 512; i += NT) lds[L::oSt + i] = 0.0;              // vectors, pads, partial sums, second-pass state
        for (int i = tid; i < L::NXP; i += NT) lds[L::oWv + i] = 0.0;
        __syncthreads();
        for (int v = tid; v < na; v += NT) {
            double ha, rb, lo, hi;
            var_h(v, ha, rb, lo, hi);
            const double zv = zg_[v];
            lds[L::oLb + v] = lo - zv; lds[L::oUb + v] = hi - zv; if (L::HAL) lds[L::oHa + v] = ha;
            rpos[v] = rhs_slot(int3_of_ext(NSEG, v));
        }
        for (int r = tid; r < ma; r += NT) {
            double lg, ug, rr, cf;
            if (r < meq) { lg = ug = -ws.ceq[((size_t)b * NARM + arm) * meq + r]; rr = rho_eq; cf = coef_T(r); }
            else {
                const int q = (r - meq) & 7;
                const double gv = ws.g[((size_t)b * NARM + arm) * 8 * N + (r - meq)];
                lg = c_lbg[q] - gv; ug = c_ubg[q] - gv;
                rr = (ug - lg < 1e-4) ? rho_eq : rho_in;
                cf = gkl[(r - meq) * GS + 21];
            }
            lds[L::oLg + r] = lg; lds[L::oUg + r] = ug; lds[L::oCf + r] = cf;
        }
        if (tid == 0) {
            misc[L::M_lbT] = cfg.lbT - T; misc[L::M_ubT] = cfg.ubT - T;
            misc[L::M_rbT] = (cfg.ubT - cfg.lbT < 1e-4) ? rho_eq : rho_in;
        }
        __syncthreads();
        for (int ip = tid; ip < na; ip += NT) lds[rhs_slot(ip)] = lds[L::lKT + ip];
        __syncthreads();
    }
    // after the solve of K_0 w = k: w and delta of the T border (one exchange between the arm workgroups)
    auto finish_border = [&]() {
        double sacc = 0.0;
        for (int v = tid; v < na; v += NT) sacc += lds[L::lKT + int3_of_ext(NSEG, v)] * xt[node_slot(v)];
        double sv[1] = {sacc};
        block_reduce<8, 1, false>(sv, redp, tid);
        for (int v = tid; v < na; v += NT) wvv[node_slot(v)] = xt[node_slot(v)];
        if (tid == 0) {
            double kap[2] = {0.0, 0.0}, sh[2] = {0.0, 0.0}, dl[2] = {0.0, 0.0};
            kap[arm] = lds[L::lKT + na]; sh[arm] = misc[L::M_sumha]; dl[arm] = sv[0];
            if (NARM == 2) {
                xch_post(xown + 0, kap[arm]); xch_post(xown + 1, sh[arm]); xch_post(xown + 2, dl[arm]);
                kap[1 - arm] = xch_poll(xpar + 0, dead); sh[1 - arm] = xch_poll(xpar + 1, dead); dl[1 - arm] = xch_poll(xpar + 2, dead);
            }
            const double hdT = (sh[0] + sh[1]) + cfg.hess_reg;
            misc[L::M_hdT] = hdT;
            misc[L::M_delta] = ((kap[0] + kap[1]) + (hdT + sigma + misc[L::M_rbT])) - (dl[0] + dl[1]);
            misc[L::M_baseT] = -1.0;                              // sigma x_T - q_T + rho_T z_T - y_T with x = z = y = 0, q_T = 1 (cost = T)
            if (!(misc[L::M_delta] > 0.0)) atomicOr(&ws.status[b], 2);
        }
        __syncthreads();
    };
    // x~_T of the bordered solve (every wave for itself: LDS reads, and for two arms one poll of the partner's share)
    auto border_xT = [&](int it, int ln) -> double {
        double s0 = misc[L::M_s0], s1 = NARM == 2 ? misc[L::M_s1] : 0.0;
        if (NARM == 2) {
            double sp = 0.0;
            if (ln == 0) sp = xch_poll(xpar + 8 + it, dead);
            sp = read_lane(sp, 0);
            dead = __builtin_amdgcn_readfirstlane(dead);
            if (arm == 0) s1 = sp; else s0 = sp;
        }
        return (misc[L::M_baseT] + (s0 + s1)) / misc[L::M_delta];
    };
    // termination test, common tail: combine the arms, add the row / column of T, decide
    const unsigned chk_base = 8 + cfg.qp_iters + 1;
    auto check_tail = [&](double (&sums)[2], double (&mx)[6], int nchk) -> int {
        block_reduce_dpp<8, 6, true>(mx, redp, tid);
        double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0};
        s1[arm] = sums[0]; s2[arm] = sums[1];
        if (NARM == 2) {
            if (tid == 0) {
                unsigned long long *po = xown + chk_base + 8 * nchk;
                const unsigned long long *pp = xpar + chk_base + 8 * nchk;
                xch_post(po + 0, sums[0]); xch_post(po + 1, sums[1]);
#pragma unroll
                for (int q = 0; q < 6; q++) xch_post(po + 2 + q, mx[q]);
#pragma unroll
                for (int q = 0; q < 8; q++) misc[L::M_c0 + q] = xch_poll(pp + q, dead);
            }
            __syncthreads();
            s1[1 - arm] = misc[L::M_c0]; s2[1 - arm] = misc[L::M_c0 + 1];
#pragma unroll
            for (int q = 0; q < 6; q++) mx[q] = fmax(mx[q], misc[L::M_c0 + 2 + q]);
            __syncthreads();
        }
        const double xTv = misc[L::M_xT], zT = misc[L::M_zbT], yT = misc[L::M_ybT];
        const double hxT = misc[L::M_hdT] * xTv + (s2[0] + s2[1]), atyT = (s1[0] + s1[1]) + yT;
        mx[0] = fmax(mx[0], fabs(xTv - zT)); mx[1] = fmax(mx[1], fabs(xTv)); mx[2] = fmax(mx[2], fabs(zT));
        mx[3] = fmax(mx[3], fabs(hxT + atyT + 1.0)); mx[4] = fmax(mx[4], fabs(hxT)); mx[5] = fmax(mx[5], fabs(atyT));
        const double ep = cfg.eps_abs + cfg.eps_rel * fmax(mx[1], mx[2]);
        const double ed = cfg.eps_abs + cfg.eps_rel * fmax(fmax(mx[4], mx[5]), 1.0);      // ||q||_inf = 1
        return (mx[0] <= ep && mx[3] <= ed) ? 1 : 0;
    };
    // Per-lane constants of the solve: LDS addresses (in doubles).  Every access of the solve is such a base plus a compile-time
    // offset (the sparse K_JC is stored with canonical slots and a column-form copy for exactly this), so it carries no index
    // arithmetic, no pattern words and no selects; lanes without a job point at zero rows / pad slots.  The bases are cheap
    // functions of the lane index and are re-derived from an OPAQUE copy of it in every iteration (hoisted out of the loop they
    // would occupy twenty registers that the factor blocks need); the three that are not cheap are packed in two registers.
    const int *patw = reinterpret_cast<const int *>(lds + L::oPat);
    int pk_x, pk_y;                                  // x~ slots of the interior row (low) and of the interface row (high); y_C base column, rhs slot
    {
        const int ln = lane;
        int srow = 4 * (tid >> 4) + (tid & 3);
        if ((tid & 15) >= 4 || srow >= nI) srow = -1;                        // (not an output lane of S^-1)
        const int xds = srow >= 0 ? NS * 3 * (srow / 14) + srow % 14 : NX;    // interface rows x_0, x_3, ...
        pk_x = xdgt[tid] | (xds << 16);
        pk_y = (ln < 49 ? (patw[ln] & 255) : 7) | ((tid < na ? rpos[tid] : 0) << 16);
    }
    struct LaneC1 {
        int tslot, op1;               // tJ slot of the own row; operand block of G b_J
        int kcj, tcc, partd;          // column form of K_JC, tJ + c % 14, destination in PA / PB
        int p1k, p1t, p1d;            // dense blocks, two lanes per column (half a column of K_XU each): coefficients, operand, destination
    };
    struct LaneC3 { int rop, ysl; };  // operand of the S^-1 block, y_I slot
    struct LaneC4 {
        int xds, xdg;                 // destinations of the interface row and of the interior row in x~
        int tslot, op4;               // tJ slot of the own row; operand block of the second G product
        int bjr, kjr, ycb;            // own rhs entry, row of the sparse K_JC, y_C + base column
        int p4k, p4y, p4d, p4r;       // dense blocks, four lanes per row (a quarter row of K_UX each); where the own row finds its dense part
    };
    // (each set is derived right in front of its phase from a fresh opaque copy of the lane index: short live ranges)
    auto calc_c1 = [&](int t) -> LaneC1 {
        asm volatile("" : "+v"(t));
        LaneC1 c;
        const int ln = t & 63, wv = wave, wg_ = wv < NSEG ? wv : 0;          // (waves without a segment: valid addresses, results unused)
        const bool g_row = ln < 49, gu_quad = wv == NSEG - 1 && ln >= 56, gu_row = gu_quad && ln < 63;
        const int tJ = L::oTJ + wv * L::TS;
        c.tslot = g_row ? tJ + ln : (gu_row ? L::oTU + ln - 56 : tJ + 56);
        c.op1 = (gu_quad ? L::oRhsU : L::oRhsJ + JS * wg_) + 13 * (ln & 3);
        const int cl = ln < 28 ? ln : 27;
        c.kcj = L::oKCJ + wg_ * 224 + cl;
        c.tcc = tJ + (cl < 14 ? cl : cl - 14);
        c.partd = ln < 14 ? L::oPA + 14 * wg_ + ln : (ln < 28 ? L::oPB + 14 * wg_ + ln : L::oPD);      // (others: pad slot)
        const int c2 = ln >> 1, j = ln & 1;
        const bool lastw = wv == NSEG - 1, useg = c2 < 14, uU = lastw && c2 >= 14 && c2 < 28;
        c.p1k = useg ? L::oKUXT + (wg_ * 14 + c2) * 8 + 4 * j : (uU ? L::oKUXT + (NSEG * 14 + c2 - 14) * 8 + 4 * j : L::oZR);
        c.p1t = (uU ? L::oTU : tJ) + 4 * j;
        c.p1d = j ? L::oPD : (useg ? L::oDP + wg_ * 14 + c2 : (uU ? L::oPA + NSEG * 14 + c2 - 14 : L::oPD));
        return c;
    };
    auto calc_c3 = [&](int t) -> LaneC3 {
        asm volatile("" : "+v"(t));
        LaneC3 c;
        const int srow = 4 * (t >> 4) + (t & 3);
        c.rop = L::oRIw + SC * (t & 15);
        c.ysl = L::oYI + (((t & 15) < 4 && srow < nI) ? srow : nI + 1);      // (lanes without an output row: pad slot)
        return c;
    };
    auto calc_c4 = [&](int t, int px, int py) -> LaneC4 {
        asm volatile("" : "+v"(t), "+v"(px), "+v"(py));
        LaneC4 c;
        const int ln = t & 63, wv = wave, wg_ = wv < NSEG ? wv : 0;
        const bool g_row = ln < 49, gu_quad = wv == NSEG - 1 && ln >= 56, gu_row = gu_quad && ln < 63;
        const int lr = g_row ? ln : (gu_row ? ln - 56 : 0), tJ = L::oTJ + wv * L::TS;
        c.xds = L::oXt + (int)((unsigned)px >> 16);
        c.xdg = L::oXt + (px & 0xFFFF);
        c.tslot = g_row ? tJ + ln : (gu_row ? L::oTU + ln - 56 : tJ + 56);
        c.op4 = (gu_quad ? L::oTU : tJ) + 13 * (ln & 3);
        c.bjr = (gu_quad ? L::oRhsU : L::oRhsJ + JS * wg_) + lr;
        c.kjr = L::lKJC + (g_row ? wg_ * 196 + ln * 4 : NSEG * 196);         // (others: the zero row)
        c.ycb = L::oYI + 14 * wg_ + (py & 0xFFFF);
        const int r4 = ln >> 2, j = ln & 3;
        const bool lastw = wv == NSEG - 1, useg = r4 < 7, uU = lastw && r4 >= 7 && r4 < 14;
        c.p4k = useg ? L::oKUXP + (wg_ * 7 + r4) * 16 + 4 * j : (uU ? L::oKUXP + (NSEG * 7 + r4 - 7) * 16 + 4 * j : L::oZR);
        c.p4y = L::oYI + 14 * (uU ? NSEG : wg_) + 4 * j;
        c.p4d = L::oDW + wv * 16 + ((j == 0 && (useg || uU)) ? r4 : 15);
        c.p4r = ln < 7 ? L::oDW + wv * 16 + ln : (gu_row ? L::oDW + wv * 16 + 7 + lr : L::oZR);
        return c;
    };
    // N = 19: the same constants from a table in LDS (one 2-byte read each instead of ~8 vector instructions: the loop is bound
    // by instruction issue, and the LDS has room for 21 KB more); N = 25 re-derives them (no LDS left).
    unsigned short *lct = reinterpret_cast<unsigned short *>(lds + L::oLCT);
    if (L::LCT) {
        const LaneC1 a = calc_c1(tid); const LaneC3 b3 = calc_c3(tid); const LaneC4 d = calc_c4(tid, pk_x, pk_y);
        const int f[L::NLC] = {a.tslot, a.op1, a.kcj, a.tcc, a.partd, a.p1k, a.p1t, a.p1d, b3.rop, b3.ysl,
                               d.xds, d.xdg, d.tslot, d.op4, d.bjr, d.kjr, d.ycb, d.p4k, d.p4y, d.p4d, d.p4r};
#pragma unroll
        for (int q = 0; q < L::NLC; q++) lct[q * 512 + tid] = (unsigned short)f[q];
    }
    auto lct_get = [&](int t, int q) -> int { return (int)*(const volatile __attribute__((address_space(3))) unsigned short *)(lct + q * 512 + t); };
    auto lane_c1 = [&](int t) -> LaneC1 {
        if (!L::LCT) return calc_c1(t);
        asm volatile("" : "+v"(t));
        LaneC1 c;
        c.tslot = lct_get(t, 0); c.op1 = lct_get(t, 1); c.kcj = lct_get(t, 2); c.tcc = lct_get(t, 3); c.partd = lct_get(t, 4);
        c.p1k = lct_get(t, 5); c.p1t = lct_get(t, 6); c.p1d = lct_get(t, 7);
        return c;
    };
    auto lane_c3 = [&](int t) -> LaneC3 {
        if (!L::LCT) return calc_c3(t);
        asm volatile("" : "+v"(t));
        LaneC3 c;
        c.rop = lct_get(t, 8); c.ysl = lct_get(t, 9);
        return c;
    };
    auto lane_c4 = [&](int t, int px, int py) -> LaneC4 {
        if (!L::LCT) return calc_c4(t, px, py);
        asm volatile("" : "+v"(t));
        LaneC4 c;
        c.xds = lct_get(t, 10); c.xdg = lct_get(t, 11); c.tslot = lct_get(t, 12); c.op4 = lct_get(t, 13); c.bjr = lct_get(t, 14);
        c.kjr = lct_get(t, 15); c.ycb = lct_get(t, 16); c.p4k = lct_get(t, 17); c.p4y = lct_get(t, 18); c.p4d = lct_get(t, 19); c.p4r = lct_get(t, 20);
        return c;
    };
    // this lane's blocks of G_s and of S^-1 (k_qp3f left them in the factor workspace).  They are loaded again at the top
    // of every termination-test period, through an opaque pointer: a value defined right in front of the hot loop and dead
    // after it is kept in VGPRs by the register allocator.
    double mm[52], sm[4 * SC];
    auto load_rows = [&]() {
        const double *fo_ = fa + L::oFG + (size_t)(wave * 52) * 64 + lane, *so_ = fa + L::oFS + tid;
        asm volatile("" : "+v"(fo_), "+v"(so_));
        // (the opaque copies are generic pointers to the compiler: name the address space again, or the 80 loads become flat loads)
        typedef const __attribute__((address_space(1))) double *gptr_t;
        gptr_t fo = (gptr_t)fo_, so = (gptr_t)so_;
#pragma unroll
        for (int j = 0; j < 52; j++) mm[j] = fo[j * 64];
#pragma unroll
        for (int j = 0; j < 4 * SC; j++) sm[j] = so[j * 512];
    };
    // One solve with K_0 (three barriers), straight-line code for every lane.
    // (c1: the lane constants of P1, fetched by the caller in front of the barrier that opens P1; those of P3 and P4 are fetched
    // in front of their barriers as well: a table read is then not one more LDS round trip at the head of the phase)
    auto solve = [&](const bool use_xT, const int it, const int tl, const int px, const int py, const LaneC1 &c1) {
        const int wv = wave;
        // ---- P1: t = G b_J, part = K_CJ t; this arm's share of the T solve, s_a = (T column of A^T w) - w^T rhs ----
        if (wv < NSEG) {
            const LaneC1 &c = c1;
            lds[c.tslot] = g_blk<SMALL>(mm, lds + c.op1);
            wave_sync();
            const double *kc = lds + c.kcj, *tc = lds + c.tcc;
            double kq[7], tv[7], dk[4], dt[4];
#pragma unroll
            for (int d = 0; d < 7; d++) { kq[d] = ldv(kc + 28 * d); tv[d] = ldv(tc + 7 * (d - 1)); }      // rows c % 14 + 7 (d - 1) of the segment
#pragma unroll
            for (int d = 0; d < 4; d++) { dk[d] = ldv(lds + c.p1k + d); dt[d] = ldv(lds + c.p1t + d); }
            const double acc = ((kq[0] * tv[0] + kq[1] * tv[1]) + (kq[2] * tv[2] + kq[3] * tv[3])) + ((kq[4] * tv[4] + kq[5] * tv[5]) + kq[6] * tv[6]);
            lds[c.partd] = acc;
            // dense blocks (K_XU t of the columns x_3s; for the last segment also the U block): half a column per lane
            const double ad = (dk[0] * dt[0] + dk[1] * dt[1]) + (dk[2] * dt[2] + dk[3] * dt[3]);
            lds[c.p1d] = ad + dpp_mov<0xB1>(ad);                                   // (odd lanes, lanes without a column: pad slot)
        }
        if (L::P8) {
            if (use_xT && wv == 7) {                                           // (wave 7 holds no segment at N = 19)
                const double sa = wave_sum(ldv(redT + lane) - ldv(redB + lane));
                if (lane == 0) {
                    misc[L::M_s0 + arm] = sa;
                    if (NARM == 2) xch_post(xown + 8 + it, sa);
                    else misc[L::M_xtT] = (misc[L::M_baseT] + sa) / misc[L::M_delta];      // one arm: x~_T right here (read in P4)
                }
            }
        } else if (use_xT && tid == 511) {
            double ssum = 0.0, bsum = 0.0;
#pragma unroll
            for (int w8 = 0; w8 < 8; w8++) { ssum += redT[w8]; bsum += redB[w8]; }
            const double sa = ssum - bsum;
            misc[L::M_s0 + arm] = sa;
            if (NARM == 2) xch_post(xown + 8 + it, sa);
            else misc[L::M_xtT] = (misc[L::M_baseT] + sa) / misc[L::M_delta];      // one arm: x~_T right here (read in P4)
        }
        const LaneC3 c3 = lane_c3(tl);
        QB(1); __syncthreads(); QS(1);
        // ---- P3: r_I = b_I - part (every wave its own copy), y_I = S^-1 r_I ----
        if (NARM == 2 && use_xT && wv == 7) {                                  // x~_T of the bordered solve, once per workgroup (needs the partner's share)
            const double xT7 = border_xT(it, lane);
            if (lane == 63) misc[L::M_xtT] = xT7;
        }
        double yi;
        {
            QM0();
            double *rIw = lds + L::oRIw;                                       // (every wave writes all of it, with identical values)
            {
                int l8 = lane;
                asm volatile("" : "+v"(l8));
                const double *src = lds + L::oRhsI + l8;
                double rv[8];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    rv[4 * u] = ldv(src + 64 * u); rv[4 * u + 1] = ldv(src + (L::oPA - L::oRhsI) + 64 * u);
                    rv[4 * u + 2] = ldv(src + (L::oPB - L::oRhsI) + 64 * u); rv[4 * u + 3] = ldv(src + (L::oDP - L::oRhsI) + 64 * u);
                }
                rIw[l8] = ((rv[0] - rv[1]) - rv[2]) - rv[3];
                rIw[l8 + 64] = ((rv[4] - rv[5]) - rv[6]) - rv[7];
            }
            wave_sync();
            QM(5);
            const LaneC3 &c = c3;
            yi = s_blk<SC>(sm, lds + c.rop);
            QM(6);
            lds[c.ysl] = yi;                                                   // (lanes without an output row: pad slot nI + 1)
        }
        const LaneC4 c4 = lane_c4(tl, px, py);
        QB(2); __syncthreads(); QS(2);
        // ---- P4: x_J = G (b_J - K_JC y_I); x~ = y - w x~_T ----
        {
            // (every LDS read that does not depend on this phase's own writes is issued first: one round trip, not five)
            const LaneC4 &c = c4;
            constexpr int dW = L::oWv - L::oXt;                                // w lives at a fixed distance from x~
            const double wds = ldv(lds + c.xds + dW), wdg = ldv(lds + c.xdg + dW);
            double dk[4], dy[4];
#pragma unroll
            for (int d = 0; d < 4; d++) { dk[d] = ldv(lds + c.p4k + d); dy[d] = ldv(lds + c.p4y + d); }
            const double *kr = lds + c.kjr, *yc = lds + c.ycb;
            const double k0 = ldv(kr), k1 = ldv(kr + 1), k2 = ldv(kr + 2), k3 = ldv(kr + 3);
            const double y0 = ldv(yc), y1 = ldv(yc + 14), y2 = ldv(yc - 7), y3 = ldv(yc + 7);
            double cr = ldv(lds + c.bjr);
            const double xT = use_xT ? ldv(misc + L::M_xtT) : 0.0;
            lds[c.xds] = yi - wds * xT;                                        // interface rows (others: the pad slot)
            if (tid < N) xt[NS * tid + 21] = xT;                               // x~_T, once per node: the 22nd operand of the path rows
            if (wv < NSEG) {
                // dense blocks (rows u_3s x x_3s; for the last segment also the U block x x_{N-1}): a quarter row per lane
                const double ad = (dk[0] * dy[0] + dk[1] * dy[1]) + (dk[2] * dy[2] + dk[3] * dy[3]);
                lds[c.p4d] = sum4(ad);                                         // (lanes 1..3 of a quad, lanes without a row: pad slot)
                cr -= (k0 * y0 + k1 * y1) + (k2 * y2 + k3 * y3);
                wave_sync();
                lds[c.tslot] = cr - ldv(lds + c.p4r);
                wave_sync();
                const double xj = g_blk<SMALL>(mm, lds + c.op4);
                lds[c.xdg] = xj - wdg * xT;                                    // (lanes without a row: the pad slot)
            }
        }
        QB(3); __syncthreads(); QS(3);
    };
    // lane t owns arm variables t, t + 512 and general rows t, t + 512 (ADMM state in registers, constants in LDS).  One packed
    // word per variable: node slot (16 bits) | offset in the node (5) | column of D for the rows of its own segment, 16 = none
    // (5) | the same for the previous segment (5) | has a -ts T term (1); one per row: first operand slot (16) | i = k % 3 (2).
    constexpr int NV = (na + 511) / 512, NR = (ma + 511) / 512;
    static_assert(NV <= 2 && NR <= 2, "two variables and two rows per lane at most");
    // WHICH lanes own a second item (N = 25: 13 variables, and the rows that do not fit the first pass): lanes SEC0V .. (wave 6) / SEC0R .. (waves 4, 5)
    // since round 5.  Their first-pass rows are dynamics rows (7 LDS reads); until round 4 the surplus sat on lanes 0 .. of wave 0, whose first pass
    // is a path row (44 reads), so that wave's two passes in phase E set the phase for all eight: -8.5 % per QP with the surplus on the dynamics-row waves
    // (tools/dual_fixed.py: 4.47 -> 4.09 ms per 700 iterations).
#ifndef MPCMP_Q3_SEC0V
#define MPCMP_Q3_SEC0V 384
#endif
#ifndef MPCMP_Q3_SEC0R
#define MPCMP_Q3_SEC0R 256
#endif
    constexpr int SEC0V = NV == 2 ? MPCMP_Q3_SEC0V : 0, SEC0R = NR == 2 ? MPCMP_Q3_SEC0R : 0;
    auto iv2 = [&](int t, int h) -> int { return h == 0 ? t : (t >= SEC0V ? 512 + (t - SEC0V) : (1 << 20)); };      // variable index of lane t's h-th variable
    // Row slots q (row_of: the 8 N path rows, 44 LDS reads each, then the dynamics rows, 7 reads).  At N = 25 the 200 path rows end inside wave 3: with
    // slot = lane that wave held 8 path rows and 56 dynamics rows, ran BOTH instruction streams in phase E and was the last at its barrier, while the
    // dynamics-row waves 4 .. 7 were done in half the time.  Now the lanes behind the last path row of that wave own no first row (PW = 256), the dynamics
    // rows start at the next wave, and what does not fit the first pass (80 rows) is the second row of the lanes from SEC0R = PW on (waves 4 and 5).
    constexpr int PW = NR == 2 ? (8 * N + 63) / 64 * 64 : 8 * N;                 // first lane of the dynamics rows
    constexpr int DR1 = NR == 2 ? 512 - PW : ma - 8 * N, DRX = (ma - 8 * N) - DR1;   // dynamics rows of the first pass / of the second
    static_assert(NR == 1 || (DRX >= 0 && DRX <= 96 && SEC0R == PW), "second rows");
    auto ir2 = [&](int t, int h) -> int {      // row slot of lane t's h-th row (none: 1 << 20)
        if (NR == 1) return h == 0 ? t : (1 << 20);
        if (h == 0) return t < 8 * N ? t : (t >= PW ? 8 * N + (t - PW) : (1 << 20));
        return (t >= SEC0R && t - SEC0R < DRX) ? 8 * N + DR1 + (t - SEC0R) : (1 << 20);
    };
    // Rows in lane order: the 8 N path rows first (44 LDS reads each), then the dynamics rows (7 reads): the few lanes that own a
    // second row (N = 25: 24 of them) get a cheap one.  The state of a lane's second variable / row lives in LDS (the 10 registers
    // it would take in EVERY lane are needed elsewhere).
    auto row_of = [&](int q) -> int { return q < 8 * N ? meq + q : q - 8 * N; };
    double xv0 = 0, zb0 = 0, yb0 = 0, zg0 = 0, yg0 = 0;             // (z_b .. y_g: registers unless L::STL)
    double *stz = lds + L::oSt, *sty = stz + 512, *stg = stz + 1024, *sth = stz + 1536;
    double *s1x = lds + L::oS1, *s1z = s1x + 32, *s1y = s1x + 64, *s1zg = s1x + 96, *s1yg = s1x + 192;
    static_assert(na - 512 <= 32 && SEC0V + 32 <= 512 && SEC0R + 96 <= 512, "second-pass state");
    static_assert(NV == 1 || 14 * N <= 512, "the lanes' second variables are controls (col_gather_u)");
    unsigned dv[NV], dr[NR];
#pragma unroll
    for (int h = 0; h < NV; h++) {
        const int v = iv2(tid, h);
        unsigned d = 0;
        if (v < na) {
            const bool isx = v < 14 * N;
            const int k = isx ? v / 14 : (v - 14 * N) / 7, o = isx ? v % 14 : 14 + (v - 14 * N) % 7, j = k % 3;
            const int ja = (isx && (j != 0 || k < N - 1)) ? j : 16, jb = (isx && j == 0 && k > 0) ? 3 : 16;
            const int fT = (o >= 7 && k <= N - 2) ? 1 : 0;
            d = (unsigned)(NS * k + o) | ((unsigned)o << 16) | ((unsigned)ja << 21) | ((unsigned)jb << 26) | ((unsigned)fT << 31);
        }
        dv[h] = d;
    }
#pragma unroll
    for (int h = 0; h < NR; h++) {
        const int r = ir2(tid, h) < ma ? row_of(ir2(tid, h)) : ma;
        unsigned d = 0;
        if (r < meq) { const int k = r / 14, rr = r % 14; d = (unsigned)(NS * 3 * (k / 3) + rr) | ((unsigned)(k % 3) << 16); }
        else if (r < ma) d = (unsigned)(NS * ((r - meq) >> 3));
        // bit 20: rho of the lane's variable is rho_eq (x_0, or any box narrower than 1e-4); bit 21: rho of the lane's row is rho_eq
        const int v = iv2(tid, h);
        if (v < na) { double ha, rb, lo, hi; var_h(v, ha, rb, lo, hi); if (rb == rho_eq) d |= 1u << 20; }
        if (r < ma && (r < meq || lds[L::oUg + r] - lds[L::oLg + r] < 1e-4)) d |= 1u << 21;
        dr[h] = d;
    }
    // (A^T w)[v] without the T row; w in node order
    auto col_gather = [&](const double *w, unsigned d) -> double {
        const int av = d & 0xFFFF, o = (d >> 16) & 31, ja = (d >> 21) & 31, jb = (d >> 26) & 31;
        const int nb = av - o;
        const int ia = av - NS * ja, ib = av - 3 * NS, it_ = av - 7;
        const double *wa = w + (ia > 0 ? ia : 0), *wb = w + (ib > 0 ? ib : 0), *wt = w + (it_ > 0 ? it_ : 0);
        const double *ca = cD + ja, *cb = cD + jb;                              // (16: the zero rows behind D)
        const double ct = (d >> 31) ? -tsT : 0.0;
        const double *gc = gkl + 8 * nb + o, *wp = w + nb + 14;
        const double a0 = ldv(ca), a1 = ldv(ca + 4), a2 = ldv(ca + 8), u0 = ldv(wa), u1 = ldv(wa + NS), u2 = ldv(wa + 2 * NS);
        const double b0 = ldv(cb), b1 = ldv(cb + 4), b2 = ldv(cb + 8), v0 = ldv(wb), v1 = ldv(wb + NS), v2 = ldv(wb + 2 * NS);
        const double wtv = ldv(wt);
        double sacc = (a0 * u0 + a1 * u1) + (a2 * u2 + ct * wtv), sac2 = (b0 * v0 + b1 * v1) + b2 * v2;
        constexpr int QB_ = SMALL ? 4 : 8;
#pragma unroll
        for (int q0 = 0; q0 < 8; q0 += QB_) {
            double g8[QB_], w8[QB_];
#pragma unroll
            for (int q = 0; q < QB_; q++) { g8[q] = ldv(gc + (q0 + q) * GS); w8[q] = ldv(wp + q0 + q); }
#pragma unroll
            for (int q = 0; q < QB_; q += 2) { sacc += g8[q] * w8[q]; sac2 += g8[q + 1] * w8[q + 1]; }
        }
        return sacc + sac2;
    };
    // the same for a control variable (no differentiation-matrix terms: their coefficient rows are the zero rows behind D, and the generic form reads
    // them and their six operands all the same): the lanes' SECOND variables (N = 25: controls of the last two nodes) are all of this kind, and their
    // pass is on the critical path of phase A.  Bitwise the generic result (0 * finite + x = x).
    auto col_gather_u = [&](const double *w, unsigned d) -> double {
        const int av = d & 0xFFFF, o = (d >> 16) & 31, nb = av - o, it_ = av - 7;
        const double ct = (d >> 31) ? -tsT : 0.0;
        const double *gc = gkl + 8 * nb + o, *wp = w + nb + 14;
        const double wtv = ldv(w + (it_ > 0 ? it_ : 0));
        double g8[8], w8[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { g8[q] = ldv(gc + q * GS); w8[q] = ldv(wp + q); }
        double sacc = ct * wtv, sac2 = 0.0;
#pragma unroll
        for (int q = 0; q < 8; q += 2) { sacc += g8[q] * w8[q]; sac2 += g8[q + 1] * w8[q + 1]; }
        return sacc + sac2;
    };
    // (A x)[r]; x in node order (slot 21 of every node: x_T)
    auto row_dot = [&](const double *xe, unsigned d, int r, double cf) -> double {
        double sacc;
        const int a0 = d & 0xFFFF;
        if (r < meq) {
            const int i = (d >> 16) & 3;
            const double *x0 = xe + a0, *cr = cD + 4 * i;
            const double c0 = ldv(cr), c1 = ldv(cr + 1), c2 = ldv(cr + 2), c3 = ldv(cr + 3);
            const double x_0 = ldv(x0), x_1 = ldv(x0 + NS), x_2 = ldv(x0 + 2 * NS), x_3 = ldv(x0 + 3 * NS), xf_ = ldv(x0 + NS * i + 7);
            const double xT_ = ldv(xe + 21);
            sacc = ((c0 * x_0 + c1 * x_1) + (c2 * x_2 + c3 * x_3)) + (cf * xT_ - tsT * xf_);
        } else {
            const double *gr = gkl + (r - meq) * GS, *xk = xe + a0;
            sacc = 0.0;
            double sac2 = 0.0;
            // (in parts: all 22 operand pairs in flight at once would take 88 registers)
            constexpr int RB_ = SMALL ? 4 : 8;
#pragma unroll
            for (int c0 = 0; c0 < 22; c0 += RB_) {
                double gq[RB_], xq[RB_];
#pragma unroll
                for (int c = 0; c < RB_; c++) if (c0 + c < 22) { gq[c] = ldv(gr + c0 + c); xq[c] = ldv(xk + c0 + c); }
                // (two accumulation chains: a single dependent FP64 FMA chain does not fill the pipeline at two waves per SIMD)
#pragma unroll
                for (int c = 0; c < RB_; c += 2) {
                    if (c0 + c < 22) sacc += gq[c] * xq[c];
                    if (c0 + c + 1 < 22) sac2 += gq[c + 1] * xq[c + 1];
                }
            }
            sacc += sac2;
        }
        return sacc;
    };
    auto w_slot = [&](unsigned d, int r) -> int {                              // slot of row r in the node-ordered w
        return r < meq ? (int)(d & 0xFFFF) + NS * (int)((d >> 16) & 3) : (int)(d & 0xFFFF) + 14 + ((r - meq) & 7);
    };
    const double inv_eq = 1.0 / rho_eq, inv_in = 1.0 / rho_in;     // (the two values 1 / rho takes; the oracle divides, both round the same way)
    double *ys = lds + L::oRhsJ;                                    // duals of the rows at the termination tests (node order), over the rhs vectors
    load_rows();
    solve(false, 0, tid, pk_x, pk_y, lane_c1(tid));   // K_0 w = k (the T border)
    finish_border();
#ifdef MPCMP_STAMPS
    for (int k = 0; k < 8; k++) st_acc[k] = st_busy[k] = 0;
    QS(6);
#endif
    if (cfg.qp_warm_start) {     // warm duals (mpcmp_config.qp_warm_start): y_0 = lambda_k, x_0 = 0, z_0 = clip(0, l, u); w_0 = rho z_0 - y_0 published
        const double *lam_vars = ws.lam + (size_t)b * mn_tot + NARM * ma + arm * na;
        double tp = 0.0;
#pragma unroll
        for (int h = 0; h < NR; h++) {
            if (ir2(tid, h) < ma) {
                const int r = row_of(ir2(tid, h));
                const double rr = (dr[h] >> 21) & 1u ? rho_eq : rho_in, yg = lam_rows[r], zg = clip(0.0, lds[L::oLg + r], lds[L::oUg + r]);
                const double w = rr * zg - yg;
                wg[w_slot(dr[h], r)] = w;
                tp += lds[L::oCf + r] * w;
                if (h) { s1zg[tid - SEC0R] = zg; s1yg[tid - SEC0R] = yg; } else if (L::STL) { stg[tid] = zg; sth[tid] = yg; } else { zg0 = zg; yg0 = yg; }
            }
        }
#pragma unroll
        for (int h = 0; h < NV; h++) {
            const int v = iv2(tid, h);
            if (v < na) {
                const double yb = lam_vars[v], zb = clip(0.0, lds[L::oLb + v], lds[L::oUb + v]);
                if (h) { s1z[tid - SEC0V] = zb; s1y[tid - SEC0V] = yb; } else if (L::STL) { stz[tid] = zb; sty[tid] = yb; } else { zb0 = zb; yb0 = yb; }
            }
        }
        if (L::P8) { tp = sum8(tp); redT[8 * wave + (lane >> 3)] = tp; }
        else { tp = wave_sum(tp); if (lane == 0) redT[wave] = tp; }
        if (tid == 511) {       // the shared variable T (replicated in the arm workgroups of the OCP, identical arithmetic)
            const double yT = ws.lam[(size_t)b * mn_tot + mn_tot - 1], zT = clip(0.0, misc[L::M_lbT], misc[L::M_ubT]);
            misc[L::M_zbT] = zT; misc[L::M_ybT] = yT;
            misc[L::M_baseT] = -1.0 + (misc[L::M_rbT] * zT - yT);
        }
        __syncthreads();
    }
    // The hot loop is the INNER loop (one termination-test period): it contains nothing but the five phases; the test itself
    // sits in the outer loop.
    int it = 0, done = 0, nchk = 0;
    while (it < cfg.qp_iters && !done) {
        const int cnt = cfg.qp_iters - it < cfg.check_every ? cfg.qp_iters - it : cfg.check_every;
#ifndef MPCMP_QP3_NO_RELOAD
        load_rows();
#endif
        for (int k = 0; k < cnt; k++) {
            // opaque copies of the lane index and of the packed words: what is derived from them stays inside the iteration
            int sio = tid, pkx = pk_x, pky = pk_y;
            unsigned dvo[NV], dro[NR];
            asm volatile("" : "+v"(sio), "+v"(pkx), "+v"(pky));
#pragma unroll
            for (int h = 0; h < NV; h++) { dvo[h] = dv[h]; asm volatile("" : "+v"(dvo[h])); }
#pragma unroll
            for (int h = 0; h < NR; h++) { dro[h] = dr[h]; asm volatile("" : "+v"(dro[h])); }
            // ---- A: rhs = sigma x - q + rho_b z_b - y_b + A^T w ----
            {
                double bp = 0.0;
#pragma unroll
                for (int h = 0; h < NV; h++) {
                    const int v = iv2(sio, h);
                    if (v < na) {
                        const double xx = h ? s1x[sio - SEC0V] : xv0, zz = h ? s1z[sio - SEC0V] : (L::STL ? ldv(stz + sio) : zb0), yy = h ? s1y[sio - SEC0V] : (L::STL ? ldv(sty + sio) : yb0);
                        const double rbv = (dro[h < NR ? h : 0] >> 20) & 1u ? rho_eq : rho_in, wv_ = ldv(wvv + (dvo[h] & 0xFFFF));     // (issued with the gather's reads)
                        const double r = (sigma * xx + (rbv * zz - yy)) + (h ? col_gather_u(wg, dvo[h]) : col_gather(wg, dvo[h]));
                        lds[h ? rpos[v] : (int)((unsigned)pky >> 16)] = r;
                        bp += wv_ * r;
                    }
                }
                if (L::P8) { bp = sum8(bp); redB[8 * wave + (lane >> 3)] = bp; }      // (all eight lanes of a group hold the sum and store it: no exec mask at the end of the phase)
                else { bp = wave_sum(bp); if (lane == 0) redB[wave] = bp; }      // (valid in lanes 0..15)
            }
            const LaneC1 c1 = lane_c1(sio);
            QB(0); __syncthreads(); QS(0);
            solve(true, it + 1 + k, sio, pkx, pky, c1);
            // ---- E: z~ = A x~, relaxation, projection, dual update ----
            {
                double tp = 0.0;
                {   // first row and first variable of the lane: every constant it needs is read up front (lanes without one read
                    // valid neighbouring words)
                    const int q0 = ir2(sio, 0);
                    const int r = q0 < ma ? row_of(q0) : 0;
                    const double rr = (dro[0] >> 21) & 1u ? rho_eq : rho_in, lg = ldv(lds + L::oLg + r), ug = ldv(lds + L::oUg + r), cf = ldv(lds + L::oCf + r);
                    const double xtv = ldv(xt + (dvo[0] & 0xFFFF)), rb = (dro[0] >> 20) & 1u ? rho_eq : rho_in, lb = ldv(lds + L::oLb + sio), ub = ldv(lds + L::oUb + sio);
                    double zgv = zg0, ygv = yg0, zbv = zb0, ybv = yb0;
                    if (L::STL) { zgv = ldv(stg + sio); ygv = ldv(sth + sio); zbv = ldv(stz + sio); ybv = ldv(sty + sio); }
                    if (q0 < ma) {
                        const double zt = row_dot(xt, dro[0], r, cf);
                        const double zr = alpha * zt + (1.0 - alpha) * zgv;
                        const double zn = clip(zr + ygv * (rr == rho_eq ? inv_eq : inv_in), lg, ug);
                        ygv += rr * (zr - zn);
                        zgv = zn;
                        if (L::STL) { stg[sio] = zgv; sth[sio] = ygv; } else { zg0 = zgv; yg0 = ygv; }
                        const double w = rr * zgv - ygv;
                        wg[w_slot(dro[0], r)] = w;
                        tp += cf * w;
                    }
                    if (sio < na) {
                        xv0 = alpha * xtv + (1.0 - alpha) * xv0;
                        const double zr = alpha * xtv + (1.0 - alpha) * zbv;
                        const double zn = clip(zr + ybv * (rb == rho_eq ? inv_eq : inv_in), lb, ub);
                        ybv += rb * (zr - zn);
                        zbv = zn;
                        if (L::STL) { stz[sio] = zbv; sty[sio] = ybv; } else { zb0 = zbv; yb0 = ybv; }
                    }
                }
                if (NR == 2 && ir2(sio, 1) < ma) {              // second row (N = 25: 24 lanes), state in LDS
                    const int r = row_of(ir2(sio, 1));
                    double zg = s1zg[sio - SEC0R], yg = s1yg[sio - SEC0R];
                    const double rr = (dro[NR - 1] >> 21) & 1u ? rho_eq : rho_in, cf = lds[L::oCf + r];
                    const double zt = row_dot(xt, dro[NR - 1], r, cf);
                    const double zr = alpha * zt + (1.0 - alpha) * zg;
                    const double zn = clip(zr + yg * (rr == rho_eq ? inv_eq : inv_in), lds[L::oLg + r], lds[L::oUg + r]);
                    yg += rr * (zr - zn);
                    zg = zn;
                    s1zg[sio - SEC0R] = zg; s1yg[sio - SEC0R] = yg;
                    const double w = rr * zg - yg;
                    wg[w_slot(dro[NR - 1], r)] = w;
                    tp += cf * w;
                }
                if (NV == 2 && iv2(sio, 1) < na) {              // second variable (N = 25: 13 lanes)
                    const int v = iv2(sio, 1);
                    double xx = s1x[sio - SEC0V], zz = s1z[sio - SEC0V], yy = s1y[sio - SEC0V];
                    const double xtv = xt[dvo[NV - 1] & 0xFFFF], rb = (dro[NR - 1] >> 20) & 1u ? rho_eq : rho_in;
                    xx = alpha * xtv + (1.0 - alpha) * xx;
                    const double zr = alpha * xtv + (1.0 - alpha) * zz;
                    const double zn = clip(zr + yy * (rb == rho_eq ? inv_eq : inv_in), lds[L::oLb + v], lds[L::oUb + v]);
                    yy += rb * (zr - zn);
                    zz = zn;
                    s1x[sio - SEC0V] = xx; s1z[sio - SEC0V] = zz; s1y[sio - SEC0V] = yy;
                }
                if (L::P8) { tp = sum8(tp); redT[8 * wave + (lane >> 3)] = tp; }
                else { tp = wave_sum(tp); if (lane == 0) redT[wave] = tp; }      // (read by the next iteration's P1: two barriers away)
                if (sio == 511) {       // the shared variable T: replicated in the arm workgroups of the OCP, identical arithmetic
                    const double xtv = xt[21], rb = misc[L::M_rbT];
                    double xx = misc[L::M_xT], zz = misc[L::M_zbT], yy = misc[L::M_ybT];
                    xx = alpha * xtv + (1.0 - alpha) * xx;
                    const double zr = alpha * xtv + (1.0 - alpha) * zz;
                    const double zn = clip(zr + yy / rb, misc[L::M_lbT], misc[L::M_ubT]);
                    yy += rb * (zr - zn);
                    zz = zn;
                    misc[L::M_xT] = xx; misc[L::M_zbT] = zz; misc[L::M_ybT] = yy;
                    misc[L::M_baseT] = (sigma * xx - 1.0) + (rb * zz - yy);
                }
            }
            QB(4); __syncthreads(); QS(4);
        }
        it += cnt;
        if (cnt == cfg.check_every) {
            // (opaque copies here as well: addresses derived from the plain descriptors would be hoisted out of the OUTER loop,
            // spilled across the hot loop and re-read from scratch one by one in every test)
            int sio = tid;
            unsigned dvc[NV], drc[NR];
            asm volatile("" : "+v"(sio));
#pragma unroll
            for (int h = 0; h < NV; h++) { dvc[h] = dv[h]; asm volatile("" : "+v"(dvc[h])); }
#pragma unroll
            for (int h = 0; h < NR; h++) { drc[h] = dr[h]; asm volatile("" : "+v"(drc[h])); }
            // ---- termination test: r_prim = ||[A;I]x - z||inf, r_dual = ||Hx + q + [A;I]^T y||inf (oracle/ocp.c admm) ----
            double sums[2] = {0.0, 0.0};              // T row: sum coefT_r y_r, sum ha_i x_i of this arm
            for (int i = sio; i < NX; i += 512) ys[i] = 0.0;          // (the unused slots of the node-ordered dual vector)
            __syncthreads();
#pragma unroll
            for (int h = 0; h < NR; h++) {
                if (ir2(sio, h) < ma) {
                    const int r = row_of(ir2(sio, h));
                    const double yg = h ? s1yg[sio - SEC0R] : (L::STL ? sth[sio] : yg0); ys[w_slot(drc[h], r)] = yg; sums[0] += lds[L::oCf + r] * yg;
                }
            }
#pragma unroll
            for (int h = 0; h < NV; h++) {
                const int v = iv2(sio, h);
                if (v < na) {
                    const double xx = h ? s1x[sio - SEC0V] : xv0;
                    double ha, rb_, lo_, hi_;
                    if (L::HAL) ha = lds[L::oHa + v]; else var_h(v, ha, rb_, lo_, hi_);
                    xt[dvc[h] & 0xFFFF] = xx; sums[1] += ha * xx;
                }
            }
            if (sio < N) xt[NS * sio + 21] = misc[L::M_xT];
            block_reduce_dpp<8, 2, false>(sums, redp, tid);      // (its barriers publish xt / ys)
            double mx[6] = {0, 0, 0, 0, 0, 0};                   // rp, |Ax|, |z|, rd, |Hx|, |A^T y|
            const double xTc = xt[21];
#pragma unroll
            for (int h = 0; h < NR; h++) {
                if (ir2(sio, h) < ma) {
                    const int r = row_of(ir2(sio, h));
                    const double zg = h ? s1zg[sio - SEC0R] : (L::STL ? stg[sio] : zg0), ax = row_dot(xt, drc[h], r, lds[L::oCf + r]);
                    mx[0] = fmax(mx[0], fabs(ax - zg)); mx[1] = fmax(mx[1], fabs(ax)); mx[2] = fmax(mx[2], fabs(zg));
                }
            }
#pragma unroll
            for (int h = 0; h < NV; h++) {
                const int v = iv2(sio, h);
                if (v < na) {
                    const double xx = h ? s1x[sio - SEC0V] : xv0, zz = h ? s1z[sio - SEC0V] : (L::STL ? stz[sio] : zb0), yy = h ? s1y[sio - SEC0V] : (L::STL ? sty[sio] : yb0);
                    double ha, rb_, lo_, hi_;
                    if (L::HAL) ha = lds[L::oHa + v]; else var_h(v, ha, rb_, lo_, hi_);
                    const double hx = (fabs(ha) + cfg.hess_reg) * xx + ha * xTc, aty = col_gather(ys, dvc[h]) + yy;
                    mx[0] = fmax(mx[0], fabs(xx - zz)); mx[1] = fmax(mx[1], fabs(xx)); mx[2] = fmax(mx[2], fabs(zz));
                    mx[3] = fmax(mx[3], fabs(hx + aty)); mx[4] = fmax(mx[4], fabs(hx)); mx[5] = fmax(mx[5], fabs(aty));
                }
            }
            done = check_tail(sums, mx, nchk++);
            for (int i = sio; i < NX; i += 512) ys[i] = 0.0;          // (the rhs vectors under the dual vector: their pads must read zero)
            __syncthreads();
            QS(5);
        }
    }
    // ---------------- results ----------------
#pragma unroll
    for (int h = 0; h < NV; h++) {
        const int v = iv2(tid, h);
        if (v < na) {
            ws.p[(size_t)b * n_tot + arm * na + v] = h ? s1x[tid - SEC0V] : xv0;
            ws.y[(size_t)b * mn_tot + NARM * ma + arm * na + v] = h ? s1y[tid - SEC0V] : (L::STL ? sty[tid] : yb0);
        }
    }
#pragma unroll
    for (int h = 0; h < NR; h++) {
        if (ir2(tid, h) < ma) ws.y[(size_t)b * mn_tot + arm * ma + row_of(ir2(tid, h))] = h ? s1yg[tid - SEC0R] : (L::STL ? sth[tid] : yg0);
    }
    if (tid == 511 && arm == 0) {
        ws.p[(size_t)b * n_tot + NARM * na] = misc[L::M_xT];
        ws.y[(size_t)b * mn_tot + mn_tot - 1] = misc[L::M_ybT];
    }
#ifdef MPCMP_STAMPS
    if (lane == 0 && arm == 0) {
        unsigned long long *o = ws.dbg + (size_t)b * MPCMP_DBG_WORDS;
        for (int k = 0; k < 8; k++) o[16 + wave * 8 + k] = st_busy[k];
        if (wave == 0) { for (int k = 0; k < 7; k++) o[k] = st_acc[k]; o[15] = it; }
    }
#endif
    {
        const int any = __syncthreads_or(dead ? 4 : 0);         // bit 2: the partner workgroup never answered
        if (tid == 0) {
            if (arm == 0) { ws.qpit[b] = it; ws.qp_total[b] += it; if (!done) atomicAdd(&ws.status[b], MPCMP_ST_CAP_ONE); }
            if (any) atomicOr(&ws.status[b], any);
        }
    }
}

}  // namespace mpcmp
