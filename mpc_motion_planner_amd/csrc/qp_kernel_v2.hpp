// qp_kernel_v2.hpp — k_qp2: the QP kernel specialised for NUM_SEG <= 4 (N <= 13), one OCP per workgroup of 1024
// threads (= one CU: 16 waves, 4 per SIMD, <=128 VGPRs).
//
// Same arithmetic as k_qp (solver_kernels.hpp) — assemble K, nested-dissection factorisation with explicit block
// inverses, OSQP-form ADMM — re-mapped for the CDNA4 execution model:
//   * every factor matrix lives in VGPRs for the whole ADMM loop ([G_s; E_s^T], E_s, S^-1: ~28k doubles spread
//     over the 1024 threads); LDS carries only the exchanged vectors and the (padded) path Jacobians;
//   * each thread owns a 2-row x k-column register block, so one LDS operand read feeds two FMAs; the
//     k-way partial sums are combined with DPP (quad_perm / row_half_mirror), not through LDS;
//   * waves are role-specialised (A1: G_s / E_s blocks of the interior solves; A2: E_s^T blocks grouped by interface entry
//     pair + the path rows; B: S^-1 blocks, variables, dynamics rows), 5 workgroup barriers per ADMM iteration, and the
//     roles that are idle in a phase prefetch their constants for the next one.
// Factorisation: one augmented symmetric sweep per segment (all segments concurrently, 4x4 register tiles) yields -G_s,
// E_s and the Schur contribution; a second sweep inverts S.  The factor never leaves the CU: the role threads copy their
// register blocks straight out of the factor area in LDS.
#pragma once
#include "solver_kernels.hpp"

namespace mpcmp {

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    // every lane of the 16-lane row is a valid source for the permutations used here, so `old` is never selected:
    // passing the source itself avoids a zero-initialising v_mov per half
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sum2(double x) { return x + dpp_mov<0xB1>(x); }                  // lanes i, i^1
__device__ __forceinline__ double sum4(double x) { x += dpp_mov<0xB1>(x); return x + dpp_mov<0x4E>(x); }
__device__ __forceinline__ double sum8(double x) { x = sum4(x); return x + dpp_mov<0x141>(x); }   // + row_half_mirror
// Two-row quad reduction.  A lane of a quad holds partial sums of TWO rows; lanes store the row of their own parity first
// ("own": row 2p + (lane & 1)) and the other row second, so the exchange with the xor-1 neighbour completes both rows at
// once: even lanes end with the total of row 2p, odd lanes with that of row 2p+1 -- half the DPP traffic of two sum4()
// and no select afterwards (the summation order per row is the one of sum4).
__device__ __forceinline__ double quad_sum2(double own, double other) {
    const double s = own + dpp_mov<0xB1>(other);
    return s + dpp_mov<0x4E>(s);
}
// value of lane ^ 4 (parity preserving, unlike row_half_mirror): row_shl:4 into the even banks, row_shr:4 into the odd ones
__device__ __forceinline__ double dpp_xor4(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    int tl = __builtin_amdgcn_update_dpp(lo, lo, 0x104, 0xf, 0x5, false), th = __builtin_amdgcn_update_dpp(hi, hi, 0x104, 0xf, 0x5, false);
    tl = __builtin_amdgcn_update_dpp(tl, lo, 0x114, 0xf, 0xA, false); th = __builtin_amdgcn_update_dpp(th, hi, 0x114, 0xf, 0xA, false);
    return __hiloint2double(th, tl);
}
__device__ __forceinline__ double read_lane(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double x) {      // the total, valid in lanes 0..15 (no LDS permutes, no index registers)
    x = sum8(x); x += dpp_mov<0x140>(x);                                                           // row_mirror: 16 lanes
    return (x + read_lane(x, 16)) + (read_lane(x, 32) + read_lane(x, 48));
}

// reciprocal of a positive, normal-range pivot: hardware estimate + two Newton steps (the full IEEE division sequence is
// ~25 dependent instructions on the critical path of every sweep step; scaling/denormal handling is not needed here)
__device__ __forceinline__ double pivot_rcp(double d) {
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    x = fma(fma(-d, x, 1.0), x, x);
    return x;
}

// 16-byte LDS read of two consecutive doubles (ds_read_b128: full LDS rate; ds_read2_b64 runs at half rate)
struct alignas(16) D2 { double x, y; };
__device__ __forceinline__ D2 lds2(const double *p) { return *reinterpret_cast<const D2 *>(__builtin_assume_aligned(p, 16)); }

template <bool MAX>
__device__ __forceinline__ double red_op(double a, double b) { return MAX ? fmax(a, b) : a + b; }
template <bool MAX>
__device__ __forceinline__ double reduce16(double x) {      // all 16 lanes of a DPP row end up with the result
    x = red_op<MAX>(x, dpp_mov<0xB1>(x)); x = red_op<MAX>(x, dpp_mov<0x4E>(x));
    x = red_op<MAX>(x, dpp_mov<0x141>(x)); x = red_op<MAX>(x, dpp_mov<0x140>(x));
    return x;
}
// Workgroup reduction of K <= 8 values for 16 waves with a small register footprint: per-wave DPP reduction,
// one LDS slot per (wave, k), then wave 0 combines the 16 partials of each k inside one DPP row. 3 barriers.
// red: >= 16*K + K doubles.
template <int K, bool MAX>
__device__ __forceinline__ void block_reduce16(double (&v)[K], double *red, int tid) {
    static_assert(K <= 8, "K");
#pragma unroll
    for (int k = 0; k < K; k++) {
        double x = reduce16<MAX>(v[k]);
        x = red_op<MAX>(x, __shfl_xor(x, 16));
        x = red_op<MAX>(x, __shfl_xor(x, 32));
        if ((tid & 63) == 0) red[(tid >> 6) * K + k] = x;
    }
    __syncthreads();
    if (tid < 64) {
        const int w = tid & 15, k0 = tid >> 4;
        double a = (k0 < K) ? red[w * K + k0] : 0.0;
        double b = (k0 + 4 < K) ? red[w * K + k0 + 4] : 0.0;
        a = reduce16<MAX>(a); b = reduce16<MAX>(b);
        if (w == 0) {
            if (k0 < K) red[16 * K + k0] = a;
            if (k0 + 4 < K) red[16 * K + k0 + 4] = b;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = red[16 * K + k];
    __syncthreads();
}

// Workgroup reduction used inside the ADMM loop, where only the waves >= FIRSTW hold non-identity values (role A1
// contributes nothing): contributing waves reduce inside 8-lane groups (three DPP steps) and publish one partial per
// group; the 16 lanes of a DPP row then combine the partials of the 16 waves.  v_max_f64 is emitted directly: the
// operands are never signalling NaNs, so the canonicalising self-max the compiler adds to fmax() is dead weight.
// Two barriers; `red` (>= 128*K + 8 doubles) must not be shared with a reduction issued right before or after.
__device__ __forceinline__ double max_raw(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <bool MAX>
__device__ __forceinline__ double red_raw(double a, double b) { return MAX ? max_raw(a, b) : a + b; }
template <int K, bool MAX, int FIRSTW>
__device__ __forceinline__ void block_reduce_roles(double (&v)[K], double *red, int tid, bool contributes) {
    static_assert(K <= 8, "K");
    if (contributes) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            double x = v[k];
            x = red_raw<MAX>(x, dpp_mov<0xB1>(x)); x = red_raw<MAX>(x, dpp_mov<0x4E>(x)); x = red_raw<MAX>(x, dpp_mov<0x141>(x));
            if ((tid & 7) == 0) red[(tid >> 3) * K + k] = x;
        }
    }
    __syncthreads();
    if (tid < 16 * K) {
        const int w = tid & 15, k = tid >> 4;
        double a = 0.0;                                   // identity of both reductions (maxima are of magnitudes)
        if (w >= FIRSTW) {
            const double *pr = red + (w * 8) * K + k;
            a = pr[0];
#pragma unroll
            for (int j = 1; j < 8; j++) a = red_raw<MAX>(a, pr[j * K]);
        }
        a = red_raw<MAX>(a, dpp_mov<0xB1>(a)); a = red_raw<MAX>(a, dpp_mov<0x4E>(a));
        a = red_raw<MAX>(a, dpp_mov<0x141>(a)); a = red_raw<MAX>(a, dpp_mov<0x140>(a));
        if (w == 0) red[128 * K + k] = a;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = red[128 * K + k];
}

template <int NSEG>
struct Qp2 {
    using D = Dim<NSEG>;
    static constexpr int NT = 1024, NW = 16;
    static constexpr int NGQ = 25 * NSEG, NEQ = 15 * NSEG;          // G row-pair quads, E^T row-pair quads
    static constexpr int NA1 = (4 * NGQ + 63) / 64 * 64;           // role A1 threads (448 at NSEG=4)
    static constexpr int NA2 = (4 * NEQ + 63) / 64 * 64;           // role A2 threads (256)
    static constexpr int NB = NT - NA1 - NA2;                      // role B threads  (320)
    static constexpr int NPR = (D::nI + 1) / 2;                    // interface row pairs (39)
    static constexpr int GS = 24;                                  // row stride of the path Jacobians in LDS (22 + 2 zero pads)
    static constexpr int XS = 24;                                  // node-major x~ stride: [x_k(14) u_k(7) T pad pad]
    static constexpr int HS = 2;                                   // segments factorised concurrently
    static_assert(8 * NPR <= NB && D::n <= NB, "role B mapping");
    static_assert(16 * D::N <= 4 * NEQ && D::meq <= NB, "path-row lanes must fit the E^T quads, dynamics rows role B");
    // LDS (doubles)
    static constexpr int RS = 7;                               // padded stride of the dynamics-row coefficients
    static constexpr int oRv = 0;                              // [meq][RS]    dynamics rows: D_i0..D_i3, -ts*T, -ts*f  (V part 1)
    static constexpr int oGk = oRv + D::meq * RS;              // [N][8][GS]   path Jacobians                           (V part 2)
    static constexpr int oRhsJ = oGk + D::N * 8 * GS;          // [NSEG][56]   rhs, interior part (zero padded)
    static constexpr int oRhsI = oRhsJ + NSEG * 56;            // [80]         rhs, interface part
    static constexpr int oRI = oRhsI + 2 * ((D::nI + 2) / 2);  // [80]         r_I = b_I - sum_s E_s^T b_Js
    static constexpr int oXn = oRI + 80;                       // [N][XS]      x~ node-major
    static constexpr int oXx = oXn + D::N * XS;                // [N][XS]      x  node-major (termination tests)
    static constexpr int oWg = oXx + D::N * XS;                // [m]          w = rho z - y   (general rows)
    static constexpr int oYs = oWg + D::m;                     // [m]          y (termination tests)
    static constexpr int oTp = oYs + D::m;                     // [m]          coefT_r * w_r
    static constexpr int oMisc = oTp + D::m;                   // [16] 0: zero slot, 1: T base, 2: T column sum, 3: sum|ha|, 6..7 and 8..13: write-only pad slots (unconditional stores of lanes without an output)
    static constexpr int oRed = oMisc + 16;                    // [NW*8]
    static_assert(oMisc % 2 == 0, "16-byte stores into the pad slots");
    // factor area (must match build_streams): S | KJJ[NSEG] | KJC[HS] | Eh[HS] | scratch[32] | rdv[8]
    static constexpr int oS = oRed + NW * 8;                   // packed S, then -(S^-1)
    static constexpr int oKJJ = oS + D::SP;                    // [NSEG][JP]
    static constexpr int oKJC = oKJJ + NSEG * D::JP;           // [NSEG][JC] = stream group 0 [HS][JC] | group 1 (the 'Eh' slot of build_streams); E_s in place after the sweep
    static constexpr int oEh = oKJC + HS * D::JC;              // [HS][JC]
    static constexpr int oScr = oEh + HS * D::JC;              // [32] partial sums of split entries
    static constexpr int oRdv = oScr + 32;                     // [2][4] pivot reciprocals of the running sweep (double buffered)
    static constexpr int oCol = (oRdv + 8 + 1) / 2 * 2;                      // [2][CB] current / next pivot column(s) of the sweep
    static constexpr int CB = NSEG * 80 > ((D::nI + 15) / 16 * 16) ? NSEG * 80 : ((D::nI + 15) / 16 * 16);   // column buffer: NSEG blocks of 78 (+2 pads)
    static constexpr int oEndF = oCol + 2 * CB;
    static_assert(oCol % 2 == 0 && CB % 2 == 0, "16-byte reads of the pivot column");
    // ADMM view, overlaying [oS, ...) once the factor blocks have been picked up by their owners:
    static constexpr int oPc = oS;                             // [5][NA2]   path-row lg, ug, rho, coefT, 1/rho
    static constexpr int oVc = oPc + 5 * NA2;                  // [13][NB]   variable role: cf, dA[3], dB[3], hd, ha, qv, lb, ub, 1/rho_b
    static constexpr int oGp = oVc + 13 * NB;                  // [N][XS]    path-row part of A^T w, node-major like x~
    static constexpr int oGpy = oGp + D::N * XS;               // [N][XS]    path-row part of A^T y (termination tests)
    static constexpr int oRedS = oGpy + D::N * XS;             // [128*2+8]  loop reductions: sums
    static constexpr int oRedM = oRedS + 128 * 2 + 8;          // [2][128*3+8]  loop reductions: maxima of the primal test, of the dual test
    static constexpr int RM3 = 128 * 3 + 8;
    static constexpr int oStamp = oRedM + 2 * RM3;             // [16]       cycle stamps (diagnostic builds only)
    static constexpr int oBusy = oStamp + 16;                  // [16][8]    per-wave busy cycles (diagnostic builds only)
    static constexpr int oEndA = oBusy + 128;
    static constexpr int size = oEndF > oEndA ? oEndF : oEndA;
    static_assert(size * 8 <= 160 * 1024 - 512, "LDS budget (a 256-byte static block precedes the dynamic region)");
    static_assert(oGk + D::N * 8 * GS < (1 << 14), "assembly stream operand offsets are 14 bits");
};

// shared context of the role groups
template <int NSEG>
struct Qp2Ctx {
    const mpcmp_config *cfg;
    WS ws;
    double *lds;
    int tid, b;
    double ts, tsT, rho_in, rho_eq, sigma, alpha;
};

#ifndef MPCMP_WPERM
#define MPCMP_WPERM 0xFEDCBA9876543210ull
#endif
// what-if profiling (tools/ablate.py): -DMPCMP_ABL=n removes one role's work in one phase (results are then wrong); the
// change in run time at a fixed iteration count is that piece's share of the critical path.  0 = product build.
#ifndef MPCMP_ABL
#define MPCMP_ABL 0
#endif
// LDS-queue staggering (tools/isa_phases.py): a phase lasts (LDS-array time of everything queued ahead of the critical role's operand reads) +
// (that role's chain).  The roles that are NOT on a phase's critical path sleep 64 x n cycles before they issue their own reads of that phase.
#ifndef MPCMP_SLP_P1
#define MPCMP_SLP_P1 0      /* role A1 before its P1 operand reads (critical: role A2) */
#endif
#ifndef MPCMP_SLP_P2
#define MPCMP_SLP_P2 0      /* role A1 before its P2 operand reads (critical: role B) */
#endif
#ifndef MPCMP_SLP_P3
#define MPCMP_SLP_P3 0      /* roles A2 and B before their constant prefetches in P3 (critical: role A1) */
#endif
#define MPCMP_SLEEP(n) do { if ((n) > 0) __builtin_amdgcn_s_sleep(n); } while (0)
#define ABL_ON(n) (MPCMP_ABL != (n) && MPCMP_ABL != 10)      /* 10: every piece off — the bare five-barrier loop with its termination tests */
#ifdef MPCMP_STAMPS
// the accumulators of the loop stamps live in LDS: role A1 has no registers to spare
#define STAMP2(slot) do { if (c.tid == 0) { const unsigned long long now_ = clock64(); stamp_acc[slot] += now_ - stamp_t; stamp_t = now_; } } while (0)
#ifndef MPCMP_STAMPS_LIGHT      // (-DMPCMP_STAMPS_LIGHT: wave 0's phase stamps only; they perturb the loop by < 1 %, the busy counters by ~30 %)
// per-wave busy time of each ADMM phase (barrier exit -> arrival at the phase's closing barrier): dbg[16 + wave*8 + phase]
#define BUSY_DECL unsigned long long busy_t = clock64(); \
    unsigned long long *busy_acc = reinterpret_cast<unsigned long long *>(c.lds + L::oBusy) + (c.tid >> 6) * 8; \
    if ((c.tid & 63) == 0) for (int k_ = 0; k_ < 8; k_++) busy_acc[k_] = 0
#define BUSY_SYNC(ph) do { if ((c.tid & 63) == 0) busy_acc[ph] += clock64() - busy_t; __syncthreads(); busy_t = clock64(); } while (0)
#define BUSY_DUMP do { if ((c.tid & 63) == 0) for (int k_ = 0; k_ < 8; k_++) \
    c.ws.dbg[(size_t)c.b * MPCMP_DBG_WORDS + 16 + (c.tid >> 6) * 8 + k_] = busy_acc[k_]; } while (0)
#else
#define BUSY_DECL do { } while (0)
#define BUSY_SYNC(ph) __syncthreads()
#define BUSY_DUMP do { } while (0)
#endif
#else
#define STAMP2(slot) do { } while (0)
#define BUSY_DECL do { } while (0)
#define BUSY_SYNC(ph) __syncthreads()
#define BUSY_DUMP do { } while (0)
#endif

// termination test shared by all roles (every thread contributes its maxima; result is workgroup-uniform)
// The test is taken in two stages (the result is the same conjunction): the PRIMAL residual first — on the bench workload it fails in 92 % of the
// tests (oracle count over 24 problems x 20 QPs: both fail 70 %, only the primal 22 %, only the dual 4 %, both pass 4 %) — and only if it passes the
// dual one, whose operands (the T column sums, A^T y of the path rows, the gathers of the variable lanes) then are not formed at all.
template <int NSEG>
__device__ __forceinline__ int qp2_primal_ok(const mpcmp_config &cfg, double (&mp)[3], double *lds, int tid, bool contributes) {
    using L = Qp2<NSEG>;
    block_reduce_roles<3, true, L::NA1 / 64>(mp, lds + L::oRedM, tid, contributes);
    return mp[0] <= cfg.eps_abs + cfg.eps_rel * fmax(mp[1], mp[2]) ? 1 : 0;
}
template <int NSEG>
__device__ __forceinline__ int qp2_dual_ok(const mpcmp_config &cfg, double (&md)[3], double *lds, int tid, bool contributes) {
    using L = Qp2<NSEG>;
    block_reduce_roles<3, true, L::NA1 / 64>(md, lds + L::oRedM + L::RM3, tid, contributes);
    return md[0] <= cfg.eps_abs + cfg.eps_rel * fmax(fmax(md[1], md[2]), 1.0) ? 1 : 0;
}

// ---- role A1: G row-pair quads — t = G_s b_J (P1) and x_J = t - E_s x_C (P3); wave 0 also sums the T column ----
template <int NSEG>
__device__ __forceinline__ void qp2_role_a1(const Qp2Ctx<NSEG> &c) {
    using D = Dim<NSEG>;
    using L = Qp2<NSEG>;
    constexpr int N = D::N, meq = D::meq, XS = L::XS;
    double *lds = c.lds;
    const mpcmp_config &cfg = *c.cfg;
    const int tid = c.tid;
    double *rhsJ = lds + L::oRhsJ, *xn = lds + L::oXn, *tpl = lds + L::oTp, *misc = lds + L::oMisc;
#ifdef MPCMP_STAMPS
    unsigned long long *stamp_acc = reinterpret_cast<unsigned long long *>(lds + L::oStamp), stamp_t = clock64();
#endif
    const int Q = tid >> 2, part = tid & 3;
    const bool act = Q < L::NGQ;
    const int seg = act ? Q / 25 : 0, lp = act ? Q % 25 : 0;
    // rows (2lp, 2lp+1) of G_s x columns part*14..+13 and of E_s x columns part*8..+7, all in registers (own row first)
    double m1[2][14], e1[2][8];
    {
        const double *Gn = lds + L::oKJJ + seg * D::JP, *Es = lds + L::oKJC + seg * D::JC;      // still in the factor area
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int row = 2 * lp + (a ^ (part & 1));      // [0]: the row of this lane's parity, [1]: the other (quad_sum2)
#pragma unroll
            for (int j = 0; j < 14; j++) {
                const int col = part * 14 + j;
                m1[a][j] = (act && row < 49 && col < 49) ? -Gn[packed(row, col)] : 0.0;
                if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);     // one-time loads: keep address temporaries few
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                // x_C = [x_3s | x_3s+3 | T] is read in place from the node-major x~ (see xa/xb/xc3 below): the last quad lane
                // takes columns 24..27 and, from the slot pair (u_6, T) of node 3s+3, column 28
                const int col = part < 3 ? part * 8 + j : (j < 4 ? 24 + j : (j == 5 ? 28 : 64));
                e1[a][j] = (act && row < 49 && col < 29) ? Es[row * 29 + col] : 0.0;
                if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();      // every owner has picked up its blocks from the factor area (S is read by role B)
#ifdef MPCMP_STAMPS
        if (tid == 0) for (int k = 0; k < 16; k++) stamp_acc[k] = 0;      // (the accumulators overlay the factor area)
#endif
        __syncthreads();      // LDS-resident constants published
    }
    int jdst = -1;      // where row 2lp+part of x_J goes in the node-major x~
    if (act && part < 2 && 2 * lp + part < 49) {
        const int v = c.ws.ext_of_int[49 * seg + 2 * lp + part];
        jdst = v < 14 * N ? (v / 14) * XS + v % 14 : ((v - 14 * N) / 7) * XS + 14 + (v - 14 * N) % 7;
    }
    double *xw = jdst >= 0 ? xn + jdst : misc + 7;     // unconditional store of x_J (exec masking costs more than the store)
    const double *bj = rhsJ + 56 * seg + 14 * part;
    // 16-byte reads of x_C: pairs 0,1 at xa, xa+2; pair 2 at xc3; pair 3 at xb
    const double *xn0 = xn + 3 * seg * XS, *xn1 = xn0 + 3 * XS;
    const double *xa = part == 0 ? xn0 : part == 1 ? xn0 + 8 : part == 2 ? xn1 + 2 : xn1 + 10;
    const double *xc3 = part == 3 ? xn1 + 20 : xa + 4;
    const double *xb = part == 0 ? xn0 + 6 : part == 1 ? xn1 : part == 2 ? xn1 + 8 : xn1 + 20;
    if (cfg.qp_warm_start) __syncthreads();          // (warm duals: roles A2 and B publish w_0 = rho z_0 - y_0)
    // Loop structure (all three roles alike): periods of check_every iterations in an inner loop that contains NO termination-test code, the test after
    // it.  (Until round 4 the test block sat inside the one loop and every iteration branched over it; whatever changed in that cold block moved the
    // hot loops by up to 2 %.)  A period cut short by qp_iters ends the loop untested, as before: tests happen at multiples of check_every.
    int it = 0, done = 0;
    BUSY_DECL;
    STAMP2(12);                 // role prologue: register blocks fetched from the factor scratch, constants published
    while (it < cfg.qp_iters) {
      const int period = MPCMP_PERIOD(cfg, it);
#pragma nounroll
      for (int kk = 0; kk < period; kk++) {
        // ---- A: wave 0 sums the T column of A^T w ----
        if (tid < 64 && ABL_ON(4)) {
            // all operand reads in flight at once (a rolled loop serialises one LDS round trip per 64 rows)
            constexpr int NR = (meq + 63) / 64;
            double tv[NR + 1];
#pragma unroll
            for (int q = 0; q < NR; q++) tv[q] = tpl[tid + 64 * q < meq ? tid + 64 * q : meq];     // dynamics rows (slot meq: zero)
            tv[NR] = lds[tid < N ? L::oGp + tid * XS + 21 : L::oMisc];  // path rows: per-node column sums (role A2)
            double sacc = 0.0;
#pragma unroll
            for (int q = 0; q <= NR; q++) sacc += tv[q];
            sacc = wave_sum(sacc);
            if (tid == 0) misc[2] = sacc;
        }
        BUSY_SYNC(0);
        STAMP2(3);
        // ---- P1 / P2: t = G_s b_J is not needed before P3, so its operand reads are spread over both phases (P1 is
        // bound by the LDS reads of b by roles A1 and A2 together, P2 only has role B's reads of r_I) ----
        double tq;              // (G_s b_J)[own row]
        {
#ifdef MPCMP_FOLDSIM
            constexpr int JS = 7;                                          // (what-if build: the whole product in P1, phase P2 does not exist)
#else
            constexpr int JS = 4;                                          // 16-byte operand pairs taken in P1
#endif
            double a0 = 0.0, a1 = 0.0;
            D2 bv[7];
            if (ABL_ON(6)) {
                MPCMP_SLEEP(MPCMP_SLP_P1);
#pragma unroll
                for (int j = 0; j < JS; j++) bv[j] = lds2(bj + 2 * j);
#pragma unroll
                for (int j = 0; j < JS; j++) {
                    a0 += m1[0][2 * j] * bv[j].x; a1 += m1[1][2 * j] * bv[j].x;
                    a0 += m1[0][2 * j + 1] * bv[j].y; a1 += m1[1][2 * j + 1] * bv[j].y;
                }
            }
            BUSY_SYNC(1);
            STAMP2(4);
            if (ABL_ON(9)) {
                MPCMP_SLEEP(MPCMP_SLP_P2);
#pragma unroll
                for (int j = JS; j < 7; j++) bv[j] = lds2(bj + 2 * j);
#pragma unroll
                for (int j = JS; j < 7; j++) {
                    a0 += m1[0][2 * j] * bv[j].x; a1 += m1[1][2 * j] * bv[j].x;
                    a0 += m1[0][2 * j + 1] * bv[j].y; a1 += m1[1][2 * j + 1] * bv[j].y;
                }
                tq = quad_sum2(a0, a1);
            } else { tq = a0 + a1; }
        }
        // ---- P2 (role B) ----
#ifndef MPCMP_FOLDSIM
        BUSY_SYNC(2);
#endif
        STAMP2(5);
        // ---- P3 ----
        if (ABL_ON(8)) {
            double a0 = 0.0, a1 = 0.0;
            D2 xv[4];
#pragma unroll
            for (int j = 0; j < 2; j++) xv[j] = lds2(xa + 2 * j);
            xv[2] = lds2(xc3); xv[3] = lds2(xb);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                a0 += e1[0][2 * j] * xv[j].x; a1 += e1[1][2 * j] * xv[j].x;
                a0 += e1[0][2 * j + 1] * xv[j].y; a1 += e1[1][2 * j + 1] * xv[j].y;
            }
            const double aq = quad_sum2(a0, a1);         // (E_s x_C)[own row]
            *xw = tq - aq;                                 // (lanes without a row: the write-only pad slot)
        }
        BUSY_SYNC(3);
        STAMP2(6);
        // ---- E (roles A2, B) ----
        BUSY_SYNC(4);
        STAMP2(7);
      }
      it += period;
      if (MPCMP_NO_TEST(cfg, period)) break;
      {
            __syncthreads();                             // (role B publishes x and the dynamics-row duals for the test)
            double mp[3] = {0, 0, 0};
            if (qp2_primal_ok<NSEG>(cfg, mp, lds, tid, false)) {
                double sums[2] = {0.0, 0.0};
                block_reduce_roles<2, false, L::NA1 / 64>(sums, lds + L::oRedS, tid, false);
                double md[3] = {0, 0, 0};
                done = qp2_dual_ok<NSEG>(cfg, md, lds, tid, false);
            }
      }
      STAMP2(8);
      if (done) break;
    }
    const bool capped = !done;                         // ran out of iterations without meeting the termination test
    BUSY_DUMP;
    if (tid == 0) { c.ws.qpit[c.b] = it; c.ws.qp_total[c.b] += it; if (capped) atomicAdd(&c.ws.status[c.b], MPCMP_ST_CAP_ONE); }
#ifdef MPCMP_STAMPS
    if (tid == 0) { unsigned long long *o = c.ws.dbg + (size_t)c.b * MPCMP_DBG_WORDS; for (int k = 3; k < 9; k++) o[k] = stamp_acc[k]; o[12] = stamp_acc[12]; o[15] = it; }
#endif
}

// ---- role A2: E^T row-pair quads — E_s^T b_J (P1) — and every general row (z~ = A x~, projection, duals) ----
template <int NSEG>
__device__ __forceinline__ void qp2_role_a2(const Qp2Ctx<NSEG> &c) {
    using D = Dim<NSEG>;
    using L = Qp2<NSEG>;
    constexpr int N = D::N, meq = D::meq, GS = L::GS, XS = L::XS;
    double *lds = c.lds;
    const mpcmp_config &cfg = *c.cfg;
    const int tid = c.tid, b = c.b;
    double *gkl = lds + L::oGk;
    double *rhsJ = lds + L::oRhsJ, *rhsI = lds + L::oRhsI, *rI = lds + L::oRI, *xn = lds + L::oXn, *xx = lds + L::oXx;
    const int et = tid - L::NA1, part = et & 3;
    // P1 lanes of this role: one lane group per PAIR of interface entries, spanning every segment the entries couple to,
    // so the group's reduction is the complete sum_s (E_s^T b_Js) of its entries and r_I is written without a further
    // phase.  Groups: T (column 28 of every segment, 4*NSEG lanes, one row), then the entries of the interior interface
    // nodes (two segments, 8 lanes per pair), then those of the first and last node (one segment, 4 lanes per pair).
    // A lane holds rows (ec0, ec1) of E_seg^T x columns part*14..+13 in registers.
    constexpr int TL = 4 * NSEG, G8 = (NSEG - 1) * 56;
    static_assert(TL + G8 + 56 == 4 * L::NEQ && (TL == 8 || TL == 16), "lane groups of role A2");
    int seg = 0, ec0 = -1, ec1 = -1, irow = -1;         // segment, E columns of the two rows, interface index of row 0
    double f8 = 0.0, f16 = 0.0;                         // reduction width beyond the quad, as multipliers
    int lig = et & 3;                                   // lane index inside the group
    if (et < TL) {
        seg = et >> 2; ec0 = 28; irow = D::nI - 1; lig = et;
        f8 = 1.0; f16 = TL == 16 ? 1.0 : 0.0;
    } else if (et < TL + G8) {
        const int e = et - TL, nd = 1 + e / 56, pr = (e % 56) / 8, half = (e % 8) / 4, cc = 2 * pr;
        seg = nd - 1 + half; ec0 = half == 0 ? 14 + cc : cc; ec1 = ec0 + 1; irow = 14 * nd + cc; lig = e % 8;
        f8 = 1.0;
    } else if (et < TL + G8 + 56) {
        const int e = et - TL - G8, side = e / 28, pr = (e % 28) / 4, cc = 2 * pr;
        seg = side == 0 ? 0 : NSEG - 1; ec0 = side == 0 ? cc : 14 + cc; ec1 = ec0 + 1; irow = (side == 0 ? 0 : 14 * NSEG) + cc;
    }
#ifdef MPCMP_FOLDSIM
    double m1[2][18];
    for (int a = 0; a < 2; a++) for (int j = 14; j < 18; j++) m1[a][j] = 1e-3 * (tid + j);
#else
    double m1[2][14];
#endif
    {
        const double *Es = lds + L::oKJC + seg * D::JC;
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int cc = (a ^ (lig & 1)) == 0 ? ec0 : ec1;      // own row first (quad_sum2); lig and part have the same parity
#pragma unroll
            for (int j = 0; j < 14; j++) {
                const int i = part * 14 + j;
                m1[a][j] = (cc >= 0 && i < 49) ? Es[i * 29 + cc] : 0.0;
                if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();          // (matches role A1/B: register blocks picked up)
    // writer lanes (0 and 1 of a group): r_I[i] = (b_I[i] + extra) - sum, extra = the T column sum for T, else the zero slot
    const int rdst = (lig < 2 && irow >= 0 && (lig == 0 || ec1 >= 0)) ? irow + lig : -1;
    const int xoff = (rdst == D::nI - 1) ? L::oMisc + 2 : L::oMisc;
    const int rsrc = rdst >= 0 ? rdst : 0;
    double *rw = rdst >= 0 ? rI + rdst : lds + L::oMisc + 6;
    // path rows: the row constants live in LDS (lane-transposed), only the ADMM state (z, y) of the owned row stays in registers
    const bool isPath = et < 16 * N;                    // four lanes per pair of path rows (six columns each)
    const int pk = et >> 4, prp = (et & 15) >> 2, pq = et & 3;
    const bool ownsRow = isPath && pq < 2;             // lanes 0,1 of the group own rows 2prp, 2prp+1
    double *pcl = lds + L::oPc + et;
    double zg = 0, yg = 0;
    int myrow = 0;
    if (ownsRow) {
        const int q = 2 * prp + pq;
        myrow = meq + 8 * pk + q;
        const double gv = c.ws.g[(size_t)b * 8 * N + 8 * pk + q];
        const double lg = cfg.lbg[q] - gv, ug = cfg.ubg[q] - gv;
        pcl[0] = lg; pcl[L::NA2] = ug;
        pcl[2 * L::NA2] = (ug - lg < 1e-4) ? c.rho_eq : c.rho_in;
        pcl[3 * L::NA2] = c.ws.Gk[((size_t)b * N + pk) * 176 + q * 22 + 21];
        pcl[4 * L::NA2] = 1.0 / pcl[2 * L::NA2];
    }
    if (isPath && prp == 0) {      // A^T w is read by the first rhs before any row has been updated
#pragma unroll
        for (int j = 0; j < 6; j++) lds[L::oGp + pk * XS + pq * 6 + j] = 0.0;
    }
    const int groff = isPath ? (pk * 8 + 2 * prp) * GS + pq * 6 : 0;     // columns 6pq .. 6pq+5 of the 24-wide padded rows
    const int xnoff = isPath ? pk * XS + pq * 6 : 0;
    // z~ of the owned row (lanes 0,1 of the quad) and, from the same Jacobian operands, this node's path-row part of
    // A^T w: every lane forms its six columns of g_row0*w0 + g_row1*w1, the four row pairs of the node (lane bits 2,3 of
    // the DPP row) are summed with two row rotations, and the lanes of pair 0 publish the node's 24 padded columns.
    double *padw = lds + L::oMisc + 8;                  // 48 bytes of write-only pad
    auto path_rows = [&](const D2 (&p0)[3], const D2 (&p1)[3], const double *xe, double *gdst, auto &&row_update) -> double {
        const double *xv = xe + xnoff;
        D2 x2[3];
#pragma unroll
        for (int j = 0; j < 3; j++) x2[j] = lds2(xv + 2 * j);
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            a0 += p0[j].x * x2[j].x; a1 += p1[j].x * x2[j].x;
            a0 += p0[j].y * x2[j].y; a1 += p1[j].y * x2[j].y;
        }
        const double ax = quad_sum2(a0, a1);                                // p0 = own row (2 prp + (pq & 1)), p1 = the other
        const double wq = row_update(ax);                                   // owners: the row's multiplier-like value
        const double w0 = dpp_mov<0x44>(wq), w1 = dpp_mov<0x11>(wq);        // quad broadcasts: owner of the own row, of the other row
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double cx = p0[j].x * w0 + p1[j].x * w1, cy = p0[j].y * w0 + p1[j].y * w1;
            cx += dpp_mov<0x128>(cx); cy += dpp_mov<0x128>(cy);             // row_ror:8
            cx += dpp_mov<0x124>(cx); cy += dpp_mov<0x124>(cy);             // row_ror:4
            { D2 o; o.x = cx; o.y = cy; *reinterpret_cast<D2 *>((prp == 0 ? gdst + xnoff : padw) + 2 * j) = o; }   // (pairs 1..3 of a node: pad)
        }
        return ax;
    };
    __syncthreads();          // constants published (matches the barrier of the other roles)
    const double alpha = c.alpha;
    const double *bj = rhsJ + 56 * seg + 14 * part;
    if (cfg.qp_warm_start) {     // warm duals (mpcmp_config.qp_warm_start): y_0 = lambda_k, z_0 = clip(0, l, u); the node's path-row part of A^T w_0
        if (ownsRow) { yg = c.ws.lam[(size_t)b * D::mn + myrow]; zg = clip(0.0, pcl[0], pcl[L::NA2]); }
        if (isPath) {
            D2 q0[3], q1[3];
            const double *g0 = gkl + groff + (pq & 1) * GS, *g1 = gkl + groff + (1 - (pq & 1)) * GS;
#pragma unroll
            for (int j = 0; j < 3; j++) { q0[j] = lds2(g0 + 2 * j); q1[j] = lds2(g1 + 2 * j); }
            const double w0_ = ownsRow ? pcl[2 * L::NA2] * zg - yg : 0.0;
            path_rows(q0, q1, xn, lds + L::oGp, [&](double) -> double { return w0_; });
        }
        __syncthreads();
    }
    int it = 0, done = 0;
    BUSY_DECL;
    while (it < cfg.qp_iters) {
      const int period = MPCMP_PERIOD(cfg, it);
#pragma nounroll
      for (int kk = 0; kk < period; kk++) {
        // ---- A (role B) ----
        BUSY_SYNC(0);
        // ---- P1 ----
        if (ABL_ON(5)) {
            double a0 = 0.0, a1 = 0.0;
#ifdef MPCMP_FOLDSIM
            constexpr int NJ = 9;                                          // (what-if build: a 2 x 18 block of W = S^-1 [-E^T | I]; numbers are garbage)
#else
            constexpr int NJ = 7;
#endif
#ifdef MPCMP_FOLDSIM
            const double bI0 = rhsI[rsrc], bI1 = lds[xoff];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j0 = h ? 5 : 0, j1 = h ? NJ : 5;
                D2 bv[5];
#pragma unroll
                for (int j = j0; j < j1; j++) bv[j - j0] = lds2(bj + 2 * j);
#pragma unroll
                for (int j = j0; j < j1; j++) {
                    a0 += m1[0][2 * j] * bv[j - j0].x; a1 += m1[1][2 * j] * bv[j - j0].x;
                    a0 += m1[0][2 * j + 1] * bv[j - j0].y; a1 += m1[1][2 * j + 1] * bv[j - j0].y;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#else
            D2 bv[NJ];
#pragma unroll
            for (int j = 0; j < NJ; j++) bv[j] = lds2(bj + 2 * j);          // all operand reads in flight first
            const double bI0 = rhsI[rsrc], bI1 = lds[xoff];                // (every lane reads: keeps the loads off the tail)
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                a0 += m1[0][2 * j] * bv[j].x; a1 += m1[1][2 * j] * bv[j].x;
                a0 += m1[0][2 * j + 1] * bv[j].y; a1 += m1[1][2 * j + 1] * bv[j].y;
            }
#endif
            double sq = quad_sum2(a0, a1);                  // even lanes: row ec0, odd lanes: row ec1
            sq += f8 * dpp_xor4(sq);                        // 8-lane groups: the other segment's quad
            sq += f16 * dpp_mov<0x128>(sq);                 // the T group of 16: row_ror:8 = lane ^ 8
            *rw = (bI0 + bI1) - sq;                         // (other lanes: a write-only pad slot)
        }
        BUSY_SYNC(1);
        // ---- P2 (role B) ----
#ifndef MPCMP_FOLDSIM
        BUSY_SYNC(2);
#endif
        // ---- P3 (role A1): this role is idle, so the constant operands of the E phase (Jacobian rows, row bounds) are
        // fetched now and only x~ remains to be read once P3 has produced it ----
        D2 p0[3], p1[3];
        MPCMP_SLEEP(MPCMP_SLP_P3);
        {
            const double *g0 = gkl + groff + (pq & 1) * GS, *g1 = gkl + groff + (1 - (pq & 1)) * GS;     // own row first
#pragma unroll
            for (int j = 0; j < 3; j++) { p0[j] = lds2(g0 + 2 * j); p1[j] = lds2(g1 + 2 * j); }
        }
        const double rr_ = pcl[2 * L::NA2], rri = pcl[4 * L::NA2], lgp = pcl[0], ugp = pcl[L::NA2];
        BUSY_SYNC(3);
        // ---- E: z~ = A x~, relaxation, projection, dual update ----
        if (isPath && ABL_ON(2)) {
            path_rows(p0, p1, xn, lds + L::oGp, [&](double zt) -> double {
                double w = 0.0;
                if (ownsRow) {
                    const double zr = alpha * zt + (1.0 - alpha) * zg;
                    const double zn = clip(zr + yg * rri, lgp, ugp);
                    yg += rr_ * (zr - zn);
                    zg = zn;
                    w = rr_ * zg - yg;
                }
                return w;
            });
        }
        BUSY_SYNC(4);
      }
      it += period;
      if (MPCMP_NO_TEST(cfg, period)) break;
      {
            __syncthreads();                             // (role B publishes x and the dynamics-row duals for the test)
            double mp[3] = {0, 0, 0};
            D2 p0[3], p1[3];                             // (the Jacobian rows again: nothing of the hot loop stays live for the test)
            if (isPath) {       // A x of the owned row
                const double *g0 = gkl + groff + (pq & 1) * GS, *g1 = gkl + groff + (1 - (pq & 1)) * GS;
#pragma unroll
                for (int j = 0; j < 3; j++) { p0[j] = lds2(g0 + 2 * j); p1[j] = lds2(g1 + 2 * j); }
                const double *xv = xx + xnoff;
                D2 x2[3];
#pragma unroll
                for (int j = 0; j < 3; j++) x2[j] = lds2(xv + 2 * j);
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    a0 += p0[j].x * x2[j].x; a1 += p1[j].x * x2[j].x;
                    a0 += p0[j].y * x2[j].y; a1 += p1[j].y * x2[j].y;
                }
                const double ax = quad_sum2(a0, a1);
                if (ownsRow) { mp[0] = fabs(ax - zg); mp[1] = fabs(ax); mp[2] = fabs(zg); }
            }
            if (qp2_primal_ok<NSEG>(cfg, mp, lds, tid, true)) {
                // the path-row part of A^T y (read by role B after the barriers of the sums' reduction) and the T column sum
                if (isPath) path_rows(p0, p1, xx, lds + L::oGpy, [&](double) -> double { return ownsRow ? yg : 0.0; });
                double sums[2] = {ownsRow ? pcl[3 * L::NA2] * yg : 0.0, 0.0};
                block_reduce_roles<2, false, L::NA1 / 64>(sums, lds + L::oRedS, tid, true);
                double md[3] = {0, 0, 0};
                done = qp2_dual_ok<NSEG>(cfg, md, lds, tid, true);
            }
      }
      if (done) break;
    }
    BUSY_DUMP;
    if (ownsRow) c.ws.y[(size_t)b * D::mn + myrow] = yg;
}

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
        // rhs slot: interior -> b_J, interface -> b_I; the controls of the last node couple to no interior block, so their
        // b_I entry IS r_I
        const int ia = ipos - nJ;
        vr.rpos = ipos < nJ ? L::oRhsJ + 56 * (ipos / 49) + ipos % 49
                            : ((ia >= 14 * (NSEG + 1) && ia < D::nI - 1) ? L::oRI + ia : L::oRhsI + ia);
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
__device__ __forceinline__ void qp2_role_b(const Qp2Ctx<NSEG> &c) {
    using D = Dim<NSEG>;
    using L = Qp2<NSEG>;
    constexpr int N = D::N, n = D::n, meq = D::meq, m = D::m, nI = D::nI, XS = L::XS;
    double *lds = c.lds;
    const mpcmp_config &cfg = *c.cfg;
    const int tid = c.tid, b = c.b, u = tid - L::NA1 - L::NA2;
    double *xn = lds + L::oXn, *xx = lds + L::oXx, *wg = lds + L::oWg, *ys = lds + L::oYs, *tpl = lds + L::oTp, *rI = lds + L::oRI;
    const bool isP2 = (u >> 3) < L::NPR;
    const int rp2 = u >> 3, part2 = u & 7;
    const bool isVar = u < n, isT = u == n - 1;
    const double sum_ha = lds[L::oMisc + 3];
    // register block: rows 2rp2, 2rp2+1 of S^-1 x columns part2*10..+9
#ifdef MPCMP_FOLDSIM
    constexpr int NS2 = 18;
    double s2[NS2], s2b[NS2];
    for (int j = 10; j < NS2; j++) { s2[j] = 1e-3 * (tid + j); s2b[j] = 1e-3 * (tid - j); }
#else
    constexpr int NS2 = 10;
    double s2[NS2], s2b[NS2];
#endif
    {
        const double *S = lds + L::oS;
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const int col = part2 * 10 + j;
            const int ro = 2 * rp2 + (part2 & 1), rx = 2 * rp2 + 1 - (part2 & 1);      // own row first (quad_sum2)
            s2[j] = (isP2 && ro < nI && col < nI) ? -S[packed(ro, col)] : 0.0;
            s2b[j] = (isP2 && rx < nI && col < nI) ? -S[packed(rx, col)] : 0.0;
        }
    }
    __syncthreads();          // S consumed; the staging area may now be overwritten
    // variable role: the per-iteration state (x, z_b, y_b) and the box stay in registers, the gather coefficients
    // and Hessian entries live in LDS (lane-transposed)
    double *vcl = lds + L::oVc + u;
    double v_rb;
    int v_rA, v_rB, v_rf, v_xpos, v_rpos;
    double cv[7];                                       // column of the dynamics rows: cf, dA[3], dB[3]
    {
        VarRole vr = make_var_role<NSEG>(cfg, c.ws, b, u, isVar, c.ts, c.tsT, c.rho_in, c.rho_eq);
        if (isT) { vr.hd = sum_ha + cfg.hess_reg; vr.ha = 0.0; }
        cv[0] = vr.cf;
#pragma unroll
        for (int i = 0; i < 3; i++) { cv[1 + i] = vr.dA[i]; cv[4 + i] = vr.dB[i]; }
#ifdef MPCMP_FOLDSIM
        for (int i = 0; i < 7; i++) vcl[i * L::NB] = cv[i];       // (what-if build: the dynamics-column coefficients live in LDS too)
#endif
        vcl[7 * L::NB] = vr.hd; vcl[8 * L::NB] = vr.ha; vcl[9 * L::NB] = vr.qv;
        vcl[10 * L::NB] = vr.lb; vcl[11 * L::NB] = vr.ub; vcl[12 * L::NB] = 1.0 / vr.rb;
        v_rb = vr.rb;
        v_rA = vr.rA; v_rB = vr.rB; v_rf = vr.rf; v_xpos = vr.xpos; v_rpos = vr.rpos;
    }
    __syncthreads();          // LDS-resident constants published
    // interface solve output: where x_I[row] goes
    int xdst = -1, myIrow = -1;
    if (isP2 && part2 < 2) {
        const int row = 2 * rp2 + part2;
        if (row < nI) {
            myIrow = row;
            if (row < 14 * (NSEG + 1)) {
                const int sb = row / 14, cc = row % 14;
                xdst = 3 * sb * XS + cc;
            } else if (row < nI - 1) {
                xdst = (N - 1) * XS + 14 + (row - 14 * (NSEG + 1));
            }
        }
    }
    double *xiw = (myIrow >= 0 && myIrow != nI - 1) ? xn + xdst : lds + L::oMisc + 7;
    // A^T w restricted to this variable's column: the path-row part is formed by role A2 (gp, node-major like x~), the
    // dynamics-row part is gathered here
    auto col_gather = [&](const double *w, const double *gp) -> double {
        double wv[7];
        double s = gp[v_xpos];
        wv[0] = w[v_rf];
#pragma unroll
        for (int i = 0; i < 3; i++) { wv[1 + i] = w[v_rA + 14 * i]; wv[4 + i] = w[v_rB + 14 * i]; }
#ifdef MPCMP_FOLDSIM
        double cq[7];
#pragma unroll
        for (int i = 0; i < 7; i++) cq[i] = vcl[i * L::NB];
#pragma unroll
        for (int i = 0; i < 7; i++) s += cq[i] * wv[i];
#else
#pragma unroll
        for (int i = 0; i < 7; i++) s += cv[i] * wv[i];
#endif
        return s;
    };
    // dynamics row owned by this lane (u < meq): ADMM state in registers, coefficients in the V area of LDS
    const double *rcl = lds + L::oRv + (u < meq ? u : 0) * L::RS;
    const bool isDyn = u < meq;
    const bool waveDyn = (u & ~63) < meq;              // wave-uniform: some lane of this wave owns a dynamics row
    static_assert((meq + 63) / 64 * 64 <= n - 1, "every lane of a wave with dynamics rows owns a variable other than T");
    double zgd = 0, ygd = 0, lgd = 0;
    double rcT = 0;                                    // the T coefficient -ts*f of the row (D_i0..D_i3 stay in LDS, -ts*T is uniform)
    int ix0 = 0, ixf = 0;
    if (isDyn) {
        const int r = u, k = r / 14, rr = r % 14, s = k / 3;
        ix0 = 3 * s * XS + rr;
        ixf = k * XS + ((rr < 7) ? 7 + rr : 14 + rr - 7);
        lgd = -c.ws.ceq[(size_t)b * meq + r];          // the row's bound l = u = -c_eq
        rcT = rcl[5];
    }
    const double mtsT = -c.tsT;
    auto row_dot_dyn = [&](const double (&rc)[4], const double *xe) -> double {
        return rc[0] * xe[ix0] + rc[1] * xe[ix0 + XS] + rc[2] * xe[ix0 + 2 * XS] + rc[3] * xe[ix0 + 3 * XS] +
               mtsT * xe[ixf] + rcT * xe[21];           // T is replicated at slot 21 of every node row
    };
    const double alpha = c.alpha, sigma = c.sigma, rho_eq = c.rho_eq;
    double x = 0, zb = 0, yb = 0;
    if (cfg.qp_warm_start) {     // warm duals: y_0 = lambda_k, x_0 = 0, z_0 = clip(0, l, u); w_0 of the dynamics rows
        const double *lamb = c.ws.lam + (size_t)b * D::mn;
        if (isVar) { yb = lamb[m + u]; zb = clip(0.0, vcl[10 * L::NB], vcl[11 * L::NB]); }
        if (isDyn) { ygd = lamb[u]; zgd = lgd; const double w = rho_eq * lgd - ygd; wg[u] = w; tpl[u] = rcT * w; }
        __syncthreads();
    }
    int it = 0, done = 0;
    BUSY_DECL;
    while (it < cfg.qp_iters) {
      const int period = MPCMP_PERIOD(cfg, it);
#pragma nounroll
      for (int kk = 0; kk < period; kk++) {
        // ---- A: rhs = sigma x - q + rho_b zb - yb + A^T w ----
        if (isVar && ABL_ON(3)) {
            const double sx = sigma * x, bz = v_rb * zb - yb;      // q is zero except for T (cost = T)
            if (isT) lds[v_rpos] = (sx - 1.0) + bz;      // b_T; its column sum and coupling terms are added by roles A1/A2
            else lds[v_rpos] = (sx + bz) + col_gather(wg, lds + L::oGp);
        }
        BUSY_SYNC(0);
        // ---- P1: (group A) ----
#ifndef MPCMP_FOLDSIM
        BUSY_SYNC(1);
#endif
        // ---- P2: x_I = S^-1 r_I  (2 rows x 10 columns per lane, 8-lane reduction; r_I was completed by role A2) ----
        if (isP2 && ABL_ON(7)) {
            double a0 = 0.0, a1 = 0.0;
            const double *rv = rI + part2 * 10;
#ifdef MPCMP_FOLDSIM
#pragma unroll
            for (int h = 0; h < 2; h++) {                  // (two operand batches: 5 + 4 pairs)
                const int j0 = h ? 5 : 0, j1 = h ? NS2 / 2 : 5;
                D2 r[5];
#pragma unroll
                for (int j = j0; j < j1; j++) r[j - j0] = lds2(rv + 2 * j);
#pragma unroll
                for (int j = j0; j < j1; j++) {
                    a0 += s2[2 * j] * r[j - j0].x; a1 += s2b[2 * j] * r[j - j0].x;
                    a0 += s2[2 * j + 1] * r[j - j0].y; a1 += s2b[2 * j + 1] * r[j - j0].y;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#else
            D2 r[NS2 / 2];
#pragma unroll
            for (int j = 0; j < NS2 / 2; j++) r[j] = lds2(rv + 2 * j);
#pragma unroll
            for (int j = 0; j < NS2 / 2; j++) {
                a0 += s2[2 * j] * r[j].x; a1 += s2b[2 * j] * r[j].x;
                a0 += s2[2 * j + 1] * r[j].y; a1 += s2b[2 * j + 1] * r[j].y;
            }
#endif
            double xi = quad_sum2(a0, a1);                 // even lanes: row 2 rp2, odd lanes: row 2 rp2 + 1
            xi += dpp_xor4(xi);
            *xiw = xi;                                     // (lanes without an output row, and the T row: the write-only pad slot)
            if (rp2 == (nI - 1) / 2 && (part2 & 1) == ((nI - 1) & 1)) {
                // x_T is replicated (slot 21 of every node row).  Here, at the end of this role's critical chain, only the copies role
                // A1 reads in P3 (nodes 3, 6, ..: one store per lane of the group); the other nodes' copies are made in P3, below
                const int nd = 3 * (1 + (part2 >> 1));
                if (nd < N) xn[nd * XS + 21] = xi;
            }
        }
        BUSY_SYNC(2);
        // ---- P3: (group A); this role is idle: the remaining copies of x_T, and the constant operands of the E phase ----
        if (u < N && (u % 3 != 0 || u == 0)) xn[u * XS + 21] = xn[3 * XS + 21];
        MPCMP_SLEEP(MPCMP_SLP_P3);
        const double rc[4] = {rcl[0], rcl[1], rcl[2], rcl[3]};
        const double vrbi = vcl[12 * L::NB], vlb = vcl[10 * L::NB], vub = vcl[11 * L::NB];
        BUSY_SYNC(3);
        // ---- E: variables and dynamics rows ----
        if (MPCMP_ABL == 1 || MPCMP_ABL == 10) {
        } else if (waveDyn) {
            // waves whose lanes own dynamics rows (all of them also own a variable): one straight-line block, so the two
            // independent update chains interleave; lanes past the last row compute on row 0's operands and store nothing
            const double zt = row_dot_dyn(rc, xn);
            const double xtv = xn[v_xpos];
            const double zr = alpha * zt + (1.0 - alpha) * zgd;
            ygd += rho_eq * (zr - lgd);                  // the row is an equality: the projection of anything onto [l, l] is l
            zgd = lgd;
            const double w = rho_eq * lgd - ygd;
            x = alpha * xtv + (1.0 - alpha) * x;
            const double zrv = alpha * xtv + (1.0 - alpha) * zb;
            const double znv = clip(zrv + yb * vrbi, vlb, vub);
            yb += v_rb * (zrv - znv);
            zb = znv;
            if (isDyn) {
                wg[u] = w;
                tpl[u] = rcT * w;
            }
        } else if (isVar) {
            const double xtv = xn[v_xpos];
            x = alpha * xtv + (1.0 - alpha) * x;
            const double zr = alpha * xtv + (1.0 - alpha) * zb;
            const double zn = clip(zr + yb * vrbi, vlb, vub);
            yb += v_rb * (zr - zn);
            zb = zn;
        }
        BUSY_SYNC(4);
      }
      it += period;
      if (MPCMP_NO_TEST(cfg, period)) break;
      {
            // the relaxed iterate x (node-major, T replicated) and the duals of the dynamics rows, for all roles' tests
            if (isDyn) ys[u] = ygd;
            if (isVar) {
                if (isT) { for (int k = 0; k < N; k++) xx[k * XS + 21] = x; }
                else xx[v_xpos] = x;
            }
            __syncthreads();
            const double rc[4] = {rcl[0], rcl[1], rcl[2], rcl[3]};
            double mp[3] = {0, 0, 0};
            if (isDyn) {
                const double ax = row_dot_dyn(rc, xx);
                mp[0] = fabs(ax - zgd); mp[1] = fabs(ax); mp[2] = fabs(zgd);
            }
            if (isVar) { mp[0] = fmax(mp[0], fabs(x - zb)); mp[1] = fmax(mp[1], fabs(x)); mp[2] = fmax(mp[2], fabs(zb)); }
            if (qp2_primal_ok<NSEG>(cfg, mp, lds, tid, true)) {
                double sums[2] = {isDyn ? rcT * ygd : 0.0, (isVar && !isT) ? vcl[8 * L::NB] * x : 0.0};
                block_reduce_roles<2, false, L::NA1 / 64>(sums, lds + L::oRedS, tid, true);
                double md[3] = {0, 0, 0};
                if (isVar) {
                    double hx, aty;
                    const double hdv = vcl[7 * L::NB];
                    if (isT) { hx = hdv * x + sums[1]; aty = sums[0] + yb; }
                    else { hx = hdv * x + vcl[8 * L::NB] * xx[21]; aty = col_gather(ys, lds + L::oGpy) + yb; }
                    md[0] = fabs(hx + aty + vcl[9 * L::NB]); md[1] = fabs(hx); md[2] = fabs(aty);
                }
                done = qp2_dual_ok<NSEG>(cfg, md, lds, tid, true);
            }
      }
      if (done) break;
    }
    BUSY_DUMP;
    if (isDyn) c.ws.y[(size_t)b * D::mn + u] = ygd;
    if (isVar) {
        c.ws.p[(size_t)b * n + u] = x;
        c.ws.y[(size_t)b * D::mn + m + u] = yb;
    }
}

// pass metadata of the assembly streams (structure.hpp: build_streams)
struct Qp2Streams {
    const uint32_t *words;
    int npass;
    int off[8], W[8];
    int split_dst, split_scr, split_n;
};

template <int NSEG>
__global__ __launch_bounds__(1024) void k_qp2(mpcmp_config cfg, WS ws, Qp2Streams st) {
    using D = Dim<NSEG>;
    using L = Qp2<NSEG>;
    constexpr int N = D::N, n = D::n, meq = D::meq, nJ = D::nJ, nI = D::nI, NT = L::NT;
    constexpr int GS = L::GS;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // logical wave of hardware wave w = nibble w of MPCMP_WPERM (hardware wave w runs on SIMD w % 4; the roles are wave ranges of
    // the LOGICAL thread index, so the permutation decides which roles share a SIMD's issue slots)
    const int tid = (MPCMP_WPERM == 0xFEDCBA9876543210ull) ? (int)threadIdx.x
                                                            : (int)((MPCMP_WPERM >> (4 * (threadIdx.x >> 6))) & 15ull) * 64 + (int)(threadIdx.x & 63);
    const int b = ws.perm[blockIdx.x];      // launch order: solver_kernels.hpp k_order
    if (MPCMP_RETIRED(ws, b)) return;       // receding horizon: an arrived instance is not re-solved
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
    const int u = tid - L::NA1 - L::NA2;
    const bool isVar = u >= 0 && u < n, isT = u == n - 1;

    for (int i = tid; i < L::oRed - L::oRhsJ; i += NT) lds[L::oRhsJ + i] = 0.0;     // exchanged vectors and their pads
    // V = [dynamics-row coefficients | path Jacobians]: operands of the assembly, reused by the ADMM rows
    for (int i = tid; i < N * 8 * GS; i += NT) gkl[i] = (i % GS < 22) ? Gkg[(i / GS) * 22 + i % GS] : 0.0;
    for (int i = tid; i < meq * L::RS; i += NT) {
        const int r = i / L::RS, a = i % L::RS, k = r / 14, rr = r % 14;
        double v = 0.0;
        if (a < 4) v = c_D[4 * (k % 3) + a];
        else if (a == 4) v = -tsT;
        else if (a == 5) v = -ts * zg_[(rr < 7) ? 14 * k + 7 + rr : 14 * N + 7 * k + rr - 7];
        lds[L::oRv + i] = v;
    }
    __syncthreads();
    // ---- variable role (role B): only the Hessian/rho entries needed by the assembly are kept live here ----
    double v_diag = 0.0, v_ha = 0.0;
    int ipos = 0;
    {
        VarRole vr = make_var_role<NSEG>(cfg, ws, b, u, isVar, ts, tsT, rho_in, rho_eq);
        double sv[1] = {isVar && !isT ? fabs(vr.ha) : 0.0};
        block_reduce16<1, false>(sv, red, tid);
        if (isT) { vr.hd = sv[0] + cfg.hess_reg; vr.ha = 0.0; }
        if (tid == 0) lds[L::oMisc + 3] = sv[0];
        v_diag = vr.hd + sigma + vr.rb; v_ha = vr.ha;
        if (isVar) ipos = int_of_ext(NSEG, u);
    }
    STAMP(0);
    // ---------------- assembly + factorisation ----------------
    double *F = lds + L::oS;                      // factor area: S | KJJ[NSEG] | KJC[NSEG] (two stream groups of HS) | scratch | rdv
    double *S = F, *KJJ = lds + L::oKJJ, *KJC = lds + L::oKJC, *rdv = lds + L::oRdv;
    const double *V = lds + L::oRv;
    // one assembly pass: every thread interprets its own (load-balanced) word stream
    auto run_pass = [&](int p, int dst_off) {
        const uint32_t *wp = st.words + st.off[p] + tid;
        const int W = st.W[p];
        double acc = 0.0;
        constexpr int WC = 16;                       // stream words fetched per round trip (coalesced, L2-resident)
        for (int w0 = 0; w0 < W; w0 += WC) {
            uint32_t xw[WC];
#pragma unroll
            for (int q = 0; q < WC; q++) xw[q] = (w0 + q < W) ? wp[(size_t)(w0 + q) * NT] : 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < WC; q++) {
                const uint32_t x = xw[q];
                if ((int)x >= 0) {
                    acc += ((x >> 28) & 1u ? rho_eq : rho_in) * V[x & 0x3fffu] * V[(x >> 14) & 0x3fffu];
                } else if (x != 0xFFFFFFFFu) {
                    F[(x & 0xFFFFFu) + dst_off] = acc;
                    acc = 0.0;
                }
            }
        }
    };
    auto tri_decode = [](int e, int &i, int &j) {
        i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= e) i++;
        while (i * (i + 1) / 2 > e) i--;
        j = e - i * (i + 1) / 2;
    };
    // Symmetric sweep of `nblk` symmetric nb x nb blocks on their first `npiv` pivots:
    //     a_ij -= a_ik a_jk / a_kk,   a_ik <- a_ik / a_kk,   a_kk <- -1 / a_kk      for k = 0 .. npiv-1,
    // i.e. [[A, B], [B^T, D]] -> [[-A^-1, A^-1 B], [(A^-1 B)^T, D - B^T A^-1 B]] (npiv = nb: the whole block becomes -(A^-1)).
    // A thread owns one 4x4 tile of the lower block triangle (diagonal tiles hold both triangles, the lower one is
    // authoritative) in registers for the whole sweep, so a step is 16 FMAs on four 16-byte LDS reads of the pivot column.
    // Per step only the pivot column and the pivot reciprocal travel through LDS (double buffered): ONE barrier per step
    // and no division outside the pivot owner. The step loop is unrolled over k mod 4, which makes every register index
    // of the pivot row/column handling static.  ld(blk,i,j) (i >= j) reads an entry; st(blk,i,j,v) stores an entry that has a
    // pivot index (j < npiv); st_trail(blk,i,j,v) receives the trailing (Schur) entries, one block at a time with a barrier
    // in between (the trailing blocks of different segments overlap in S).
    auto sweep = [&](int nb, int npiv, int nblk, int cst, auto &&ld, auto &&st_, auto &&st_trail) {
        constexpr int CB = L::CB;
        const int nt4 = (nb + 3) >> 2, ntile = nt4 * (nt4 + 1) / 2;
        const bool live = tid < ntile * nblk;
        int blk = 0, Ib = 0, Jb = 0;
        if (live) { blk = tid / ntile; tri_decode(tid % ntile, Ib, Jb); }
        const bool diag = live && Ib == Jb;
        double *cb0 = lds + L::oCol + blk * cst;
        double v[4][4];
#pragma unroll
        for (int a = 0; a < 4; a++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = 4 * Ib + a, j = 4 * Jb + q;
                v[a][q] = (live && i < nb && j < nb) ? (i >= j ? ld(blk, i, j) : ld(blk, j, i)) : 0.0;
            }
        }
        if (live && Jb == 0) {                       // column 0 (and its pivot reciprocal)
#pragma unroll
            for (int a = 0; a < 4; a++) cb0[4 * Ib + a] = v[a][0];
            if (Ib == 0) { if (!(v[0][0] > 0.0)) status |= 2; rdv[blk] = pivot_rcp(v[0][0]); }
        }
        __syncthreads();
        for (int kb = 0; 4 * kb < npiv; kb++) {
#pragma unroll
            for (int ka = 0; ka < 4; ka++) {
                const int k = 4 * kb + ka;
                if (k < npiv) {                      // workgroup-uniform
                    const int k1a = (ka + 1) & 3, k1b = kb + (ka == 3 ? 1 : 0);
                    if (live) {
                        const double *cur = cb0 + (k & 1) * CB;
                        double *nxt = cb0 + ((k + 1) & 1) * CB;
                        const D2 ci0 = lds2(cur + 4 * Ib), ci1 = lds2(cur + 4 * Ib + 2);
                        const D2 cj0 = lds2(cur + 4 * Jb), cj1 = lds2(cur + 4 * Jb + 2);
                        const double rd = rdv[(k & 1) * 4 + blk];
                        const double cI[4] = {ci0.x, ci0.y, ci1.x, ci1.y};
                        const double rJ[4] = {cj0.x * rd, cj0.y * rd, cj1.x * rd, cj1.y * rd};
#pragma unroll
                        for (int a = 0; a < 4; a++) {
#pragma unroll
                            for (int q = 0; q < 4; q++) v[a][q] = v[a][q] - cI[a] * rJ[q];
                        }
                        if (Ib == kb) {              // pivot row
#pragma unroll
                            for (int q = 0; q < 4; q++) v[ka][q] = rJ[q];
                        }
                        if (Jb == kb) {              // pivot column
#pragma unroll
                            for (int a = 0; a < 4; a++) v[a][ka] = cI[a] * rd;
                            if (Ib == kb) v[ka][ka] = -rd;
                        }
                        if (k + 1 < npiv) {          // publish column k+1 and its pivot reciprocal
                            if (Jb == k1b) {
#pragma unroll
                                for (int a = 0; a < 4; a++) nxt[4 * Ib + a] = (diag && a < k1a) ? v[k1a][a] : v[a][k1a];
                                if (diag) {
                                    const double pv = v[k1a][k1a];
                                    if (!(pv > 0.0)) status |= 2;
                                    rdv[((k + 1) & 1) * 4 + blk] = pivot_rcp(pv);
                                }
                            } else if (Ib == k1b) {
#pragma unroll
                                for (int q = 0; q < 4; q++) nxt[4 * Jb + q] = v[k1a][q];
                            }
                        }
                    }
                    __syncthreads();
                }
            }
        }
        if (live) {
#pragma unroll
            for (int a = 0; a < 4; a++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = 4 * Ib + a, j = 4 * Jb + q;
                    if (i < nb && j <= i && j < npiv) st_(blk, i, j, v[a][q]);
                }
            }
        }
        __syncthreads();
        if (npiv < nb) {
            for (int sblk = 0; sblk < nblk; sblk++) {
                if (live && blk == sblk) {
#pragma unroll
                    for (int a = 0; a < 4; a++) {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int i = 4 * Ib + a, j = 4 * Jb + q;
                            if (i < nb && j <= i && j >= npiv) st_trail(blk, i, j, v[a][q]);
                        }
                    }
                }
                __syncthreads();
            }
        }
    };
    // pass 0: interface block S and every interior diagonal block K_JJ,s
    run_pass(0, 0);
    __syncthreads();
    if (isT && st.split_dst >= 0) {               // the T-T entry was accumulated in chunks
        double acc = 0.0;
        for (int q = 0; q < st.split_n; q++) acc += F[st.split_scr + q];
        F[st.split_dst] = acc;
    }
    __syncthreads();
    if (isVar) {
        if (ipos >= nJ) {
            const int a = ipos - nJ;
            S[packed(a, a)] += v_diag;
            if (!isT) S[packed(nI - 1, a)] += v_ha;
        } else {
            KJJ[(ipos / 49) * D::JP + packed(ipos % 49, ipos % 49)] += v_diag;
        }
    }
    __syncthreads();
    // coupling blocks K_JC of every segment (the stream passes are grouped by HS segments: later groups land behind)
    for (int g = 0; g < st.npass - 1; g++) run_pass(1 + g, g * L::HS * D::JC);
    __syncthreads();
    if (isVar && ipos < nJ) KJC[(ipos / 49) * D::JC + (ipos % 49) * 29 + 28] += v_ha;
    __syncthreads();
    STAMP(9);
#ifndef MPCMP_QP2_AUGSWEEP
    // Interior blocks: K_JJ,s <- -G_s = -(K_JJ,s^-1) by one symmetric sweep per segment (49 pivots, all segments concurrently, 91 tiles each), then
    // the two block GEMMs of the QP on the MATRIX CORES (v_mfma_f64_16x16x4_f64; north star: "MFMA only for the one true dense block GEMM"):
    //     E_s = G_s K_JC,s        (49 x 49 by 49 x 29)        and        K_CJ,s E_s        (29 x 49 by 49 x 29, the segment's Schur complement term).
    // Two waves per segment, one 16-column tile of E_s each.  Lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15]; register r of the result holds
    // row (l >> 4) + 4 r, column l & 15.  So register r of row tile mt of E_s IS the B operand of k-step 4 mt + r of the second product, and a
    // fragment of K_JC is the B operand of the first product and the A operand (K_CJ = K_JC^T) of the second: no data movement between the GEMMs.
    // (Until round 4 one sweep of the augmented [[K_JJ, K_JC], [K_CJ, 0]] (78 x 78) did all three on the vector ALUs: -DMPCMP_QP2_AUGSWEEP.)
    sweep(49, 49, NSEG, 80,
          [&](int blk, int i, int j) -> double { return KJJ[blk * D::JP + packed(i, j)]; },
          [&](int blk, int i, int j, double val) { KJJ[blk * D::JP + packed(i, j)] = val; },
          [&](int, int, int, double) {});
    STAMP(11);
    {
        using V4 = __attribute__((ext_vector_type(4))) double;
        const int wv = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
        const int sw = wv >> 1, hf = wv & 1;
        const bool mm = wv < 2 * NSEG;
        const double *Gn = KJJ + (mm ? sw : 0) * D::JP;
        double *Kc = KJC + (mm ? sw : 0) * D::JC;
        V4 e[4], sacc[2];
#pragma unroll
        for (int mt = 0; mt < 4; mt++) e[mt] = V4{0.0, 0.0, 0.0, 0.0};
        sacc[0] = V4{0.0, 0.0, 0.0, 0.0}; sacc[1] = V4{0.0, 0.0, 0.0, 0.0};
        if (mm) {
            // fragment (ks, t) of K_JC: entry [4 ks + lk][16 t + li] (rows >= 49, columns >= 29: zero); loads with safe indices, then selects
            auto frag = [&](int ks, int t) -> double {
                const int row = 4 * ks + lk, col = 16 * t + li;
                const bool in = row < 49 && col < 29;
                const double v = Kc[(in ? row : 0) * 29 + (in ? col : 0)];
                return in ? v : 0.0;
            };
#pragma unroll
            for (int ks = 0; ks < 13; ks++) {
                const int k = 4 * ks + lk;
                const double bf = frag(ks, hf);
#pragma unroll
                for (int mt = 0; mt < 4; mt++) {
                    const int i = 16 * mt + li;
                    const bool in = i < 49 && k < 49;
                    const double g = -Gn[packed(in ? i : 0, in ? k : 0)];
                    e[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(in ? g : 0.0, bf, e[mt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int ks = 0; ks < 13; ks++) {
#pragma unroll
                for (int mt2 = 0; mt2 < 2; mt2++)
                    if (mt2 >= hf) sacc[mt2] = __builtin_amdgcn_mfma_f64_16x16x4f64(frag(ks, mt2), e[ks >> 2][ks & 3], sacc[mt2], 0, 0, 0);
            }
        }
        __syncthreads();          // every fragment of K_JC has been read: E_s takes its place
        if (mm) {
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int R = 16 * mt + lk + 4 * r, Cc = 16 * hf + li;
                    if (R < 49 && Cc < 29) Kc[R * 29 + Cc] = e[mt][r];
                }
            }
        }
        // S -= K_CJ,s E_s (lower triangle), one segment at a time: neighbouring segments share the block of their common interface node, all share T
        for (int sblk = 0; sblk < NSEG; sblk++) {
            if (mm && sw == sblk) {
#pragma unroll
                for (int mt2 = 0; mt2 < 2; mt2++) {
                    if (mt2 >= hf) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int R = 16 * mt2 + lk + 4 * r, Cc = 16 * hf + li;
                            if (R < 29 && Cc < 29 && R >= Cc) {
                                const int ia = R < 28 ? 14 * sw + R : nI - 1, ib = Cc < 28 ? 14 * sw + Cc : nI - 1;
                                S[packed(ia, ib)] -= sacc[mt2][r];
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
#else
    // One sweep per segment of the symmetric [[K_JJ, K_JC], [K_CJ, 0]] (78 x 78) on its 49 interior pivots, all segments
    // concurrently:  K_JJ <- -G_s,  K_JC <- E_s = G_s K_JC,  trailing block = -K_CJ E_s, which is the segment's Schur
    // complement contribution and is added to S.
    sweep(49 + 29, 49, NSEG, 80,
          [&](int blk, int i, int j) -> double {
              return i < 49 ? KJJ[blk * D::JP + packed(i, j)] : (j < 49 ? KJC[blk * D::JC + j * 29 + (i - 49)] : 0.0);
          },
          [&](int blk, int i, int j, double val) {
              if (i < 49) KJJ[blk * D::JP + packed(i, j)] = val;
              else KJC[blk * D::JC + j * 29 + (i - 49)] = val;
          },
          [&](int blk, int i, int j, double val) {
              const int ca = i - 49, cb = j - 49;
              const int ia = ca < 28 ? 14 * blk + ca : nI - 1, ib = cb < 28 ? 14 * blk + cb : nI - 1;
              S[packed(ia, ib)] += val;
          });
    STAMP(11);
#endif
    // -G_s and E_s stay where the sweep left them (K_JJ / K_JC areas): the role threads pick their register blocks up from
    // there before the ADMM view overlays the factor area
    STAMP(14);
    STAMP(1);
    sweep(nI, nI, 1, L::CB,
          [&](int, int i, int j) -> double { return S[packed(i, j)]; },
          [&](int, int i, int j, double val) { S[packed(i, j)] = val; },
          [&](int, int, int, double) {});                  // S <- -(S^-1)
    STAMP(2);
    {
        const int any = __syncthreads_or(status);
        if (tid == 0 && any) ws.status[b] |= any;
    }
#ifdef MPCMP_STAMPS
    if (tid == 0) { unsigned long long *dbg = ws.dbg + (size_t)b * MPCMP_DBG_WORDS; for (int k = 0; k < 3; k++) dbg[k] = stamp_acc[k]; for (int k = 9; k < 15; k++) if (k != 12) dbg[k] = stamp_acc[k]; }
#endif
    if (tid < L::NA1) qp2_role_a1<NSEG>(c);
    else if (tid < L::NA1 + L::NA2) qp2_role_a2<NSEG>(c);
    else qp2_role_b<NSEG>(c);
}

}  // namespace mpcmp
