// solver_kernels.hpp — gfx950 kernels of the batched min-time OCP solver. One workgroup = one OCP.
//
//   k_init : warm start (given, or built-in stand-in for warm_start_RK, motionPlanner.cpp:146-175),
//            lambda = 0, first linearisation
//   k_qp   : one QP of the SQP: assemble K = H + sigma I + rho_box + A^T rho A, nested-dissection
//            factorisation with explicit block inverses, box-ADMM iterations (the dominant kernel)
//   k_step : l1-merit line search (polympc_redef.hpp:73-121), primal/dual update, exact re-linearisation
//            (polympc_redef.hpp:133-147), final report
// Data layout: every per-problem array is contiguous per problem ([B][len]) so a workgroup's loads/stores
// are coalesced; everything touched inside the ADMM loop lives in VGPRs (matrix rows, per-row/per-variable
// ADMM state) or LDS (E, S^-1, exchanged vectors).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "rbd_device.hpp"
#include "structure.hpp"

namespace mpcmp {

// cubic Chebyshev-Gauss-Lobatto differentiation matrix on ascending nodes {-1,-1/2,1/2,1}
static __device__ __constant__ double c_D[16] = {-19.0 / 6, 4.0, -4.0 / 3, 0.5,  -1.0, 1.0 / 3, 1.0, -1.0 / 3,
                                          1.0 / 3, -1.0, -1.0 / 3, 1.0,  -0.5, 4.0 / 3, -4.0, 19.0 / 6};

struct WS {
    const mpcmp_model *model;
    const int *ext_of_int;
    const int *entry_ptr;
    const uint32_t *terms;
    const double *x0, *xf;    // [B][14]
    double *z;                // [B][n]
    double *lam;              // [B][m+n]
    double *ceq;              // [B][meq]
    double *g;                // [B][8N]
    double *Gk;               // [B][N][8][22]
    double *p;                // [B][n]
    double *y;                // [B][m+n]
    int *qpit;                // [B] iterations of the last QP
    int *perm;                // [B] launch order of the QP kernel: workgroup g solves problem perm[g] (longest expected first)
    int *okey;                // [B] ordering key: decayed maximum of the problem's previous QP iteration counts
    int *done;                // [1] workgroups of the running step launch that have finished (the last one computes the next QP order)
    int *qp_total;            // [B]
    int *status;              // [B]
    double *alpha;            // [B]
    unsigned long long *dbg;  // [B][MPCMP_DBG_WORDS] phase cycle stamps (diagnostic builds with -DMPCMP_STAMPS only)
    const int *retired;       // [B] or null: receding-horizon instances that have arrived (k_advance) are not re-solved: every solve kernel returns at once
    const double *alt_x, *alt_u, *alt_T;   // or null: the receding-horizon driver's start guess (jerk-limited trajectory from the current state) for an instance
                                           // that cannot be re-guessed from its previous solve (k_init: reguess && prev_bad); null = the built-in initialiser
};
// a retired instance of the receding-horizon loop: the whole workgroup leaves (uniform)
#define MPCMP_RETIRED(ws_, b_) ((ws_).retired != nullptr && (ws_).retired[b_] != 0)

// ws.status[b] while a solve runs: bits 0..7 = status bits of mpcmp_info (include/mpcmp.h), bits 8..30 = number of QPs that ran out of
// iterations (every QP kernel adds MPCMP_ST_CAP_ONE then); k_step / k_step_m fold it into the record
#define MPCMP_ST_CAP_ONE 0x100
// tools/isa_phases.py compiles with -DMPCMP_NOCHECK: the ADMM loops then contain no termination-test block, so that the hot path of every role is the
// straight-line text between its barriers (diagnostic build: the QPs never converge early)
#ifdef MPCMP_NOCHECK
#define MPCMP_CHECK_NOW(c) (false && (c))
#else
#define MPCMP_CHECK_NOW(c) (c)
#endif
// ADMM loops of k_qp2 / k_qp5: periods of check_every iterations in an inner loop WITHOUT test code, the termination test after it.
// iterations of the next period, and whether a period ends without a test (cut short by qp_iters: tests happen at multiples of check_every)
#define MPCMP_PERIOD(cfg, it) (((cfg).qp_iters - (it)) < (cfg).check_every ? ((cfg).qp_iters - (it)) : (cfg).check_every)
#ifdef MPCMP_NOCHECK
#define MPCMP_NO_TEST(cfg, period) (true || (period) < (cfg).check_every)
#else
#define MPCMP_NO_TEST(cfg, period) ((period) < (cfg).check_every)
#endif
#define MPCMP_DBG_WORDS 160   /* 16 workgroup stamps + [16 waves][8] per-wave busy cycles of k_qp2 + 16 stamps of k_step */
#ifdef MPCMP_STAMPS
#define STAMP(slot) do { if (tid == 0) { const unsigned long long now_ = clock64(); stamp_acc[slot] += now_ - stamp_t; stamp_t = now_; } } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

template <int NW, int K, bool MAX>
__device__ __forceinline__ void block_reduce(double (&v)[K], double *red, int tid) {
#pragma unroll
    for (int k = 0; k < K; k++) {
        double x = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double y = __shfl_xor(x, o);
            x = MAX ? fmax(x, y) : x + y;
        }
        v[k] = x;
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

__device__ __forceinline__ double viol(double v, double lo, double hi) {
    return v < lo ? lo - v : (v > hi ? v - hi : 0.0);
}
__device__ __forceinline__ double clip(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
// status word and capped-QP count of the result record (include/mpcmp.h: MPCMP_STATUS_*; the oracle's orc_solve_multi does the same)
__device__ __forceinline__ void report_status(const mpcmp_config &cfg, int st, int anybad, mpcmp_info &o) {
    int s = (st & 0xFF) | (anybad ? MPCMP_STATUS_NAN : 0);
    o.qp_capped = (st >> 8) & 0x7FFFFF;      // (23 bits: the count cannot wrap for any sqp_iters a solve can run)
    if (o.qp_capped) s |= MPCMP_STATUS_QP_CAPPED;
    if (o.defect_inf > cfg.eps_abs || o.path_viol_inf > cfg.eps_abs || o.term_err_inf > cfg.eps_target + cfg.eps_abs) s |= MPCMP_STATUS_OUTSIDE_TOL;
    if (!(o.T >= cfg.lbT - 1e-9 && o.T <= cfg.ubT + 1e-9)) s |= MPCMP_STATUS_T_OUT_OF_BOX;
    o.status = s;
}

// variable box of external variable v (motionPlanner.cpp:33,47,66-79)
template <int NSEG>
__device__ __forceinline__ void var_box(const mpcmp_config &c, const double *x0, const double *xf, int v, double &lo,
                                        double &hi) {
    using D = Dim<NSEG>;
    if (v < 14 * D::N) {
        const int k = v / 14, r = v % 14;
        if (k == 0) { lo = hi = x0[r]; }
        else if (k == D::N - 1) { lo = xf[r] - c.eps_target; hi = xf[r] + c.eps_target; }
        else { lo = c.lbx[r]; hi = c.ubx[r]; }
    } else if (v < 21 * D::N) {
        const int r = (v - 14 * D::N) % 7;
        lo = c.lbu[r]; hi = c.ubu[r];
    } else { lo = c.lbT; hi = c.ubT; }
}

template <int NSEG>
struct LinLds {
    static constexpr bool TW = (NSEG <= 4);     // tangent wrenches parked in LDS (does not fit next to the rest for N = 19)
    static constexpr int size = Dim<NSEG>::N * (14 + 147) + (TW ? 42 * Dim<NSEG>::NT : 0);
};

// ------------------------------------------------------------------------------------------------
// Linearisation of all nodes of one problem (block-wide). zl: iterate in LDS (external order).
// scr: LDS scratch of N*14 + N*147 + N*7 doubles. Writes g [8N], Gk [N][8][22], ceq [meq] (global).
// GMODEL: mdl points to global memory (WS::model), and the recursion reads the constants of a joint where it processes it (rnea_dir's RELOAD)
template <int NSEG, bool GMODEL = false>
__device__ __forceinline__ void linearise_block(const mpcmp_config &cfg, const mpcmp_model *__restrict__ mdl, const double *zl,
                                double *scr, double *g_out, double *Gk_out, double *ceq_out, int tid, unsigned long long *stamps = nullptr) {
    using D = Dim<NSEG>;
    constexpr int N = D::N;
#ifdef MPCMP_STAMPS
    unsigned long long lst_t = clock64();
#define LSTAMP(slot) do { if (stamps != nullptr && tid == 0) { const unsigned long long now_ = clock64(); stamps[slot] = now_ - lst_t; lst_t = now_; } } while (0)
#else
#define LSTAMP(slot) do { } while (0)
#endif
    double *sc = scr;               // [N][14]
    double *raw = scr + N * 14;     // [N][7][21]
    double *tw = raw + N * 147;     // [42][NT] tangent wrenches of rnea_dir (lane-transposed)
    for (int t = tid; t < N * 7; t += D::NT) {
        double s, c;
        sincos(zl[14 * (t / 7) + (t % 7)], &s, &c);
        sc[2 * t] = s; sc[2 * t + 1] = c;
    }
    __syncthreads();
    LSTAMP(0);
    // one pass (a plain `if`, not a loop: a loop makes the compiler hoist the ~180 model constants of the recursion
    // out of it as loop invariants and spill them).  Lane map: the 14 N directions in (q, v) first — the generic tangent recursion —, then, from
    // the next wave boundary on, the 7 N directions in a (mass-matrix columns: rnea_mcol, a quarter of the instructions), then the N tool rows.
    // With N = 13 the workgroup is five waves on four SIMDs; node-major order gave waves 0 and 4 — one SIMD — two generic streams (40 k cycles),
    // now that SIMD runs one generic and one short stream.
    // Five waves (N = 13): waves 0 and 4 share a SIMD, so the generic block starts at wave 1 and the short items take wave 0 and wave 4.
    constexpr int LG0 = (D::NW == 5 && 7 * N >= 64) ? 64 : 0, LH = 14 * N, LGE = LG0 + LH;          // generic items: lanes [LG0, LGE)
    constexpr int LC1 = ((LGE + 63) / 64 * 64 + 8 * N - LG0 <= D::NT) ? (LGE + 63) / 64 * 64 : LGE;      // short items: lanes [0, LG0) and [LC1, LF0)
    constexpr int LF0 = LC1 + 7 * N - LG0;
    static_assert(LF0 + N <= D::NT, "one (node, direction) pair per thread");
    if (tid < LGE || (tid >= LC1 && tid < LF0 + N)) {
        const int t = tid;
        const bool gen = t >= LG0 && t < LGE, tool = t >= LF0;
        const int ci = t < LG0 ? t : LG0 + (t - LC1);                  // index of a short item
        const int k = gen ? (t - LG0) / 14 : (tool ? t - LF0 : ci / 7);
        const int d = gen ? (t - LG0) % 14 : (tool ? 21 : 14 + ci % 7);
        const double *q_sc = sc + 14 * k;
        const double *v = zl + 14 * k + 7, *a = zl + 14 * N + 7 * k;
        if (d < 14) {
            double tau[7], dtau[7];
            rnea_dir<true, LinLds<NSEG>::TW, GMODEL>(mdl, q_sc, v, a, d / 7, d % 7, tau, dtau, tw + tid, D::NT);
#pragma unroll
            for (int i = 0; i < 7; i++) raw[(k * 7 + i) * 21 + d] = dtau[i];
            if (d == 0) {
#pragma unroll
                for (int i = 0; i < 7; i++) g_out[8 * k + i] = tau[i];
            }
        } else if (d < 21) {
            double dtau[7];
            rnea_mcol<LinLds<NSEG>::TW, GMODEL>(mdl, q_sc, d - 14, dtau, tw + tid, D::NT);
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
    LSTAMP(1);
    // rows 0..6 of every node: [dtau/dq | dtau/dqd | M symmetrised | quirk column]  (robot_ocp.hpp:129-142)
    // (the quirk column — a 14-term sum — has a pass of its own: inside the pass below one lane in 22 ran its loop and the other 21 waited, in every
    //  round of the pass: 9.5 k of k_step<4>'s 93 k cycles)
    for (int t = tid; t < N * 7 * 22; t += D::NT) {
        const int k = t / 154, i = (t % 154) / 22, c = t % 22;
        const double *rk = raw + k * 147;
        if (c == 21) continue;
        double val;
        if (c < 14) val = rk[i * 21 + c];
        else { const int j = c - 14; val = (i <= j) ? rk[i * 21 + 14 + j] : rk[j * 21 + 14 + i]; }
        Gk_out[(k * 8 + i) * 22 + c] = val;
    }
    for (int t = tid; t < N * 7; t += D::NT) {
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
    LSTAMP(2);
    if (ceq_out) {
        const double T = zl[D::n - 1], ts = 1.0 / (2.0 * NSEG);
        for (int r = tid; r < D::meq; r += D::NT) {
            const int k = r / 14, rr = r % 14, s = k / 3, i = k % 3;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) acc += c_D[4 * i + j] * zl[14 * (3 * s + j) + rr];
            const double f = (rr < 7) ? zl[14 * k + 7 + rr] : zl[14 * N + 7 * k + rr - 7];
            ceq_out[r] = acc - ts * T * f;
        }
    }
    __syncthreads();
    LSTAMP(3);
}


// ------------------------------------------------------------------------------------------------
// k_init
template <int NSEG>
__global__ __launch_bounds__(Dim<NSEG>::NT) void k_init(mpcmp_config cfg, WS ws, const double *warm_x,
                                                        const double *warm_u, const double *warm_T, int reguess) {
    // (the model is read from the context's device copy, WS::model, joint by joint: rnea_dir's RELOAD.  Until round 5 it was a by-value kernel
    //  argument whose ~180 constants the compiler loaded once, ahead of everything, and parked in VGPR lanes for lack of SGPRs)
    using D = Dim<NSEG>;
    constexpr int N = D::N, n = D::n;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *zl = lds, *scr = lds + n;
    const int tid = threadIdx.x, b = blockIdx.x;
    if (tid == 0) { ws.perm[b] = b; ws.okey[b] = 0; if (b == 0) *ws.done = 0; }     // no history yet: problems are solved in batch order
    if (MPCMP_RETIRED(ws, b)) return;      // arrived: state, solution and record stay as the last solve left them
    const double *x0 = ws.x0 + 14 * b, *xf = ws.xf + 14 * b;
    // the slot's previous solve (0 in a fresh context): after a hard failure or a final time outside its box neither its iterate (re-guess of the
    // receding-horizon loop) nor its multipliers (mpcmp_config.carry_multipliers) are a start: built-in initialiser, lambda_0 = 0
    const bool prev_bad = (ws.status[b] & (MPCMP_STATUS_NAN | MPCMP_STATUS_NOT_PD | MPCMP_STATUS_XCH_DEAD | MPCMP_STATUS_T_OUT_OF_BOX)) != 0;
    __syncthreads();                                      // (thread 0 resets the status word below)
    if (reguess && prev_bad) { warm_x = ws.alt_x; warm_u = ws.alt_u; warm_T = ws.alt_T; reguess = 0; }      // (null: the built-in initialiser below)
    if (warm_x) {
        for (int v = tid; v < n; v += D::NT) {
            double val;
            if (v < 14 * N) val = warm_x[(size_t)b * 14 * N + v];
            else if (v < 21 * N) val = warm_u[(size_t)b * 7 * N + (v - 14 * N)];
            else val = warm_T[b];
            // re-guess from the previous solution: "Fix initial and final point at correct place" (motionPlanner.cpp:199-207)
            if (reguess && v < 14) val = x0[v];
            if (reguess && v >= 14 * (N - 1) && v < 14 * N) val = xf[v - 14 * (N - 1)];
            zl[v] = val;
        }
    } else {
        // stand-in for Ruckig: per-joint quintic, zero boundary accelerations, common duration on a geometric grid
        double T = 0.05;
        double *coef = scr;   // [7][6]
        for (int it = 0; it < 200; it++) {
            if (tid < 7) {
                const int j = tid;
                const double q0 = x0[j], v0 = x0[7 + j], q1 = xf[j], v1 = xf[7 + j];
                const double h = q1 - q0, T2 = T * T, T3 = T2 * T;
                coef[6 * j + 0] = q0; coef[6 * j + 1] = v0; coef[6 * j + 2] = 0.0;
                coef[6 * j + 3] = (20.0 * h - (8.0 * v1 + 12.0 * v0) * T) / (2.0 * T3);
                coef[6 * j + 4] = (-30.0 * h + (14.0 * v1 + 16.0 * v0) * T) / (2.0 * T3 * T);
                coef[6 * j + 5] = (12.0 * h - 6.0 * (v1 + v0) * T) / (2.0 * T3 * T2);
            }
            __syncthreads();
            int bad = 0;
            for (int t = tid; t < 7 * 65; t += D::NT) {
                const int j = t / 65, s = t % 65;
                const double *c = coef + 6 * j;
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
        if (tid < 7) {
            const int j = tid;
            const double q0 = x0[j], v0 = x0[7 + j], q1 = xf[j], v1 = xf[7 + j];
            const double h = q1 - q0, T2 = T * T, T3 = T2 * T;
            coef[6 * j + 0] = q0; coef[6 * j + 1] = v0; coef[6 * j + 2] = 0.0;
            coef[6 * j + 3] = (20.0 * h - (8.0 * v1 + 12.0 * v0) * T) / (2.0 * T3);
            coef[6 * j + 4] = (-30.0 * h + (14.0 * v1 + 16.0 * v0) * T) / (2.0 * T3 * T);
            coef[6 * j + 5] = (12.0 * h - 6.0 * (v1 + v0) * T) / (2.0 * T3 * T2);
        }
        __syncthreads();
        const double ts[4] = {-1.0, -0.5, 0.5, 1.0};
        for (int t = tid; t < N * 7; t += D::NT) {
            const int k = t / 7, j = t % 7;
            const int s = (k == N - 1) ? NSEG - 1 : k / 3, jj = k - 3 * s;
            const double tau = (s + 0.5 * (ts[jj] + 1.0)) / NSEG;
            const double *c = coef + 6 * j;
            const double tt = tau * T;
            zl[14 * k + j] = c[0] + tt * (c[1] + tt * (c[2] + tt * (c[3] + tt * (c[4] + tt * c[5]))));
            zl[14 * k + 7 + j] = c[1] + tt * (2 * c[2] + tt * (3 * c[3] + tt * (4 * c[4] + tt * 5 * c[5])));
            zl[14 * N + 7 * k + j] = 2 * c[2] + tt * (6 * c[3] + tt * (12 * c[4] + tt * 20 * c[5]));
        }
        __syncthreads();
        if (tid < 14) { zl[tid] = x0[tid]; zl[14 * (N - 1) + tid] = xf[tid]; }
        if (tid == 0) zl[n - 1] = T;
    }
    __syncthreads();
    for (int v = tid; v < n; v += D::NT) ws.z[(size_t)b * n + v] = zl[v];
    if (!cfg.carry_multipliers || prev_bad) for (int i = tid; i < D::mn; i += D::NT) ws.lam[(size_t)b * D::mn + i] = 0.0;      // (carried: the slot's multipliers of the previous solve stay)
    if (tid == 0) { ws.qp_total[b] = 0; ws.status[b] = 0; ws.alpha[b] = 0.0; }
    linearise_block<NSEG, true>(cfg, ws.model, zl, scr, ws.g + (size_t)b * 8 * N, ws.Gk + (size_t)b * N * 176,
                          ws.ceq + (size_t)b * D::meq, tid);
}
template <int NSEG>
struct InitLds { static constexpr int size = Dim<NSEG>::n + LinLds<NSEG>::size; };

// ------------------------------------------------------------------------------------------------
// k_qp
template <int NSEG>
struct QpLds {
    using D = Dim<NSEG>;
    static constexpr bool Z_LDS = (NSEG <= 4);
    static constexpr int GS = 23;                      // padded row stride of the path Jacobians in LDS
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int oE = 0;                       // E_s = G_s K_JC,s        [NSEG][49][29]
    static constexpr int oS = oE + NSEG * D::JC;       // packed -(S^-1)
    static constexpr int oGk = oS + D::SP;             // path Jacobians [N][8][GS]
    static constexpr int oU = oGk + D::N * 8 * GS;
    // factorisation view of the union region
    static constexpr int fKJJ = oU, fKJC = fKJJ + D::JP, fZ = fKJC + D::JC;
    static constexpr int fEnd = fZ + (Z_LDS ? D::n : 0);
    // ADMM view
    static constexpr int aRhs = oU, aXt = aRhs + D::n, aXI = aXt + D::n, aRI = aXI + D::nI,
                         aPart = aRI + D::nI, aWg = aPart + NSEG * 29, aYs = aWg + D::m, aEnd = aYs + D::m;
    static constexpr int oRed = cmax(fEnd, aEnd);
    static constexpr int size = oRed + D::NW * 8;
};

template <int NSEG>
__global__ __launch_bounds__(Dim<NSEG>::NT) void k_qp(mpcmp_config cfg, WS ws) {
    // (workgroup -> problem through ws.perm: see k_order)
    using D = Dim<NSEG>;
    using L = QpLds<NSEG>;
    constexpr int N = D::N, n = D::n, meq = D::meq, m = D::m, nJ = D::nJ, nI = D::nI, NT = D::NT;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, b = ws.perm[blockIdx.x];
    if (MPCMP_RETIRED(ws, b)) return;
    double *red = lds + L::oRed;
    const double ts = 1.0 / (2.0 * NSEG);
    const double rho_in = cfg.rho, rho_eq = cfg.rho * cfg.rho_eq_scale, sigma = cfg.sigma, alpha = cfg.alpha;
    const double *zg_ = ws.z + (size_t)b * n;
    const double *Gkg = ws.Gk + (size_t)b * N * 176;
    const double T = zg_[n - 1];
    const double tsT = ts * T;
    int status = 0;
#ifdef MPCMP_STAMPS
    unsigned long long stamp_acc[16] = {0}, stamp_t = clock64();
#endif

    // ---------------- per-thread static roles ----------------
    const bool isVar = tid < n, isRow = tid < m;
    // variable role
    double lb = 0, ub = 0, rb = rho_in, hd = 0, ha = 0, qv = 0, cf = 0;
    double dA[3] = {0, 0, 0}, dB[3] = {0, 0, 0};
    int rA = 0, rB = 0, rf = 0, pb = meq, ipos = 0, gcol = 0;
    bool hasG = false;
    const bool isT = (tid == n - 1);
    if (isVar) {
        const int v = tid;
        ipos = int_of_ext(NSEG, v);
        double lo, hi;
        var_box<NSEG>(cfg, ws.x0 + 14 * b, ws.xf + 14 * b, v, lo, hi);
        rb = (hi - lo < 1e-4) ? rho_eq : rho_in;
        const double zv = zg_[v];
        lb = lo - zv; ub = hi - zv;
        if (v < 14 * N) {
            const int k = v / 14, c = v % 14;
            pb = meq + 8 * k;
            if (k % 3 != 0) {
                const int s = k / 3, j = k % 3;
                rA = 14 * 3 * s + c;
#pragma unroll
                for (int i = 0; i < 3; i++) dA[i] = c_D[4 * i + j];
            } else {
                if (k < N - 1) {
                    rA = 14 * k + c;
#pragma unroll
                    for (int i = 0; i < 3; i++) dA[i] = c_D[4 * i + 0];
                }
                if (k > 0) {
                    rB = 14 * (k - 3) + c;
#pragma unroll
                    for (int i = 0; i < 3; i++) dB[i] = c_D[4 * i + 3];
                }
            }
            if (c >= 7 && k <= N - 2) { rf = 14 * k + (c - 7); cf = -tsT; ha = -ts * ws.lam[(size_t)b * D::mn + rf]; }
            gcol = k * 8 * L::GS + c; hasG = true;
        } else if (v < 21 * N) {
            const int k = (v - 14 * N) / 7, c = (v - 14 * N) % 7;
            pb = meq + 8 * k;
            if (k <= N - 2) { rf = 14 * k + 7 + c; cf = -tsT; ha = -ts * ws.lam[(size_t)b * D::mn + rf]; }
            gcol = k * 8 * L::GS + 14 + c; hasG = true;
        } else {
            qv = 1.0;   // cost gradient e_T (robot_ocp.hpp:201-213)
        }
        hd = fabs(ha) + cfg.hess_reg;     // Gershgorin shift, polympc_redef.hpp:57-70
    }
    {
        double sv[1] = {isVar && !isT ? fabs(ha) : 0.0};
        block_reduce<D::NW, 1, false>(sv, red, tid);
        if (isT) { hd = sv[0] + cfg.hess_reg; ha = 0.0; }
    }
    // row role
    double rcoef[6], lg = 0, ug = 0, rr_ = rho_in, coefT = 0;
    int ix0 = 0, ixf = 0, bx = 0, bu = 0, grow_off = 0;
    const bool isDyn = tid < meq;
#pragma unroll
    for (int c = 0; c < 6; c++) rcoef[c] = 0.0;
    if (isRow) {
        const int r = tid;
        if (isDyn) {
            const int k = r / 14, rr = r % 14, s = k / 3, i = k % 3;
            ix0 = 14 * 3 * s + rr;
            ixf = (rr < 7) ? 14 * k + 7 + rr : 14 * N + 7 * k + rr - 7;
#pragma unroll
            for (int j = 0; j < 4; j++) rcoef[j] = c_D[4 * i + j];
            rcoef[4] = -tsT;
            rcoef[5] = -ts * zg_[ixf];
            coefT = rcoef[5];
            const double ce = ws.ceq[(size_t)b * meq + r];
            lg = ug = -ce;
            rr_ = rho_eq;
        } else {
            const int k = (r - meq) / 8, q = (r - meq) % 8;
            bx = 14 * k; bu = 14 * N + 7 * k;
            grow_off = (k * 8 + q) * L::GS;
            coefT = Gkg[(k * 8 + q) * 22 + 21];
            const double gv = ws.g[(size_t)b * 8 * N + 8 * k + q];
            lg = cfg.lbg[q] - gv; ug = cfg.ubg[q] - gv;
            rr_ = (ug - lg < 1e-4) ? rho_eq : rho_in;
        }
    }
    double *gkl = lds + L::oGk;
    auto row_dot = [&](const double *xe) -> double {
        double s;
        if (isDyn) {
            s = rcoef[0] * xe[ix0] + rcoef[1] * xe[ix0 + 14] + rcoef[2] * xe[ix0 + 28] + rcoef[3] * xe[ix0 + 42] +
                rcoef[4] * xe[ixf] + rcoef[5] * xe[n - 1];
        } else {
            const double *gr = gkl + grow_off;
            s = gr[21] * xe[n - 1];
#pragma unroll
            for (int c = 0; c < 14; c++) s += gr[c] * xe[bx + c];
#pragma unroll
            for (int c = 0; c < 7; c++) s += gr[14 + c] * xe[bu + c];
        }
        return s;
    };
    auto col_gather = [&](const double *w) -> double {
        double s = cf * w[rf];
#pragma unroll
        for (int i = 0; i < 3; i++) s += dA[i] * w[rA + 14 * i];
#pragma unroll
        for (int i = 0; i < 3; i++) s += dB[i] * w[rB + 14 * i];
        if (hasG) {
            const double *gc = gkl + gcol;
#pragma unroll
            for (int q = 0; q < 8; q++) s += gc[q * L::GS] * w[pb + q];
        }
        return s;
    };

    STAMP(0);
    // ---------------- assembly + factorisation ----------------
    double *E = lds + L::oE, *S = lds + L::oS;
    double *KJJ = lds + L::fKJJ, *KJC = lds + L::fKJC;
    for (int i = tid; i < N * 176; i += NT) gkl[(i / 22) * L::GS + (i % 22)] = Gkg[i];
    const double *zl = zg_;
    if (L::Z_LDS) {
        double *zz = lds + L::fZ;
        for (int v = tid; v < n; v += NT) zz[v] = zg_[v];
        zl = zz;
    }
    __syncthreads();
    auto term_val = [&](uint32_t t) -> double {
        const int r = t >> 16, a = (t >> 8) & 255, c = t & 255;
        double va, vb, rho;
        if (r < meq) {
            const int k = r / 14, rr = r % 14, i = k % 3;
            const int fc = (rr < 7) ? 14 * k + 7 + rr : 14 * N + 7 * k + rr - 7;
            const double cT = -ts * zl[fc];
            va = a < 4 ? c_D[4 * i + a] : (a == 4 ? -tsT : cT);
            vb = c < 4 ? c_D[4 * i + c] : (c == 4 ? -tsT : cT);
            rho = rho_eq;
        } else {
            const double *row = gkl + (r - meq) * L::GS;
            va = row[a]; vb = row[c];
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
    // symmetric sweep of a packed nb x nb SPD matrix in LDS: A <- -(A^-1)
    auto sweep = [&](double *A, int nb, int cnt) {
        for (int k = 0; k < nb; k++) {
            const double d = A[packed(k, k)];
            if (!(d > 0.0)) status |= 2;
            const double rd = 1.0 / d;
            for (int e = tid; e < cnt; e += NT) {
                int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
                while ((i + 1) * (i + 2) / 2 <= e) i++;
                while (i * (i + 1) / 2 > e) i--;
                const int j = e - i * (i + 1) / 2;
                if (i != k && j != k) A[e] -= A[packed(i, k)] * (A[packed(j, k)] * rd);
            }
            __syncthreads();
            if (tid < nb) {
                if (tid != k) A[packed(tid, k)] *= rd;
                else A[packed(k, k)] = -rd;
            }
            __syncthreads();
        }
    };
    // interface block
    assemble(NSEG * (D::JP + D::JC), D::SP, S);
    __syncthreads();
    if (isVar && ipos >= nJ) {
        const int a = ipos - nJ;
        S[packed(a, a)] += hd + sigma + rb;
        if (!isT) S[packed(nI - 1, a)] += ha;
    }
    __syncthreads();
    double grow[49];
#pragma unroll
    for (int j = 0; j < 49; j++) grow[j] = 0.0;
    const int my_s = tid / 49, my_i = tid % 49;       // G-row role (tid < nJ)
    for (int s = 0; s < NSEG; s++) {
        assemble(s * (D::JP + D::JC), D::JP, KJJ);
        assemble(s * (D::JP + D::JC) + D::JP, D::JC, KJC);
        __syncthreads();
        if (isVar && ipos < nJ && ipos / 49 == s) {
            const int li = ipos % 49;
            KJJ[packed(li, li)] += hd + sigma + rb;
            KJC[li * 29 + 28] += ha;
        }
        __syncthreads();
        sweep(KJJ, 49, D::JP);
        if (tid < nJ && my_s == s) {
#pragma unroll
            for (int j = 0; j < 49; j++) grow[j] = -KJJ[packed(my_i, j)];
            // E_s = G_s K_JC,s
            for (int c = 0; c < 29; c++) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < 49; j++) acc += grow[j] * KJC[j * 29 + c];
                E[(s * 49 + my_i) * 29 + c] = acc;
            }
        }
        __syncthreads();
        // S -= K_CJ,s E_s  on the lower triangle of the 29x29 coupled-interface block
        for (int e = tid; e < 29 * 30 / 2; e += NT) {
            int ca = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
            while ((ca + 1) * (ca + 2) / 2 <= e) ca++;
            while (ca * (ca + 1) / 2 > e) ca--;
            const int cb = e - ca * (ca + 1) / 2;
            double acc = 0.0;
            for (int i = 0; i < 49; i++) acc += KJC[i * 29 + ca] * E[(s * 49 + i) * 29 + cb];
            const int ia = ca < 28 ? 14 * s + ca : nI - 1, ib = cb < 28 ? 14 * s + cb : nI - 1;
            S[packed(ia, ib)] -= acc;
        }
        __syncthreads();
    }
    STAMP(1);
    sweep(S, nI, D::SP);
    STAMP(2);

    // ---------------- ADMM (OSQP form on [A; I], reduced KKT) ----------------
    double *rhs = lds + L::aRhs, *xt = lds + L::aXt, *xI = lds + L::aXI, *rI = lds + L::aRI, *part = lds + L::aPart,
           *wg = lds + L::aWg, *ys = lds + L::aYs;
    double x = 0, zb = 0, yb = 0, zg = 0, yg = 0;
    double tsum = 0.0;      // sum_r coefT_r * w_r  (T column of A^T w)
    const int ext_my = (tid < nJ) ? ws.ext_of_int[tid] : 0;                       // P3 role
    const int ext_if = ((tid >> 2) < nI) ? ws.ext_of_int[nJ + (tid >> 2)] : 0;    // P2 role
    if (isRow) wg[tid] = 0.0;
    if (cfg.qp_warm_start) {     // warm duals (mpcmp_config.qp_warm_start): y_0 = lambda_k, x_0 = 0, z_0 = clip(0, l, u); w_0 = rho z_0 - y_0
        const double *lamb = ws.lam + (size_t)b * D::mn;
        double tp0 = 0.0;
        if (isRow) { yg = lamb[tid]; zg = clip(0.0, lg, ug); const double w = rr_ * zg - yg; wg[tid] = w; tp0 = coefT * w; }
        if (isVar) { yb = lamb[m + tid]; zb = clip(0.0, lb, ub); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tp0 += __shfl_xor(tp0, o);
        if ((tid & 63) == 0) red[tid >> 6] = tp0;
        __syncthreads();
        if (isT) { double sT0 = 0.0; for (int w = 0; w < D::NW; w++) sT0 += red[w]; tsum = sT0; }
    }
    __syncthreads();
    int it = 0, done = 0;
    for (it = 1; it <= cfg.qp_iters; it++) {
        // A: rhs = sigma x - q + rho_b zb - yb + A^T w
        if (isVar) {
            double r = sigma * x - qv + (rb * zb - yb);
            r += isT ? tsum : col_gather(wg);
            rhs[ipos] = r;
        }
        __syncthreads();
        STAMP(3);
        // P1: t = G_s b_Js ; partial = E_s^T b_Js
        double tloc = 0.0;
        if (tid < nJ) {
            const double *bj = rhs + 49 * my_s;
            double t0 = 0.0, t1 = 0.0;
#pragma unroll
            for (int jc = 0; jc < 49; jc += 7) {
                // chunks of 7 keep the LDS reads in flight without hoisting all 49 (VGPR pressure)
                double bv[7];
#pragma unroll
                for (int j = 0; j < 7; j++) bv[j] = bj[jc + j];
#pragma unroll
                for (int j = 0; j < 7; j++) { if (j & 1) t1 += grow[jc + j] * bv[j]; else t0 += grow[jc + j] * bv[j]; }
                __builtin_amdgcn_sched_barrier(0);
            }
            tloc = t0 + t1;
        } else if (tid < nJ + 29 * NSEG) {
            const int s = (tid - nJ) / 29, c = (tid - nJ) % 29;
            const double *bj = rhs + 49 * s, *Es = E + s * D::JC + c;
            double acc = 0.0;
#pragma unroll 7
            for (int i = 0; i < 49; i++) acc += Es[i * 29] * bj[i];
            part[s * 29 + c] = acc;
        }
        __syncthreads();
        STAMP(4);
        // P2a: r_I = b_I - sum_s E_s^T b_Js
        if (tid < nI) {
            const int a = tid;
            double r = rhs[nJ + a];
            if (a == nI - 1) {
#pragma unroll
                for (int s = 0; s < NSEG; s++) r -= part[s * 29 + 28];
            } else if (a < 14 * (NSEG + 1)) {
                const int sb = a / 14, c = a % 14;
                if (sb < NSEG) r -= part[sb * 29 + c];
                if (sb > 0) r -= part[(sb - 1) * 29 + 14 + c];
            }
            rI[a] = r;
        }
        __syncthreads();
        STAMP(5);
        // P2b: x_I = S^-1 r_I   (S holds -(S^-1); 4 lanes per row)
        if ((tid >> 2) < nI) {
            const int a = tid >> 2, pt = tid & 3;
            double acc = 0.0;
            for (int j = pt; j < nI; j += 4) acc += S[packed(a, j)] * rI[j];
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            if (pt == 0) { xI[a] = -acc; xt[ext_if] = -acc; }
        }
        __syncthreads();
        STAMP(6);
        // P3: x_J = t - E_s x_C(s)
        if (tid < nJ) {
            const double *Er = E + (my_s * 49 + my_i) * 29, *xc = xI + 14 * my_s;
            double acc = tloc - Er[28] * xI[nI - 1];
#pragma unroll
            for (int c = 0; c < 28; c++) acc -= Er[c] * xc[c];
            xt[ext_my] = acc;
        }
        __syncthreads();
        STAMP(7);
        // E: z~ = A x~, relaxation, projection, dual update
        double tp = 0.0;
        if (isRow) {
            const double zt = row_dot(xt);
            const double zr = alpha * zt + (1.0 - alpha) * zg;
            const double zn = clip(zr + yg / rr_, lg, ug);
            yg += rr_ * (zr - zn);
            zg = zn;
            const double w = rr_ * zg - yg;
            wg[tid] = w;
            tp = coefT * w;
        }
        if (isVar) {
            const double xtv = xt[tid];
            x = alpha * xtv + (1.0 - alpha) * x;
            const double zr = alpha * xtv + (1.0 - alpha) * zb;
            const double zn = clip(zr + yb / rb, lb, ub);
            yb += rb * (zr - zn);
            zb = zn;
        }
        const bool check = (it % cfg.check_every == 0);
        if (!check) {
            double sv[1] = {tp};
            // wave partial sums -> red, combined by the T thread after the barrier
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sv[0] += __shfl_xor(sv[0], o);
            if ((tid & 63) == 0) red[tid >> 6] = sv[0];
            __syncthreads();
            if (isT) {
                double s = 0.0;
#pragma unroll
                for (int w = 0; w < D::NW; w++) s += red[w];
                tsum = s;
            }
            // (red is next written after at least one more barrier)
        } else {
            // termination test: r_prim = ||[A;I]x - z||inf, r_dual = ||Hx + q + [A;I]^T y||inf
            if (isVar) rhs[tid] = x;          // rhs is free here: use it as x in external order
            if (isRow) ys[tid] = yg;
            double sums[3] = {tp, isRow ? coefT * yg : 0.0, (isVar && !isT) ? ha * x : 0.0};
            block_reduce<D::NW, 3, false>(sums, red, tid);   // includes the barrier publishing rhs/ys
            if (isT) tsum = sums[0];
            double mx[6] = {0, 0, 0, 0, 0, 0};   // rp, |Ax|, |z|, rd, |Hx|, |A^T y|
            if (isRow) {
                const double ax = row_dot(rhs);
                mx[0] = fabs(ax - zg); mx[1] = fabs(ax); mx[2] = fabs(zg);
            }
            if (isVar) {
                mx[0] = fmax(mx[0], fabs(x - zb)); mx[1] = fmax(mx[1], fabs(x)); mx[2] = fmax(mx[2], fabs(zb));
                double hx, aty;
                if (isT) { hx = hd * x + sums[2]; aty = sums[1] + yb; }
                else { hx = hd * x + ha * rhs[n - 1]; aty = col_gather(ys) + yb; }
                mx[3] = fabs(hx + aty + qv); mx[4] = fabs(hx); mx[5] = fabs(aty);
            }
            block_reduce<D::NW, 6, true>(mx, red, tid);
            const double ep = cfg.eps_abs + cfg.eps_rel * fmax(mx[1], mx[2]);
            const double ed = cfg.eps_abs + cfg.eps_rel * fmax(fmax(mx[4], mx[5]), 1.0);
            if (mx[0] <= ep && mx[3] <= ed) { done = 1; }
        }
        STAMP(8);
        if (done) break;
    }
    const bool capped = it > cfg.qp_iters;             // ran out of iterations without meeting the termination test
    if (capped) it = cfg.qp_iters;
    // ---------------- results ----------------
    if (isVar) {
        ws.p[(size_t)b * n + tid] = x;
        ws.y[(size_t)b * D::mn + m + tid] = yb;
    }
    if (isRow) ws.y[(size_t)b * D::mn + tid] = yg;
#ifdef MPCMP_STAMPS
    if (tid == 0) { for (int k = 0; k < 16; k++) ws.dbg[(size_t)b * MPCMP_DBG_WORDS + k] = stamp_acc[k]; ws.dbg[(size_t)b * MPCMP_DBG_WORDS + 15] = it; }
#endif
    {
        int any = __syncthreads_or(status);
        if (tid == 0) {
            ws.qpit[b] = it;
            ws.qp_total[b] += it;
            if (any) ws.status[b] |= any;
            if (capped) ws.status[b] += MPCMP_ST_CAP_ONE;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_step: line search + update + re-linearisation (+ final report)
template <int NSEG>
struct StepLds {
    using D = Dim<NSEG>;
    static constexpr int oZ = 0, oP = oZ + D::n, oPv = oP + D::n, oScr = oPv + 9 * D::N * 1,
                         oRed = oScr + LinLds<NSEG>::size + 9 * D::N * 14, size = oRed + D::NW * 12;
};

// ------------------------------------------------------------------------------------------------
// Launch order of the next QP kernel.  A QP runs between 25 and qp_iters ADMM iterations, a 1024-problem batch is only four
// workgroups per CU, and workgroups are dispatched in index order: whatever starts last sets the tail of the launch.  The
// iteration counts of a problem's previous QPs predict the next one well (correlation 0.9+), so problems are ordered
// longest-first by counting sort.  A long QP that was predicted short and therefore starts last costs a whole QP of tail, so
// the key is a decayed maximum over the history (key <- max(count, 0.9 key)) rather than the last count alone: replaying the
// bench workload's counts (tools/sched_sim.py) gives 0.985 of the batch-order makespan for the last count, 0.944 for this
// key, 0.904 for a perfect oracle.  The order only changes which workgroup solves which problem, never a result.
__device__ __forceinline__ void order_body(int B, const int *qpit, int *okey, int *perm, int *hist /* [256] LDS */, int tid, int nt) {
    int *base = hist + 128;
    if (tid < 128) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < B; i += nt) {
        const int decayed = (okey[i] * 29) >> 5, cnt = qpit[i];
        const int key = cnt > decayed ? cnt : decayed;
        okey[i] = key;
        const int bucket = key >> 3;
        atomicAdd(&hist[127 - (bucket > 127 ? 127 : bucket)], 1);
    }
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int k = 0; k < 128; k++) { base[k] = acc; acc += hist[k]; } }
    __syncthreads();
    for (int i = tid; i < B; i += nt) {
        const int bucket = okey[i] >> 3;
        perm[atomicAdd(&base[127 - (bucket > 127 ? 127 : bucket)], 1)] = i;
    }
}
// (order_body runs in the last workgroup of k_step to finish: a separate one-workgroup launch would queue behind the other
//  stream's QP launch, whose workgroups fill whole CUs, and stall its own stream's chain for hundreds of microseconds)

// launch order of the next QP launch: computed by the workgroup of the step launch that finishes last (every workgroup passes here, also
// those of retired receding-horizon instances)
__device__ __forceinline__ void step_tail(const WS &ws, double *lds, int tid, int nt) {
    __syncthreads();
    int *flag = reinterpret_cast<int *>(lds);
    if (tid == 0) {
        __threadfence();
        flag[0] = atomicAdd(ws.done, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (flag[0]) {
        __threadfence();
        order_body((int)gridDim.x, ws.qpit, ws.okey, ws.perm, flag + 16, tid, nt);
        if (tid == 0) *ws.done = 0;
    }
}

template <int NSEG>
__global__ __launch_bounds__(Dim<NSEG>::NT) void k_step(mpcmp_config cfg, WS ws, int final_iter, int sqp_it,
                                                        double *sol_x, double *sol_u, double *sol_T, mpcmp_info *info) {
    using D = Dim<NSEG>;
    using L = StepLds<NSEG>;
    constexpr int N = D::N, n = D::n, meq = D::meq, NT = D::NT;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    if (MPCMP_RETIRED(ws, b)) { if (tid == 0) ws.qpit[b] = 0; step_tail(ws, lds, tid, NT); return; }      // (sorted last in the next order)
    double *zl = lds + L::oZ, *pl = lds + L::oP, *pv = lds + L::oPv, *scr = lds + L::oScr, *red = lds + L::oRed;
    const double *x0 = ws.x0 + 14 * b, *xf = ws.xf + 14 * b;
    double *lam = ws.lam + (size_t)b * D::mn;
    const double *y = ws.y + (size_t)b * D::mn;
    const double ts = 1.0 / (2.0 * NSEG);
#ifdef MPCMP_STAMPS
    unsigned long long kst_t = clock64();
#define KSTAMP(slot) do { if (tid == 0) { const unsigned long long now_ = clock64(); ws.dbg[(size_t)b * MPCMP_DBG_WORDS + 144 + (slot)] = now_ - kst_t; kst_t = now_; } } while (0)
#else
#define KSTAMP(slot) do { } while (0)
#endif
    for (int v = tid; v < n; v += NT) { zl[v] = ws.z[(size_t)b * n + v]; pl[v] = ws.p[(size_t)b * n + v]; }
    // mu = ||lambda||_inf (polympc_redef.hpp:86), l1 violation at the current iterate (:79)
    double r2[1] = {0.0};
    for (int i = tid; i < D::mn; i += NT) r2[0] = fmax(r2[0], fabs(lam[i]));
    block_reduce<D::NW, 1, true>(r2, red, tid);      // also publishes zl/pl
    const double mu = r2[0];
    const bool isVar = tid < n, isDyn = tid < meq;
    double lo = 0, hi = 0;
    if (isVar) var_box<NSEG>(cfg, x0, xf, tid, lo, hi);
    double c0[1] = {0.0};
    if (isDyn) c0[0] += fabs(ws.ceq[(size_t)b * meq + tid]);
    if (tid < 8 * N) c0[0] += viol(ws.g[(size_t)b * 8 * N + tid], cfg.lbg[tid % 8], cfg.ubg[tid % 8]);
    if (isVar) c0[0] += viol(zl[tid], lo, hi);
    block_reduce<D::NW, 1, false>(c0, red, tid);
    KSTAMP(0);
    const double constr = c0[0];
    const double Tcur = zl[n - 1], pT = pl[n - 1];
    const double phi = Tcur + mu * constr;           // :93  (cost = T)
    const double Dphi = pT - mu * constr;            // :94  (cost gradient = e_T)
    // trial points alpha_t = tau^t, t = 0..ls_iters-2 (at most 9 evaluated in one pass)
    const int ntr = cfg.ls_iters - 1 < 9 ? cfg.ls_iters - 1 : 9;
    double *sct = scr + LinLds<NSEG>::size;          // [9][N][14] sin/cos of trial configurations
    for (int t = tid; t < ntr * N * 7; t += NT) {
        const int tr = t / (N * 7), kj = t % (N * 7), k = kj / 7, j = kj % 7;
        double al = 1.0;
        for (int q = 0; q < tr; q++) al *= cfg.ls_tau;
        double s, c;
        sincos(zl[14 * k + j] + al * pl[14 * k + j], &s, &c);
        sct[2 * t] = s; sct[2 * t + 1] = c;
    }
    __syncthreads();
    KSTAMP(1);
    static_assert(9 * N <= NT, "one (trial, node) pair per thread");
    if (tid < ntr * N) {           // (an `if`, not a loop: see linearise_block)
        const int t = tid;
        const int tr = t / N, k = t % N;
        double al = 1.0;
        for (int q = 0; q < tr; q++) al *= cfg.ls_tau;
        double v[7], a[7], tau[7];
#pragma unroll
        for (int j = 0; j < 7; j++) {
            v[j] = zl[14 * k + 7 + j] + al * pl[14 * k + 7 + j];
            a[j] = zl[14 * N + 7 * k + j] + al * pl[14 * N + 7 * k + j];
        }
        rnea_dir<false, false, true>(ws.model, sct + 14 * t, v, a, 0, 0, tau, nullptr);
        V3 ptool;
        fk_tool(ws.model, sct + 14 * t, &ptool, nullptr, nullptr, nullptr);
        double s = viol(ptool.z, cfg.lbg[7], cfg.ubg[7]);
#pragma unroll
        for (int j = 0; j < 7; j++) s += viol(tau[j], cfg.lbg[j], cfg.ubg[j]);
        pv[t] = s;
    }
    __syncthreads();
    KSTAMP(2);
    double acc[9];
    double al = 1.0;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        acc[t] = 0.0;
        if (t > 0) al *= cfg.ls_tau;
        if (t < ntr) {
            if (isDyn) {
                const int r = tid, k = r / 14, rr = r % 14, s = k / 3, i = k % 3;
                double d = 0.0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int ix = 14 * (3 * s + j) + rr;
                    d += c_D[4 * i + j] * (zl[ix] + al * pl[ix]);
                }
                const int fc = (rr < 7) ? 14 * k + 7 + rr : 14 * N + 7 * k + rr - 7;
                d -= ts * (Tcur + al * pT) * (zl[fc] + al * pl[fc]);
                acc[t] += fabs(d);
            }
            if (tid < N) acc[t] += pv[t * N + tid];
            if (isVar) acc[t] += viol(zl[tid] + al * pl[tid], lo, hi);
        }
    }
    KSTAMP(3);
    block_reduce<D::NW, 9, false>(acc, red, tid);
    KSTAMP(4);
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
    if (isVar) { zl[tid] += alpha * pl[tid]; ws.z[(size_t)b * n + tid] = zl[tid]; }
    for (int i = tid; i < D::mn; i += NT) lam[i] += alpha * (y[i] - lam[i]);
    __syncthreads();
    double *gout = ws.g + (size_t)b * 8 * N, *ceqo = ws.ceq + (size_t)b * meq;
    KSTAMP(5);
#ifdef MPCMP_STAMPS
    linearise_block<NSEG, true>(cfg, ws.model, zl, scr, gout, ws.Gk + (size_t)b * N * 176, ceqo, tid, ws.dbg + (size_t)b * MPCMP_DBG_WORDS + 152);
#else
    linearise_block<NSEG, true>(cfg, ws.model, zl, scr, gout, ws.Gk + (size_t)b * N * 176, ceqo, tid);
#endif
    KSTAMP(6);
    if (tid == 0) ws.alpha[b] = alpha;
    if (final_iter) {
        __threadfence_block();
        double s1[1] = {0.0}, mxs[3] = {0, 0, 0};
        int bad = 0;
        if (isDyn) { const double c = ceqo[tid]; s1[0] += fabs(c); mxs[0] = fabs(c); }
        if (tid < 8 * N) { const double vv = viol(gout[tid], cfg.lbg[tid % 8], cfg.ubg[tid % 8]); s1[0] += vv; mxs[1] = vv; }
        if (isVar) { s1[0] += viol(zl[tid], lo, hi); if (!isfinite(zl[tid])) bad = 1; }
        if (tid < 14) mxs[2] = fabs(zl[14 * (N - 1) + tid] - xf[tid]);
        block_reduce<D::NW, 1, false>(s1, red, tid);
        block_reduce<D::NW, 3, true>(mxs, red, tid);
        const int anybad = __syncthreads_or(bad);
        for (int v = tid; v < n; v += NT) {
            if (v < 14 * N) sol_x[(size_t)b * 14 * N + v] = zl[v];
            else if (v < 21 * N) sol_u[(size_t)b * 7 * N + v - 14 * N] = zl[v];
            else sol_T[b] = zl[v];
        }
        if (tid == 0) {
            mpcmp_info o;
            o.T = zl[n - 1]; o.viol_l1 = s1[0]; o.defect_inf = mxs[0]; o.path_viol_inf = mxs[1]; o.term_err_inf = mxs[2];
            o.last_alpha = alpha; o.qp_iters_total = ws.qp_total[b]; o.sqp_iters = sqp_it + 1;
            report_status(cfg, ws.status[b], anybad, o);
            if (info) info[b] = o;
            // the hard bits of the FINAL iterate (NaN, T outside its box) exist only here: written back, so that the slot's next k_init (no re-guess
            // from it, no carried multipliers) and k_advance (state held) see them like the bits the QP kernels set during the solve
            ws.status[b] |= o.status & (MPCMP_STATUS_NAN | MPCMP_STATUS_T_OUT_OF_BOX);
        }
    }
    KSTAMP(7);
    step_tail(ws, lds, tid, NT);
}

#ifndef MPCMP_V3_TU     /* (the leaf kernels below are not templates: they belong to the main translation unit only) */
// ------------------------------------------------------------------------------------------------
// leaf kernels
__global__ __launch_bounds__(64) void k_rnea_batch(const mpcmp_model *mdl, int n, const double *q, const double *v, const double *a,
                             double *tau) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double sc[14], vv[7], aa[7], t[7];
#pragma unroll
    for (int j = 0; j < 7; j++) { sincos(q[7 * i + j], &sc[2 * j], &sc[2 * j + 1]); vv[j] = v[7 * i + j]; aa[j] = a[7 * i + j]; }
    rnea_dir<false>(mdl, sc, vv, aa, 0, 0, t, nullptr);
#pragma unroll
    for (int j = 0; j < 7; j++) tau[7 * i + j] = t[j];
}

// evalConstraints AD overload for a list of (x,u): processed in chunks of N "nodes" per workgroup
template <int NSEG>
__global__ __launch_bounds__(Dim<NSEG>::NT) void k_eval_constraints(mpcmp_config cfg, const mpcmp_model *mdl, int total,
                                                                    const double *x, const double *u, double *g,
                                                                    double *G) {
    using D = Dim<NSEG>;
    constexpr int N = D::N;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *zl = lds, *scr = lds + D::n;
    const int tid = threadIdx.x, base = blockIdx.x * N;
    for (int t = tid; t < N * 21; t += D::NT) {
        const int k = t / 21, c = t % 21, src = (base + k < total) ? base + k : total - 1;
        if (c < 14) zl[14 * k + c] = x[(size_t)14 * src + c];
        else zl[14 * N + 7 * k + c - 14] = u[(size_t)7 * src + c - 14];
    }
    if (tid == 0) zl[D::n - 1] = 1.0;
    __syncthreads();
    double *gl = scr + LinLds<NSEG>::size, *Gl = gl + 8 * N;
    // write into LDS staging first (tail chunk may be partial), then copy the valid part out
    linearise_block<NSEG, true>(cfg, mdl, zl, scr, gl, Gl, nullptr, tid);
    const int valid = (total - base < N) ? total - base : N;
    for (int t = tid; t < valid * 8; t += D::NT) g[(size_t)base * 8 + t] = gl[t];
    for (int t = tid; t < valid * 176; t += D::NT) G[(size_t)base * 176 + t] = Gl[t];
}
template <int NSEG>
struct EvalLds { static constexpr int size = Dim<NSEG>::n + LinLds<NSEG>::size + Dim<NSEG>::N * (8 + 176); };

// get_MPC_trajectory<n_pts> (motionPlanner.hpp:99-116): Lagrange interpolation on the segment + RNEA
__global__ __launch_bounds__(128) void k_sample(const mpcmp_model *mdl, int nseg, int B, int n_pts, const double *sx, const double *su,
                         const double *sT, double *out) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * (n_pts + 1);
    if (gid >= total) return;
    const int b = (int)(gid / (n_pts + 1)), ip = (int)(gid % (n_pts + 1));
    const int N = 3 * nseg + 1;
    const double xi[4] = {-1.0, -0.5, 0.5, 1.0};
    const double t = (double)ip / n_pts;
    int s = (int)floor(t * nseg);
    if (s >= nseg) s = nseg - 1;
    if (s < 0) s = 0;
    const double xx = 2.0 * (t * nseg - s) - 1.0;
    double Lg[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double v = 1.0;
#pragma unroll
        for (int k = 0; k < 4; k++) if (k != j) v *= (xx - xi[k]) / (xi[j] - xi[k]);
        Lg[j] = v;
    }
    const double *X = sx + (size_t)b * 14 * N, *U = su + (size_t)b * 7 * N;
    double q[7], v[7], a[7], tau[7], sc[14];
#pragma unroll
    for (int r = 0; r < 7; r++) {
        q[r] = v[r] = a[r] = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            q[r] += Lg[j] * X[14 * (3 * s + j) + r];
            v[r] += Lg[j] * X[14 * (3 * s + j) + 7 + r];
            a[r] += Lg[j] * U[7 * (3 * s + j) + r];
        }
        sincos(q[r], &sc[2 * r], &sc[2 * r + 1]);
    }
    rnea_dir<false>(mdl, sc, v, a, 0, 0, tau, nullptr);
    double *o = out + (size_t)gid * 29;
    o[0] = t * sT[b];
#pragma unroll
    for (int r = 0; r < 7; r++) { o[1 + r] = q[r]; o[8 + r] = v[r]; o[15 + r] = a[r]; o[22 + r] = tau[r]; }
}


// MotionPlanner::get_MPC_point (motionPlanner.hpp:118-128) for one physical time per problem, including the reference's clamp
// (time >= T: the normalised time becomes T, not 1), plus the RNEA torque.  out [B][28] = q(7), v(7), a(7), tau(7).
__global__ __launch_bounds__(64) void k_mpc_point(const mpcmp_model *mdl, int nseg, int B, const double *sx, const double *su, const double *sT,
                                                  const double *time, double *out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int N = 3 * nseg + 1;
    const double T = sT[b];
    const double t = (time[b] < T) ? time[b] / T : T;
    const double xi[4] = {-1.0, -0.5, 0.5, 1.0};
    int s = (int)floor(t * nseg);
    if (s >= nseg) s = nseg - 1;
    if (s < 0) s = 0;
    const double xx = 2.0 * (t * nseg - s) - 1.0;
    double Lg[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double w = 1.0;
#pragma unroll
        for (int k = 0; k < 4; k++) if (k != j) w *= (xx - xi[k]) / (xi[j] - xi[k]);
        Lg[j] = w;
    }
    const double *X = sx + (size_t)b * 14 * N, *U = su + (size_t)b * 7 * N;
    double q[7], v[7], a[7], tau[7], sc[14];
#pragma unroll
    for (int r = 0; r < 7; r++) {
        q[r] = v[r] = a[r] = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            q[r] += Lg[j] * X[14 * (3 * s + j) + r];
            v[r] += Lg[j] * X[14 * (3 * s + j) + 7 + r];
            a[r] += Lg[j] * U[7 * (3 * s + j) + r];
        }
        sincos(q[r], &sc[2 * r], &sc[2 * r + 1]);
    }
    rnea_dir<false>(mdl, sc, v, a, 0, 0, tau, nullptr);
    double *o = out + (size_t)b * 28;
#pragma unroll
    for (int r = 0; r < 7; r++) { o[r] = q[r]; o[7 + r] = v[r]; o[14 + r] = a[r]; o[21 + r] = tau[r]; }
}

// Receding horizon, one thread per instance: x0 <- MPC solution evaluated at physical time dt (MotionPlanner::get_MPC_point,
// motionPlanner.hpp:118-128), and ARRIVAL: the reference's loop has no end (the caller decides), and an OCP whose start lies in its
// terminal box degenerates (T -> lbT = 0).  Rule, the same in the oracle (orc_rh_advance):
//   * retired already                          -> nothing;
//   * the solve failed hard or left the box of T -> the state is held (there is no trajectory to follow; the next k_init starts afresh);
//   * T <= dt: the plan ends within this control period -> the instance follows it to its end (last node) and is retired;
//   * else the state advances by dt (for dt < T the clamp of get_MPC_point never acts); if the new state lies inside the terminal box
//     |x - x_target| <= eps_target, the instance is retired as well.
// A retired instance keeps its state, its last solution and its record (status |= MPCMP_STATUS_ARRIVED) and is not re-solved (MPCMP_RETIRED).
// count[0] += instances solved in this step (not retired at its start), count[1] += instances retired by this step.
__global__ __launch_bounds__(256) void k_advance(int nseg, int nx, int B, double dt, double eps_target, const double *sx, const double *sT,
                                                 const int *status, const double *xf, double *x0, int *retired, mpcmp_info *info,
                                                 unsigned long long *count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (retired[b]) return;
    atomicAdd(&count[0], 1ull);
    if (status[b] & (MPCMP_STATUS_NAN | MPCMP_STATUS_NOT_PD | MPCMP_STATUS_XCH_DEAD | MPCMP_STATUS_T_OUT_OF_BOX)) return;
    const int N = 3 * nseg + 1;
    const double T = sT[b];
    const double *X = sx + (size_t)b * N * nx;
    double *xo = x0 + (size_t)b * nx;
    bool arrive = false;
    if (T <= dt) {
        for (int r = 0; r < nx; r++) xo[r] = X[(size_t)(N - 1) * nx + r];
        arrive = true;
    } else {
        const double t = dt / T;
        const double xi[4] = {-1.0, -0.5, 0.5, 1.0};
        int s = (int)floor(t * nseg);
        if (s >= nseg) s = nseg - 1;
        if (s < 0) s = 0;
        const double xx = 2.0 * (t * nseg - s) - 1.0;
        double w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double v = 1.0;
#pragma unroll
            for (int k = 0; k < 4; k++) if (k != j) v *= (xx - xi[k]) / (xi[j] - xi[k]);
            w[j] = v;
        }
        double far = 0.0;
        for (int r = 0; r < nx; r++) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) acc += w[j] * X[(size_t)(3 * s + j) * nx + r];
            xo[r] = acc;
            far = fmax(far, fabs(acc - xf[(size_t)b * nx + r]));
        }
        arrive = far <= eps_target;
    }
    if (arrive) {
        retired[b] = 1;
        if (info) info[b].status |= MPCMP_STATUS_ARRIVED;
        atomicAdd(&count[1], 1ull);
    }
}


// Per-trajectory checks of examples/benchmark.cpp:58-160 in one pass: resample at n_pts+1 uniform times (interpolation + RNEA),
// min / max of the 28 channels, terminal error, and the four pass flags (jerk, linear / angular task velocity, table
// collision).  One workgroup per trajectory; out [B][74] = mins(28) | maxs(28) | x(T) - x_target (14) | flags (4, 1 = pass).
__global__ __launch_bounds__(256) void k_traj_stats(const mpcmp_model *mdl, int nseg, int n_pts, const double *sx, const double *su,
                                                    const double *sT, const double *xf, double max_lin, double max_ang,
                                                    const double *jerk10 /*[7] = 10 x max_jerk*/, double *out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *smp = lds;                          // [n_pts+1][28]
    __shared__ int fl[4];
    const int b = blockIdx.x, tid = threadIdx.x, N = 3 * nseg + 1;
    const double *X = sx + (size_t)b * 14 * N, *U = su + (size_t)b * 7 * N;
    const double T = sT[b];
    if (tid < 4) fl[tid] = 1;
    __syncthreads();
    const double xi[4] = {-1.0, -0.5, 0.5, 1.0};
    for (int ip = tid; ip <= n_pts; ip += blockDim.x) {
        const double t = (double)ip / n_pts;
        int s = (int)floor(t * nseg);
        if (s >= nseg) s = nseg - 1;
        if (s < 0) s = 0;
        const double xx = 2.0 * (t * nseg - s) - 1.0;
        double Lg[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double v = 1.0;
#pragma unroll
            for (int k = 0; k < 4; k++) if (k != j) v *= (xx - xi[k]) / (xi[j] - xi[k]);
            Lg[j] = v;
        }
        double q[7], v[7], a[7], tau[7], sc[14];
#pragma unroll
        for (int r = 0; r < 7; r++) {
            q[r] = v[r] = a[r] = 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                q[r] += Lg[j] * X[14 * (3 * s + j) + r];
                v[r] += Lg[j] * X[14 * (3 * s + j) + 7 + r];
                a[r] += Lg[j] * U[7 * (3 * s + j) + r];
            }
            sincos(q[r], &sc[2 * r], &sc[2 * r + 1]);
        }
        rnea_dir<false>(mdl, sc, v, a, 0, 0, tau, nullptr);
        V3 pt, vl, va;
        fk_task_velocity(mdl, sc, v, &pt, &vl, &va);
        if (sqrt(vl.x * vl.x + vl.y * vl.y + vl.z * vl.z) > max_lin) fl[1] = 0;     // benchmark.cpp:139-142
        if (sqrt(va.x * va.x + va.y * va.y + va.z * va.z) > max_ang) fl[2] = 0;     // :143-146
        if (pt.z < 0.0) fl[3] = 0;                                                  // :149-155
        double *o = smp + (size_t)ip * 28;
#pragma unroll
        for (int r = 0; r < 7; r++) { o[r] = q[r]; o[7 + r] = v[r]; o[14 + r] = a[r]; o[21 + r] = tau[r]; }
    }
    __syncthreads();
    const double dT = T / n_pts;                                                     // time(nPoints)/nPoints, :121
    for (int ip = 1 + tid; ip <= n_pts; ip += blockDim.x)
        for (int r = 0; r < 7; r++)
            if (fabs((smp[(size_t)ip * 28 + 14 + r] - smp[(size_t)(ip - 1) * 28 + 14 + r]) / dT) > jerk10[r]) fl[0] = 0;   // :126-133
    double *o = out + (size_t)b * 74;
    if (tid < 28) {
        double mn = smp[tid], mx = smp[tid];
        for (int ip = 1; ip <= n_pts; ip++) { const double v = smp[(size_t)ip * 28 + tid]; mn = fmin(mn, v); mx = fmax(mx, v); }
        o[tid] = mn; o[28 + tid] = mx;
    }
    if (tid < 14) o[56 + tid] = smp[(size_t)n_pts * 28 + tid] - xf[(size_t)b * 14 + tid];   // :176-179
    __syncthreads();
    if (tid < 4) o[70 + tid] = (double)fl[tid];
}

#endif  /* MPCMP_V3_TU */

}  // namespace mpcmp
