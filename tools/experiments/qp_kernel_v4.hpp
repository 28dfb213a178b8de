// qp_kernel_v4.hpp — k_qp4<4>: the ADMM loop of the headline QP (N = 13) sized so that TWO OCPs are resident per CU.
//
// k_qp2 (qp_kernel_v2.hpp) gives one OCP a whole CU (1024 threads, 160 KB LDS, 128 VGPRs): its waves wait 60 % of their cycles at
// the five phase barriers and nothing else can use the CU meanwhile.  k_qp4 is the same ADMM iteration on 384 threads (6 waves),
// <= 80 KB of LDS and <= 168 VGPRs per lane, so that the hardware keeps two workgroups per CU (3 waves per SIMD, two problems): one
// problem's barrier waits are the other's issue slots.  Arithmetic = k_qp3's E-free form (structure3.hpp; factorisation by
// k_qp3f<4, 1, 4>):  T bordered out;  t = G b_J;  r_I = b_I - K_CJ t;  y_I = S^-1 r_I;  x_J = G (b_J - K_JC y_I);  x~ = x - wbar x~_T.
// Where the factor lives (per problem: 32 k doubles of registers, 10 k doubles of LDS):
//     G_s (4 x 49 x 49)  rows 0..47 in registers: lane 4 Q + part keeps rows 2 lp, 2 lp + 1 x columns 14 part .. + 13 (96 quads = all 384
//                        lanes); row 48 of every segment in LDS (4 x 56)
//     S^-1 (70 x 70)     LDS, row-major (39 KB): read once per iteration by 280 lanes, 2 rows x 10 columns each
//     path Jacobians     registers of the path-row lanes (16 lanes per node: 2 rows x 6 columns each); node 12: LDS
//     K_JC               LDS: row form (canonical slots) for K_JC y_I, column form incl. the dense 7 x 14 blocks for K_CJ t
// Roles (wave ranges with their own copy of the loop, as in k_qp2, so that a role's registers hold only what it uses):
//     A  waves 0..2   G quads of segments 0, 1; path rows of nodes 0..11; shares of P1b, P3, P4a
//     B  waves 3..5   G quads of segments 2, 3; rows 48; u_{N-1} block; variables; dynamics rows; path rows of node 12; x~_T; shares
// Seven phases / barriers per iteration, each a short straight-line chain:
//     A   rhs = sigma x - q + rho z - y + A^T w          (lane = variable)
//     P1a t = G b_J                                      (G quads)
//     P1b r_I = b_I - K_CJ t;  x~_T of the bordered solve (4 lanes per interface entry; wave 5 sums the T column and wbar^T rhs)
//     P3  y_I = S^-1 r_I, x~_I                           (8 lanes per row pair)
//     P4a c = b_J - K_JC y_I                             (lane = interior row; 4 lanes per row of the dense blocks)
//     P4b x~_J = G c - wbar x~_T                         (G quads)
//     E   z~ = A x~, relaxation, projection, duals, w = rho z - y, path-row part of A^T w
// DESIGN.md section 4 has the measurements.
#pragma once
#include "qp_kernel_v3.hpp"

namespace mpcmp {

template <int NSEG>
struct Qp4 {
    static_assert(NSEG == 4, "k_qp4 is laid out for NUM_SEG = 4 (N = 13)");
    using D = Dim3<NSEG>;
    using F = Qp4Fac<NSEG>;
    static constexpr int NT = 384, NA = 192, NB = NT - NA;
    static constexpr int XS = 24;                                   // node stride of x~, gp: [x_k (14) | u_k (7) | T | 0 0]
    static constexpr int SRS = F::SRS, KX = F::KX;
    static constexpr int TD = 48;                                   // segment stride of the stride-7 copy of t[u_3s]
    static constexpr int NI2 = 72;
    // ---- LDS (doubles) ----
    static constexpr int oS = 0;                                    // [nI][SRS]  S^-1
    static constexpr int oXn = oS + D::nI * SRS;                    // [N][XS] (+8) x~, node-major; phases A..P1b: wbar_v rhs_v per variable
    static constexpr int oGp = oXn + 320;                           // [N][XS] (+8) path-row part of A^T w; [k][21]: T column
    static constexpr int oWg = oGp + 320;                           // [meq + 2]  w = rho z - y of the dynamics rows
    static constexpr int oTp = oWg + D::meq + 2;                    // [meq + 8]  coefT_r w_r of the dynamics rows
    // --- from here to oEndV: dead between E and A, overlaid by the vectors of the termination test ---
    static constexpr int oRhsJ = oTp + D::meq + 8;                  // [NSEG][56]
    static constexpr int oRhsU = oRhsJ + NSEG * 56;                 // [8]
    static constexpr int oRhsI = oRhsU + 8;                         // [NI2]
    static constexpr int oTJ = oRhsI + NI2 + 8;                     // [NSEG][56] t = G b_J  (8 pad slots in front: row index -7..-1 of segment 0)
    static constexpr int oTU = oTJ + NSEG * 56;                     // [8]
    static constexpr int oCJ = oTU + 8;                             // [NSEG][56] c = b_J - K_JC y_I, then [8] c_U;  P1a..P1b: [NSEG + 1][TD] t[u_3s], stride 7
    static constexpr int oRI = oCJ + 240;                           // [NI2] r_I
    static constexpr int oYI = oRI + NI2 + 8;                       // [NI2] y_I (8 pad slots in front)
    static constexpr int oEndV = oYI + NI2;
    static constexpr int oKJC = oEndV;                              // [KJN] row form
    static constexpr int oKX = oKJC + Qp3<NSEG>::KJN;               // [NSEG + 1][7][KX] column form
    static constexpr int oGu = oKX + (NSEG + 1) * 7 * KX;           // [7][8]
    static constexpr int oJ12 = oGu + 56;                           // [8][XS] path Jacobian of node N - 1
    static constexpr int oG48 = oJ12 + 8 * XS;                      // [NSEG][56] row 48 of G_s
    static constexpr int oZero = oG48 + NSEG * 56;                  // [48] zeros
    static constexpr int oCD = oZero + 48;                          // [16] differentiation matrix, [16] zeros
    static constexpr int oMisc = oCD + 32;                          // [32]
    static constexpr int oRed = oMisc + 32;                         // [6 waves][8] workgroup reductions
    static constexpr int oP12 = oRed + 48;                          // [5][8] path rows of node N - 1: l, u, coefT, rho, 1 / rho
    static constexpr int oWb = oP12 + 40;                           // [N][XS] border vector wbar = K_0^-1 k, node-major like x~
    static constexpr int size = oWb + D::N * XS;
    static constexpr int dWb = oWb - oXn;                           // wbar of an x~ slot sits this far behind it
    static_assert(size * 8 + 512 <= 80 * 1024, "two workgroups per CU: 80 KB of LDS each");
    // termination-test overlay
    static constexpr int oXx = oRhsJ;                               // [N][XS] x node-major
    static constexpr int oGpy = oXx + D::N * XS;                    // [N][XS] path-row part of A^T y
    static constexpr int oYs = oGpy + D::N * XS;                    // [meq + 2] duals of the dynamics rows
    static_assert(oYs + D::meq + 2 <= oEndV, "termination-test vectors must fit over the solve vectors");
    static_assert(oS % 2 == 0 && oXn % 2 == 0 && oGp % 2 == 0 && oRhsJ % 2 == 0 && oTJ % 2 == 0 && oCJ % 2 == 0 && oRI % 2 == 0 && oYI % 2 == 0 &&
                  oKJC % 2 == 0 && oKX % 2 == 0 && oJ12 % 2 == 0 && oG48 % 2 == 0 && oZero % 2 == 0 && oMisc % 2 == 0 && oGpy % 2 == 0, "16-byte LDS accesses");
    // lane-constant table (host-built, qp4_build_lanes): NF 32-bit words per lane, [field][NT]
    static constexpr int F_GTW = 0, F_GBX = 1, F_P1KT = 2, F_RIBS = 3, F_P3SR = 4, F_P3YX = 5, F_P4KY = 6, F_P4DD = 7, F_VRX = 8, F_VPK = 9,
                         F_DPK = 10, F_WGW = 11, F_VV = 12, NF = 13;
    // misc slots
    static constexpr int M_xtT = 0, M_xT = 1, M_zbT = 2, M_ybT = 3, M_baseT = 4, M_delta = 5, M_hdT = 6, M_rbT = 7, M_lbT = 8, M_ubT = 9,
                         M_sumha = 10, M_kap = 11, M_rbiT = 12, M_pad = 16 /* 16..31: write-only pad slots */;
};

// two LDS indices in one register (every index of k_qp4 is below 2^14)
MPCMP_HD inline unsigned pk2(int a, int b) { return (unsigned)a | (unsigned)b << 16; }

// Lane constants of k_qp4: every LDS address a lane uses in the ADMM loop, as packed words [field][lane] (identical for every problem;
// built once per context on the host).  The kernel reloads them at the top of every termination-test period, so that they are defined
// right in front of the hot loop and dead during the test (the register allocator otherwise spills them around the test and the set-up
// code and reloads them from scratch inside the loop).
template <int NSEG>
inline void qp4_build_lanes(const Qp3Pat &pat, const int *ext_of_int, uint32_t *out) {
    using D = Dim3<NSEG>;
    using L = Qp4<NSEG>;
    constexpr int N = D::N, na = D::na, meq = D::meq, nI = D::nI, XS = L::XS, NT = L::NT;
    auto node_slot = [&](int v) -> int { return v < 14 * N ? XS * (v / 14) + v % 14 : XS * ((v - 14 * N) / 7) + 14 + (v - 14 * N) % 7; };
    for (int tid = 0; tid < NT; tid++) {
        auto F = [&](int f) -> uint32_t & { return out[f * NT + tid]; };
        const int o_pad = L::oMisc + L::M_pad + (tid & 15);
        // G quad: rows 2 lp, 2 lp + 1 of segment gseg
        const int Q = tid >> 2, part = tid & 3, gseg = Q / 24, glp = Q % 24, grow = 2 * glp + part;
        const bool gout = part < 2;
        F(L::F_GTW) = pk2(gout ? L::oTJ + 56 * gseg + grow : o_pad, (gout && grow < 7) ? L::oCJ + L::TD * gseg + 7 * grow : o_pad);
        F(L::F_GBX) = pk2(L::oRhsJ + 56 * gseg + 14 * part, gout ? L::oXn + node_slot(ext_of_int[49 * gseg + grow]) : o_pad);
        // P1b: lane 4 i + pp of entry i: pp = 0 opening segment, 1 closing segment, 2 dense block.  Role A: entries 0..47, role B: 48..69.
        {
            const int h1 = tid < L::NA ? tid : tid - L::NA + 4 * 48;
            int o_kb = L::oKX, o_tb = L::oZero, o_rI = o_pad;
            if (h1 < 4 * nI && (tid < L::NA || tid - L::NA < 4 * (nI - 48))) {
                const int i = h1 >> 2, pp = h1 & 3, nd = i / 14, c = i % 14;
                if (pp == 0 && nd < NSEG) { o_kb = L::oKX + nd * 7 * L::KX + c; o_tb = L::oTJ + 56 * nd + c - 7; }
                if (pp == 1 && nd >= 1) { o_kb = L::oKX + (nd - 1) * 7 * L::KX + 14 + c; o_tb = L::oTJ + 56 * (nd - 1) + c - 7; }
                if (pp == 2) { o_kb = L::oKX + nd * 7 * L::KX + 28 + c; o_tb = L::oCJ + L::TD * nd; }
                if (pp == 0) o_rI = L::oRI + i;
            }
            F(L::F_P1KT) = pk2(o_kb, o_tb);
            F(L::F_RIBS) = (uint32_t)o_rI;
        }
        // P3: 8 lanes (7 used) per row pair of S^-1.  Role B: row pairs 0..23, role A: 24..34.
        {
            const int h3 = tid < L::NA ? tid + 8 * 24 : tid - L::NA;
            const int sg = h3 >> 3, scs = h3 & 7;
            const bool isS = (tid >= L::NA || tid < 8 * (nI / 2 - 24)) && scs < 7;
            const int o_so = L::oS + (isS ? (2 * sg + (scs & 1)) * L::SRS + 10 * scs : L::SRS);      // own row; the other row: +- one row stride by lane parity
            const int o_sr = isS ? L::oRI + 10 * scs : L::oZero;
            const bool sOut = isS && scs < 2;
            const int s_row = sOut ? 2 * sg + scs : 0;
            F(L::F_P3SR) = pk2(o_so, o_sr);
            F(L::F_P3YX) = pk2(sOut ? L::oYI + s_row : o_pad, sOut ? L::oXn + 3 * (s_row / 14) * XS + s_row % 14 : o_pad);
        }
        // P4a: role B: the 35 rows with a dense block (4 lanes each: u_3s rows of the four segments, then u_{N-1}), then 52 further
        // interior rows; role A: the other 116 interior rows.  Interior rows 7..48 of segment s4 in the order sr = 42 s4 + (lr - 7).
        {
            constexpr int dCW = L::oCJ - L::oRhsJ;
            int o_kr = L::oKJC + NSEG * 196, o_yc = L::oYI, o_bs = o_pad - dCW, o_dk = L::oZero, o_dy = L::oZero;
            int s4 = -1, lr = 0, dj = -1;
            if (tid >= L::NA) {
                const int bt = tid - L::NA;
                if (bt < 140) { const int dr = bt >> 2; s4 = dr / 7; lr = dr % 7; dj = bt & 3; }
                else { const int sr = bt - 140; s4 = sr / 42; lr = 7 + sr % 42; }
            } else if (tid < 168 - 52) { const int sr = tid + 52; s4 = sr / 42; lr = 7 + sr % 42; }
            if (s4 >= 0) {
                if (dj >= 0) { o_dk = L::oKX + (s4 * 7 + lr) * L::KX + 28 + 4 * dj; o_dy = L::oYI + 14 * s4 + 4 * dj; }
                if (dj <= 0) {
                    if (s4 < NSEG) {
                        o_kr = L::oKJC + (49 * s4 + lr) * 4;
                        o_yc = L::oYI + 14 * s4 + (int)(pat.jc[lr] & 255u);      // canonical slots [base, base + 14, base - 7, base + 7] (structure3.hpp)
                        o_bs = L::oRhsJ + 56 * s4 + lr;
                    } else o_bs = L::oRhsU + lr;
                }
            }
            F(L::F_P4KY) = pk2(o_kr, o_yc);
            F(L::F_P4DD) = pk2(o_dk, o_dy);
            F(L::F_RIBS) |= (uint32_t)o_bs << 16;
        }
        // one variable per lane: role B lane bt owns bt, role A lane t < na - 192 owns 192 + t (external arm order)
        {
            const int vv = tid >= L::NA ? tid - L::NA : (tid < na - L::NB ? L::NB + tid : -1);
            int o_rhs = o_pad, o_xpos = 0, wA = 0, wB = 0, wf = meq, cdA = 16, cdB3 = 0;
            if (vv >= 0) {
                const int v = vv, ip = int3_of_ext(NSEG, v);
                o_rhs = ip < D::nJ ? L::oRhsJ + 56 * (ip / 49) + ip % 49 : (ip < D::nJ + 7 ? L::oRhsU + (ip - D::nJ) : L::oRhsI + (ip - D::nJ - 7));
                if (v < 14 * N) {
                    const int k = v / 14, c = v % 14;
                    o_xpos = k * XS + c;
                    if (k % 3 != 0) { wA = 14 * 3 * (k / 3) + c; cdA = k % 3; }
                    else {
                        if (k < N - 1) { wA = 14 * k + c; cdA = 0; }
                        if (k > 0) { wB = 14 * (k - 3) + c; cdB3 = 1; }
                    }
                    if (c >= 7 && k <= N - 2) wf = 14 * k + (c - 7);
                } else {
                    const int k = (v - 14 * N) / 7, c = (v - 14 * N) % 7;
                    o_xpos = k * XS + 14 + c;
                    if (k <= N - 2) wf = 14 * k + 7 + c;
                }
            }
            F(L::F_VRX) = pk2(o_rhs, o_xpos);
            F(L::F_VPK) = (unsigned)wA | (unsigned)wB << 8 | (unsigned)wf << 16 | (unsigned)cdA << 24 | (unsigned)cdB3 << 29;
            F(L::F_VV) = (uint32_t)vv;
        }
        // dynamics row bt (role B lanes bt < meq)
        {
            int o_dx0 = 0, o_dxf = 0, o_dci = 16, o_wgw = L::oWg + meq + 1;
            if (tid >= L::NA && tid - L::NA < meq) {
                const int r = tid - L::NA, k = r / 14, rr = r % 14;
                o_dx0 = 3 * (k / 3) * XS + rr;
                o_dxf = k * XS + (rr < 7 ? 7 + rr : 14 + rr - 7);
                o_dci = 4 * (k % 3);
                o_wgw = L::oWg + r;
            }
            F(L::F_DPK) = (unsigned)o_dx0 | (unsigned)o_dxf << 9 | (unsigned)o_dci << 18;
            F(L::F_WGW) = (uint32_t)o_wgw;
        }
    }
}

#ifndef QP4_CHECK_ON
#define QP4_CHECK_ON 1
#endif
// diagnostic builds (-DMPCMP_STAMPS, tools/stamps4.py): cycles per phase of the ADMM loop as seen by lane 0 of waves 0 and 3 (QS4: after a
// barrier), the busy part of each phase (QB4: in front of the barrier), and where / when the workgroup ran
#ifdef MPCMP_STAMPS
#define QS4(k) do { const unsigned long long n_ = clock64(); st_acc[k] += n_ - st_t; st_t = n_; } while (0)
#define QB4(k) do { st_busy[k] += clock64() - st_t; } while (0)
#else
#define QS4(k) do { } while (0)
#define QB4(k) do { } while (0)
#endif
#define QP4_LO(w_) ((int)((w_) & 0xFFFFu))
#define QP4_HI(w_) ((int)((w_) >> 16))

template <int NSEG>
__global__ __launch_bounds__(384, 3) void k_qp4(mpcmp_config cfg, WS ws, const uint32_t *__restrict__ lanes, int B, const double *__restrict__ fac) {
    using D = Dim3<NSEG>;
    using L = Qp4<NSEG>;
    using F = Qp4Fac<NSEG>;
    constexpr int N = D::N, na = D::na, meq = D::meq, ma = D::ma, nI = D::nI, XS = L::XS, NT = L::NT;
    constexpr int n_tot = na + 1, mn_tot = ma + na + 1;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= B) return;
    const int b = ws.perm[blockIdx.x];
    const double ts = 1.0 / (2.0 * NSEG);
    const double rho_in = cfg.rho, rho_eq = cfg.rho * cfg.rho_eq_scale, sigma = cfg.sigma, alpha = cfg.alpha;
    const double *zg_ = ws.z + (size_t)b * n_tot;
    const double T = zg_[na], tsT = ts * T;
    const double *Gkg = ws.Gk + (size_t)b * N * 176;
    const double *lam_rows = ws.lam + (size_t)b * mn_tot;
    const double *x0e = ws.x0 + (size_t)b * 14, *xfe = ws.xf + (size_t)b * 14;
    const double *fa = fac + (size_t)b * F::FAC;
    double *misc = lds + L::oMisc, *xn = lds + L::oXn, *gpl = lds + L::oGp, *wg = lds + L::oWg, *tpl = lds + L::oTp, *bpl = lds + L::oXn;
    double *red = lds + L::oRed;
    // ---------------- LDS images ----------------
    for (int i = tid; i < nI * L::SRS; i += NT) lds[L::oS + i] = fa[F::fS + i];
    for (int i = tid; i < L::oEndV - L::oXn; i += NT) lds[L::oXn + i] = 0.0;
    for (int i = tid; i < Qp3<NSEG>::KJN; i += NT) lds[L::oKJC + i] = fa[F::fKJC + i];
    for (int i = tid; i < (NSEG + 1) * 7 * L::KX; i += NT) lds[L::oKX + i] = fa[F::fKX + i];
    if (tid < 56) lds[L::oGu + tid] = fa[F::fGu + tid];
    for (int i = tid; i < NSEG * 56; i += NT) lds[L::oG48 + i] = fa[F::fG48 + i];
    for (int i = tid; i < 8 * XS; i += NT) { const int r = i / XS, c = i % XS; lds[L::oJ12 + i] = c < 22 ? Gkg[((N - 1) * 8 + r) * 22 + c] : 0.0; }
    for (int i = tid; i < N * XS; i += NT) lds[L::oWb + i] = 0.0;
    if (tid < 48) lds[L::oZero + tid] = 0.0;
    if (tid < 32) lds[L::oCD + tid] = tid < 16 ? c_D[tid] : 0.0;
    if (tid < 32) {
        double v = 0.0;
        if (tid == L::M_baseT) v = -1.0;                             // sigma x_T - q_T + rho_T z_T - y_T with x = z = y = 0, q_T = 1 (cost = T)
        if (tid == L::M_lbT) v = cfg.lbT - T;
        if (tid == L::M_ubT) v = cfg.ubT - T;
        if (tid == L::M_rbT) v = (cfg.ubT - cfg.lbT < 1e-4) ? rho_eq : rho_in;
        if (tid == L::M_rbiT) v = (cfg.ubT - cfg.lbT < 1e-4) ? 1.0 / rho_eq : 1.0 / rho_in;
        if (tid == L::M_sumha) v = fa[F::fH];
        if (tid == L::M_kap) v = fa[F::fKT + na];
        if (tid == L::M_delta) v = 1.0;
        misc[tid] = v;
    }
    if (tid < 48) red[tid] = 0.0;
    __syncthreads();
    // rhs of the border solve K_0 wbar = k (the T column, internal order)
    for (int ip = tid; ip < na; ip += NT) {
        const int sl = ip < D::nJ ? L::oRhsJ + 56 * (ip / 49) + ip % 49 : (ip < D::nJ + 7 ? L::oRhsU + (ip - D::nJ) : L::oRhsI + (ip - D::nJ - 7));
        lds[sl] = fa[F::fKT + ip];
    }
    const int o_pad = L::oMisc + L::M_pad + (tid & 15);              // write-only slot for the lanes without an output
    const double inv_eq = 1.0 / rho_eq, inv_in = 1.0 / rho_in;
    // ---------------- lane constants: the packed address words, (re)loaded in front of every hot loop ----------------
    unsigned g_tw, g_bx, p1_kt, ribs, p3_sr, p3_yx, p4_ky, p4_dd, v_rx, v_pa, d_pk, o_wgw;
    auto load_words = [&]() {
        typedef const __attribute__((address_space(1))) uint32_t *gptr_t;
        const uint32_t *lp_ = lanes + tid;
        asm volatile("" : "+v"(lp_));
        gptr_t lp = (gptr_t)lp_;
        g_tw = lp[L::F_GTW * NT]; g_bx = lp[L::F_GBX * NT]; p1_kt = lp[L::F_P1KT * NT]; ribs = lp[L::F_RIBS * NT];
        p3_sr = lp[L::F_P3SR * NT]; p3_yx = lp[L::F_P3YX * NT]; p4_ky = lp[L::F_P4KY * NT]; p4_dd = lp[L::F_P4DD * NT];
        v_rx = lp[L::F_VRX * NT]; v_pa = lp[L::F_VPK * NT]; d_pk = lp[L::F_DPK * NT]; o_wgw = lp[L::F_WGW * NT];
    };
    // ---------------- G quad of this lane (both roles): rows 2 lp, 2 lp + 1 of segment tid / 96 ----------------
    double m1[2][14];
    // (loaded by each role right in front of its loop, after its set-up code: the block must not be live across that code)
    auto load_m1 = [&]() {
        typedef const __attribute__((address_space(1))) double *gptr_t;
        const double *fg_ = fa + F::fG + tid;
        asm volatile("" : "+v"(fg_));
        gptr_t fg = (gptr_t)fg_;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int j = 0; j < 14; j++) m1[a][j] = fg[(14 * a + j) * 384];
    };
    auto g_prod = [&](const double *op) -> double {
        D2 bv[7];
#pragma unroll
        for (int j = 0; j < 7; j++) bv[j] = lds2(op + 2 * j);
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            a0 += m1[0][2 * j] * bv[j].x; a1 += m1[1][2 * j] * bv[j].x;
            a0 += m1[0][2 * j + 1] * bv[j].y; a1 += m1[1][2 * j + 1] * bv[j].y;
        }
        return quad_sum2(a0, a1);                                     // even lanes: row 2 lp, odd lanes: row 2 lp + 1
    };
    auto ph_p1a = [&]() {                                             // t = G b_J: rows of the quad; rows < 7 also to the stride-7 copy
        const double tq = g_prod(lds + QP4_LO(g_bx));
        lds[QP4_LO(g_tw)] = tq; lds[QP4_HI(g_tw)] = tq;
    };
    auto ph_p4b = [&]() {                                             // x~_J = G c - wbar x~_T
        const double xT = misc[L::M_xtT];
        const double tq = g_prod(lds + QP4_LO(g_bx) + (L::oCJ - L::oRhsJ));
        const int oxw = QP4_HI(g_bx);
        lds[oxw] = tq - lds[oxw + L::dWb] * xT;
    };
    // path rows: 16 lanes per node; lane (prp, pq) keeps rows 2 prp + (pq & 1) [own] and the other one x columns 6 pq .. + 5.
    // z~ of the owned row (lanes 0, 1 of the quad) and, from the same Jacobian operands, the node's path-row part of A^T w: every
    // lane forms its six columns of g_own w_own + g_other w_other, the four row pairs of the node (lane bits 2, 3 of the DPP row) are
    // summed with two row rotations and the lanes of pair 0 publish the node's 24 padded columns (k_qp2's scheme).
    auto path_rows = [&](const D2 (&p0)[3], const D2 (&p1)[3], const double *xv, double *gdst, auto &&row_update) -> double {
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
        const double wq = row_update(ax);
        const double w0 = dpp_mov<0x44>(wq), w1 = dpp_mov<0x11>(wq);        // quad broadcasts: owner of the own row, of the other row
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double cx = p0[j].x * w0 + p1[j].x * w1, cy = p0[j].y * w0 + p1[j].y * w1;
            cx += dpp_mov<0x128>(cx); cy += dpp_mov<0x128>(cy);             // row_ror:8
            cx += dpp_mov<0x124>(cx); cy += dpp_mov<0x124>(cy);             // row_ror:4
            D2 o; o.x = cx; o.y = cy;
            *reinterpret_cast<D2 *>(gdst + 2 * j) = o;
        }
        return ax;
    };
    // ---------------- shares of P1b, P3, P4a (both roles; lane maps: qp4_build_lanes) ----------------
    auto ph_p1b = [&]() {
        const double *kb = lds + QP4_LO(p1_kt), *tb = lds + QP4_HI(p1_kt);
        double kq[7], tv[7];
#pragma unroll
        for (int d = 0; d < 7; d++) { kq[d] = kb[L::KX * d]; tv[d] = tb[7 * d]; }
        const int orI = QP4_LO(ribs);
        const double bI = lds[orI - (L::oRI - L::oRhsI)];            // b_I of the entry (lanes without one: some finite word, result to a pad slot)
        double acc = ((kq[0] * tv[0] + kq[1] * tv[1]) + (kq[2] * tv[2] + kq[3] * tv[3])) + ((kq[4] * tv[4] + kq[5] * tv[5]) + kq[6] * tv[6]);
        acc = sum4(acc);
        lds[orI] = bI - acc;
    };
    auto ph_p3 = [&]() {
        const double *so = lds + QP4_LO(p3_sr), *sx = so + ((tid & 1) ? -L::SRS : L::SRS), *sr = lds + QP4_HI(p3_sr);
        const double xT = misc[L::M_xtT];
        double a0 = 0.0, a1 = 0.0;
        // (two batches of reads: all fifteen 16-byte reads in flight at once would take 60 registers)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const D2 mo = lds2(so + 2 * j), mx_ = lds2(sx + 2 * j), rv = lds2(sr + 2 * j);
            a0 += mo.x * rv.x; a1 += mx_.x * rv.x;
            a0 += mo.y * rv.y; a1 += mx_.y * rv.y;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 3; j < 5; j++) {
            const D2 mo = lds2(so + 2 * j), mx_ = lds2(sx + 2 * j), rv = lds2(sr + 2 * j);
            a0 += mo.x * rv.x; a1 += mx_.x * rv.x;
            a0 += mo.y * rv.y; a1 += mx_.y * rv.y;
        }
        double yi = quad_sum2(a0, a1);
        yi += dpp_xor4(yi);
        const int oxi = QP4_HI(p3_yx);
        lds[QP4_LO(p3_yx)] = yi;
        lds[oxi] = yi - lds[oxi + L::dWb] * xT;
    };
    constexpr int dCW = L::oCJ - L::oRhsJ;                            // c of a row sits this far behind its b (also for the U rows)
    auto ph_p4a = [&]() {
        const int okr = QP4_LO(p4_ky), odk = QP4_LO(p4_dd), ody = QP4_HI(p4_dd), obs = QP4_HI(ribs);
        const D2 k01 = lds2(lds + okr), k23 = lds2(lds + okr + 2);
        const double *yc = lds + QP4_HI(p4_ky);
        const double y0 = yc[0], y1 = yc[14], y2 = yc[-7], y3 = yc[7];
        const D2 d01 = lds2(lds + odk), d23 = lds2(lds + odk + 2), e01 = lds2(lds + ody), e23 = lds2(lds + ody + 2);
        const double bs = lds[obs];
        double ad = (d01.x * e01.x + d01.y * e01.y) + (d23.x * e23.x + d23.y * e23.y);
        ad = sum4(ad);
        const double sp = (k01.x * y0 + k01.y * y1) + (k23.x * y2 + k23.y * y3);
        lds[obs + dCW] = (bs - sp) - ad;
    };
    // ---------------- one variable per lane (both roles; lane map: qp4_build_lanes); T: state in misc, handled by lane NA + 127 ----------------
    const int vv = (int)lanes[L::F_VV * NT + tid];
    double v_x = 0.0, v_zb = 0.0, v_yb = 0.0, v_lb = 0.0, v_ub = 0.0;
    bool v_req = false;
    if (vv >= 0) {
        const int v = vv;
        double lo, hi;
        if (v < 14 * N) {
            const int k = v / 14, c = v % 14;
            if (k == 0) { lo = hi = x0e[c]; }
            else if (k == N - 1) { lo = xfe[c] - cfg.eps_target; hi = xfe[c] + cfg.eps_target; }
            else { lo = cfg.lbx[c]; hi = cfg.ubx[c]; }
        } else {
            const int c = (v - 14 * N) % 7;
            lo = cfg.lbu[c]; hi = cfg.ubu[c];
        }
        const double zv = zg_[v];
        v_lb = lo - zv; v_ub = hi - zv;
        v_req = hi - lo < 1e-4;
    }
    // wbar_v rhs_v goes to slot v of the x~ area (lanes 81..127 of role A run along without a variable: slots 273..319, which nobody sums)
    auto bp_slot = [&](int t) -> int { asm volatile("" : "+v"(t)); return L::oXn + (t >= L::NA ? t - L::NA : t + L::NB); };
    // (A^T w)[v] of the dynamics rows (w: oWg or, at the tests, the duals) plus the path-row part gp[v]
    auto col_gather = [&](const double *w, const double *gp, unsigned pk, int xpos) -> double {
        const int wA = pk & 255u, wB = (pk >> 8) & 255u, wf = (pk >> 16) & 255u, cdA = (pk >> 24) & 31u, cdB = (pk >> 29) & 1u ? 3 : 16;
        const double *cA = lds + L::oCD + cdA, *cB = lds + L::oCD + cdB;
        const double *wa = w + wA, *wb = w + wB;
        const double a0 = cA[0], a1 = cA[4], a2 = cA[8], b0 = cB[0], b1 = cB[4], b2 = cB[8];
        const double u0 = wa[0], u1 = wa[14], u2 = wa[28], q0 = wb[0], q1 = wb[14], q2 = wb[28];
        const double wfv = w[wf], g = gp[xpos];
        const double cf = wf != meq ? -tsT : 0.0;
        return (g + cf * wfv) + ((a0 * u0 + a1 * u1) + (a2 * u2 + b0 * q0)) + (b1 * q1 + b2 * q2);
    };
    auto var_a = [&]() {                                              // phase A: rhs = sigma x - q + rho_b z_b - y_b + A^T w;  wbar_v rhs_v
        const int oxpos = QP4_HI(v_rx);
        const double r0 = (sigma * v_x + ((v_req ? rho_eq : rho_in) * v_zb - v_yb)) + col_gather(wg, gpl, v_pa, oxpos);
        lds[QP4_LO(v_rx)] = r0;
        lds[bp_slot(tid)] = lds[oxpos + L::oWb] * r0;
    };
    auto var_e = [&]() {                                              // phase E: relaxation, projection, dual update of the variable
        const double xtv = xn[QP4_HI(v_rx)];
        v_x = alpha * xtv + (1.0 - alpha) * v_x;
        const double zr = alpha * xtv + (1.0 - alpha) * v_zb;
        const double zn = clip(zr + v_yb * (v_req ? inv_eq : inv_in), v_lb, v_ub);
        v_yb += (v_req ? rho_eq : rho_in) * (zr - zn);
        v_zb = zn;
    };
    // termination test, variable part (the lane words are re-read from the table: they are not live across the hot loop's exit)
    auto var_ha = [&](unsigned pk) -> double { const int wf = (pk >> 16) & 255u; return wf != meq ? -ts * lam_rows[wf] : 0.0; };
    auto var_chk_p = [&](double (&mp)[3]) {                          // primal part: |x - z|, |x|, |z|
        if (vv >= 0) { mp[0] = fmax(mp[0], fabs(v_x - v_zb)); mp[1] = fmax(mp[1], fabs(v_x)); mp[2] = fmax(mp[2], fabs(v_zb)); }
    };
    auto var_chk_d = [&](unsigned pk, int xpos, double ha, double xTc, double (&md)[3]) {      // dual part: |Hx + A^T y|, |Hx|, |A^T y|
        if (vv >= 0) {
            const double hx = (fabs(ha) + cfg.hess_reg) * v_x + ha * xTc, aty = col_gather(lds + L::oYs, lds + L::oGpy, pk, xpos) + v_yb;
            md[0] = fabs(hx + aty); md[1] = fabs(hx); md[2] = fabs(aty);
        }
    };
    // u_{N-1} block (role A lanes 74..80: they own the variables u_{N-1}): G_u times op
    auto ublock = [&](int t, int opbase) -> double {
        asm volatile("" : "+v"(t));
        const int ur = (t >= 74 && t < 81) ? t - 74 : 0;
        const double *gu = lds + L::oGu + 8 * ur, *op = lds + opbase;
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 7; c++) acc += gu[c] * op[c];
        return acc;
    };
    int it = 0, done = 0;
#ifdef MPCMP_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_busy[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = clock64();
    const unsigned long long st_wall0 = wall_clock64();
#endif
    __syncthreads();

    if (tid < L::NA) {
        // =========================================== role A ===========================================
        const int pk = tid >> 4, prp = (tid & 15) >> 2, pq = tid & 3;
        const bool ownsRow = pq < 2;
        const int prow = 2 * prp + pq;                                  // (owners)
        D2 p0[3], p1[3];
        {
            const double *g0 = Gkg + (pk * 8 + 2 * prp + (pq & 1)) * 22, *g1 = Gkg + (pk * 8 + 2 * prp + 1 - (pq & 1)) * 22;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int c = 6 * pq + 2 * j;
                p0[j].x = c < 22 ? g0[c < 22 ? c : 0] : 0.0; p0[j].y = c + 1 < 22 ? g0[c + 1 < 22 ? c + 1 : 0] : 0.0;
                p1[j].x = c < 22 ? g1[c < 22 ? c : 0] : 0.0; p1[j].y = c + 1 < 22 ? g1[c + 1 < 22 ? c + 1 : 0] : 0.0;
            }
        }
        double lgp = 0.0, ugp = 0.0, cfT = 0.0, zg = 0.0, yg = 0.0;
        bool req = false;
        if (ownsRow) {
            const double gv = ws.g[(size_t)b * 8 * N + 8 * pk + prow];
            lgp = cfg.lbg[prow] - gv; ugp = cfg.ubg[prow] - gv;
            req = ugp - lgp < 1e-4;
            cfT = Gkg[(pk * 8 + prow) * 22 + 21];
        }
        // x~ columns of the lane and where its six columns of the node's A^T w go (lanes of row pairs 1..3: a pad), from the lane index
        auto path_off = [&](int t, int gbase, int &oxv) -> int {
            asm volatile("" : "+v"(t));
            const int nodeo = (t >> 4) * XS + 6 * (t & 3);
            oxv = nodeo;
            return (t & 12) == 0 ? gbase + nodeo : L::oMisc + L::M_pad + 8;
        };
        auto solve_a = [&]() {
            {   // P1a
                ph_p1a();
                if (tid >= 64 && tid < 128) {                            // wave 1: the U block (7 lanes)
                    int t = tid;
                    asm volatile("" : "+v"(t));                          // (addresses derived inside the iteration: hoisted, they are spilled)
                    const bool u = t >= 74 && t < 81;
                    const double tu = ublock(t, L::oRhsU);
                    lds[u ? L::oTU + (t - 74) : L::oMisc + L::M_pad] = tu;
                    lds[u ? L::oCJ + L::TD * NSEG + 7 * (t - 74) : L::oMisc + L::M_pad + 1] = tu;
                }
            }
            QB4(1); __syncthreads(); QS4(1);
            ph_p1b();
            QB4(2); __syncthreads(); QS4(2);
            ph_p3();
            QB4(3); __syncthreads(); QS4(3);
            ph_p4a();
            if (tid >= 128) {                                          // wave 2 (idle in P4a): x~_T at slot 21 of every node
                int t = tid;
                asm volatile("" : "+v"(t));
                const double xT = misc[L::M_xtT];
                lds[t < 128 + N ? L::oXn + (t - 128) * XS + 21 : L::oMisc + L::M_pad + 2] = xT;
            }
            QB4(4); __syncthreads(); QS4(4);
            {   // P4b
                ph_p4b();
                if (tid >= 64 && tid < 128) {
                    int t = tid;
                    asm volatile("" : "+v"(t));
                    const bool u = t >= 74 && t < 81;
                    const double xT = misc[L::M_xtT];
                    const double xu = ublock(t, L::oCJ + 56 * NSEG);
                    const int oxu = u ? L::oXn + (N - 1) * XS + 14 + (t - 74) : L::oMisc + L::M_pad + 3;
                    lds[oxu] = xu - lds[oxu + L::dWb] * xT;               // (lanes without a row: pad + dWb lies behind the wbar vector, a finite word)
                }
            }
            QB4(5); __syncthreads(); QS4(5);
        };
        // border solve K_0 wbar = k (wbar = 0 and x~_T = 0 so far)
        load_words();
        load_m1();
        solve_a();
        for (int i = tid; i < N * XS; i += NT) lds[L::oWb + i] = (i % XS < 21) ? xn[i] : 0.0;      // wbar = the solution of the border solve
        const double kw = vv >= 0 ? fa[F::fKT + int3_of_ext(NSEG, vv >= 0 ? vv : 0)] * xn[QP4_HI(v_rx)] : 0.0;
        __syncthreads();
        lds[bp_slot(tid)] = kw;   // products k_v wbar_v
        __syncthreads();          // (role B: delta)
#ifdef MPCMP_STAMPS
        for (int k = 0; k < 8; k++) st_acc[k] = st_busy[k] = 0;
        st_t = clock64();
#endif
        // The hot loop is the INNER loop (one termination-test period): it contains nothing but the seven phases; the test sits in the
        // outer loop, and the lane words are re-read in front of every period.
        while (it < cfg.qp_iters && !done) {
            const int cnt = cfg.qp_iters - it < cfg.check_every ? cfg.qp_iters - it : cfg.check_every;
            load_words();
            for (int k = 0; k < cnt; k++) {
                if (tid < 128) var_a();         // A: waves 0, 1 own the variables 192..272
                QB4(0); __syncthreads(); QS4(0);
                solve_a();
                if (tid < 128) var_e();
                // ---- E: path rows of nodes 0..11 ----
                int oxv;
                const int ogd = path_off(tid, L::oGp, oxv);
                path_rows(p0, p1, xn + oxv, lds + ogd, [&](double zt) -> double {
                    double w = 0.0;
                    if (ownsRow) {
                        const double zr = alpha * zt + (1.0 - alpha) * zg;
                        const double zn = clip(zr + yg * (req ? inv_eq : inv_in), lgp, ugp);
                        yg += (req ? rho_eq : rho_in) * (zr - zn);
                        zg = zn;
                        w = (req ? rho_eq : rho_in) * zg - yg;
                    }
                    return w;
                });
                QB4(6); __syncthreads(); QS4(6);
            }
            it += cnt;
            if (QP4_CHECK_ON && cnt == cfg.check_every) {
                const uint32_t *lp_ = lanes + tid;
                asm volatile("" : "+v"(lp_));
                const unsigned pkc = lp_[L::F_VPK * NT];
                const int xposc = QP4_HI(lp_[L::F_VRX * NT]);
                const double xTc = misc[L::M_xT];
                if (vv >= 0) lds[L::oXx + xposc] = v_x;
                __syncthreads();                                       // (x, y published)
                // (primal maxima, the two sums and the dual maxima are reduced one after the other: few values live at a time)
                double mp[3] = {0, 0, 0};
                {
                    int oxvc;
                    const int ogdy = path_off(tid, L::oGpy, oxvc);
                    const double ax = path_rows(p0, p1, lds + L::oXx + oxvc, lds + ogdy, [&](double) -> double { return ownsRow ? yg : 0.0; });
                    if (ownsRow) { mp[0] = fabs(ax - zg); mp[1] = fabs(ax); mp[2] = fabs(zg); }
                }
                var_chk_p(mp);
                block_reduce_dpp<6, 3, true>(mp, red, tid);             // (its barriers publish gpy)
                const double ha = var_ha(pkc);
                double sums[2] = {ownsRow ? cfT * yg : 0.0, ha * v_x};
                block_reduce_dpp<6, 2, false>(sums, red, tid);
                double md[3] = {0, 0, 0};
                var_chk_d(pkc, xposc, ha, xTc, md);
                block_reduce_dpp<6, 3, true>(md, red, tid);
                const double ep = cfg.eps_abs + cfg.eps_rel * fmax(mp[1], mp[2]);
                const double ed = cfg.eps_abs + cfg.eps_rel * fmax(fmax(md[1], md[2]), 1.0);
                done = (mp[0] <= ep && md[0] <= ed) ? 1 : 0;
                __syncthreads();                                       // (the overlay is dead again)
                QS4(7);
            }
        }
#ifdef MPCMP_STAMPS
        if (tid == 0) {
            unsigned long long *o = ws.dbg + (size_t)b * MPCMP_DBG_WORDS;
            for (int k = 0; k < 8; k++) { o[k] = st_acc[k]; o[16 + k] = st_busy[k]; }
            o[15] = it;
            o[150] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_ID
            o[151] = st_wall0; o[152] = wall_clock64();
            o[153] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));     // XCC_ID
        }
#endif
        if (ownsRow) ws.y[(size_t)b * mn_tot + meq + 8 * pk + prow] = yg;
        if (vv >= 0) { ws.p[(size_t)b * n_tot + vv] = v_x; ws.y[(size_t)b * mn_tot + ma + vv] = v_yb; }
    } else {
        // =========================================== role B ===========================================
        const int bt = tid - L::NA;                                     // 0..191
        // Addresses that only a few lanes of one wave need (rows 48 and path rows of node N - 1: lanes bt 176..191 of wave 5) are
        // derived from the lane index where they are used, not kept in registers.
        const bool is48 = bt >= 176;
        // ---- dynamics row bt (bt < meq) ----
        const bool isDyn = bt < meq;
        double d_y = 0.0, d_z = 0.0, d_l = 0.0, d_cT = 0.0;
        if (isDyn) {
            const int r = bt, k = r / 14, rr = r % 14;
            d_l = -ws.ceq[(size_t)b * meq + r];
            d_cT = -ts * zg_[(rr < 7) ? 14 * k + 7 + rr : 14 * N + 7 * k + rr - 7];
        }
        const bool waveDyn = (bt & ~63) < meq;
        // ---- path rows of node N - 1: lanes bt 176..191 (one DPP row of wave 5); Jacobian and row constants in LDS ----
        const int prp = (tid & 15) >> 2, pq = tid & 3;
        const bool ownsRow = is48 && pq < 2;
        const int prow = 2 * prp + pq;
        double zg = 0.0, yg = 0.0;
        if (ownsRow) {
            const double gv = ws.g[(size_t)b * 8 * N + 8 * (N - 1) + prow];
            const double lg = cfg.lbg[prow] - gv, ug = cfg.ubg[prow] - gv;
            double *pc = lds + L::oP12 + prow;
            pc[0] = lg; pc[8] = ug; pc[16] = Gkg[((N - 1) * 8 + prow) * 22 + 21]; pc[24] = (ug - lg < 1e-4) ? rho_eq : rho_in; pc[32] = (ug - lg < 1e-4) ? inv_eq : inv_in;
        }
        auto row_dot_dyn = [&](const double *xe, unsigned dw) -> double {
            const int o_dx0 = dw & 511u, o_dxf = (dw >> 9) & 511u, o_dci = dw >> 18;
            const double *cd = lds + L::oCD + o_dci, *x0 = xe + o_dx0;
            const double c0 = cd[0], c1 = cd[1], c2 = cd[2], c3 = cd[3];
            const double x_0 = x0[0], x_1 = x0[XS], x_2 = x0[2 * XS], x_3 = x0[3 * XS], xf_ = xe[o_dxf], xT_ = xe[21];
            return ((c0 * x_0 + c1 * x_1) + (c2 * x_2 + c3 * x_3)) + (d_cT * xT_ - tsT * xf_);
        };
        // row 48 of segment (bt - 176) >> 2: four lanes, a quarter row each; returns the row's dot product with op[segment]
        auto row48 = [&](int t, int opbase) -> double {
            asm volatile("" : "+v"(t));
            const int sgm = (t - (L::NA + 176)) >> 2, pt = t & 3;
            const int off = (t >= L::NA + 176) ? 56 * sgm + 14 * pt : 0;
            const double *g48 = lds + (t >= L::NA + 176 ? L::oG48 : L::oZero) + off, *o48 = lds + opbase + off;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 7; j++) { const D2 gq = lds2(g48 + 2 * j), bq = lds2(o48 + 2 * j); acc += gq.x * bq.x; acc += gq.y * bq.y; }
            return sum4(acc);
        };
        auto solve_b = [&](const bool first) {
            // ---- P1a: t = G b_J (segments 2, 3), rows 48 ----
            {
                ph_p1a();
                if (bt >= 128) {                                          // wave 5: rows 48 (16 lanes)
                    int t = tid;
                    asm volatile("" : "+v"(t));
                    const double t48 = row48(t, L::oRhsJ);
                    lds[(t >= L::NA + 176 && (t & 3) == 0) ? L::oTJ + 56 * ((t - (L::NA + 176)) >> 2) + 48 : L::oMisc + L::M_pad + 4] = t48;
                }
            }
            QB4(1); __syncthreads(); QS4(1);
            // ---- P1b: r_I = b_I - K_CJ t (waves 3, 4); wave 5: x~_T = (base_T + (T column of A^T w) - wbar^T rhs) / delta ----
            if (bt < 128) ph_p1b();
            else if (!first) {
                const int ln = bt - 128;
                const double t0 = tpl[ln], t1 = tpl[ln + 64], t2 = lds[ln < meq - 128 ? L::oTp + 128 + ln : L::oZero];
                const double g0 = lds[ln < N ? L::oGp + ln * XS + 21 : L::oZero];
                const double b0 = bpl[ln], b1 = bpl[ln + 64], b2 = bpl[ln + 128], b3 = bpl[ln + 192], b4 = lds[ln < na - 256 ? L::oXn + 256 + ln : L::oZero];
                double sacc = ((t0 + t1) + (t2 + g0)) - (((b0 + b1) + (b2 + b3)) + b4);
                sacc = wave_sum(sacc);
                if (ln == 0) misc[L::M_xtT] = (misc[L::M_baseT] + sacc) / misc[L::M_delta];
            }
            QB4(2); __syncthreads(); QS4(2);
            ph_p3();
            QB4(3); __syncthreads(); QS4(3);
            ph_p4a();
            QB4(4); __syncthreads(); QS4(4);
            // ---- P4b: x~_J = G c - wbar x~_T; rows 48 ----
            {
                ph_p4b();
                if (bt >= 128) {
                    int t = tid;
                    asm volatile("" : "+v"(t));
                    const double xT = misc[L::M_xtT];
                    const double x48 = row48(t, L::oCJ);
                    // row 48 of segment s is the last entry of u_{3s+2}: slot 20 of node 3 s + 2
                    const int o48 = (t >= L::NA + 176 && (t & 3) == 0) ? L::oXn + (3 * ((t - (L::NA + 176)) >> 2) + 2) * XS + 20 : L::oMisc + L::M_pad + 5;
                    lds[o48] = x48 - lds[o48 + L::dWb] * xT;
                }
            }
            QB4(5); __syncthreads(); QS4(5);
        };
        // path rows of node N - 1 (wave 5): x from xe, the node's part of A^T w to gbase
        auto path12 = [&](int t, const double *xe, int gbase, auto &&row_update) -> double {
            asm volatile("" : "+v"(t));
            const int rp = (t & 15) >> 2, q = t & 3;
            const int j0 = L::oJ12 + (2 * rp + (q & 1)) * XS + 6 * q, j1 = L::oJ12 + (2 * rp + 1 - (q & 1)) * XS + 6 * q;
            D2 p0[3], p1[3];
#pragma unroll
            for (int j = 0; j < 3; j++) { p0[j] = lds2(lds + j0 + 2 * j); p1[j] = lds2(lds + j1 + 2 * j); }
            double *gdst = lds + ((t >= L::NA + 176 && rp == 0) ? gbase + (N - 1) * XS + 6 * q : L::oMisc + L::M_pad + 8);
            return path_rows(p0, p1, xe + (N - 1) * XS + 6 * q, gdst, row_update);
        };
        // border solve K_0 wbar = k (wbar = 0, x~_T = 0 so far), then delta = kappa + h_TT + sigma + rho_T - k^T wbar
        load_words();
        load_m1();
        solve_b(true);
        for (int i = tid; i < N * XS; i += NT) lds[L::oWb + i] = (i % XS < 21) ? xn[i] : 0.0;
        const double kw = fa[F::fKT + int3_of_ext(NSEG, vv)] * xn[QP4_HI(v_rx)];
        __syncthreads();
        lds[bp_slot(tid)] = kw;
        __syncthreads();
        if (bt >= 128) {
            const int ln = bt - 128;
            double sacc = ((bpl[ln] + bpl[ln + 64]) + (bpl[ln + 128] + bpl[ln + 192])) + lds[ln < na - 256 ? L::oXn + 256 + ln : L::oZero];
            sacc = wave_sum(sacc);
            if (ln == 0) {
                const double hdT = misc[L::M_sumha] + cfg.hess_reg;
                misc[L::M_hdT] = hdT;
                const double delta = (misc[L::M_kap] + (hdT + sigma + misc[L::M_rbT])) - sacc;
                misc[L::M_delta] = delta;
                if (!(delta > 0.0)) atomicOr(&ws.status[b], 2);
            }
        }
#ifdef MPCMP_STAMPS
        for (int k = 0; k < 8; k++) st_acc[k] = st_busy[k] = 0;
        st_t = clock64();
#endif
        while (it < cfg.qp_iters && !done) {
            const int cnt = cfg.qp_iters - it < cfg.check_every ? cfg.qp_iters - it : cfg.check_every;
            load_words();
            for (int k = 0; k < cnt; k++) {
                var_a();              // A
                QB4(0); __syncthreads(); QS4(0);
                solve_b(false);
                // ---- E: dynamics rows, variables, T, path rows of node N - 1 ----
                if (waveDyn) {
                    const double zt = row_dot_dyn(xn, d_pk);
                    const double zr = alpha * zt + (1.0 - alpha) * d_z;
                    d_y += rho_eq * (zr - d_l);                          // the row is an equality: the projection of anything onto [l, l] is l
                    d_z = d_l;
                    const double w = rho_eq * d_l - d_y;
                    lds[o_wgw] = w; lds[o_wgw + (L::oTp - L::oWg)] = d_cT * w;      // (lanes without a row: the pad slots wg[meq + 1], tp[meq + 1])
                }
                var_e();
                if (bt >= 128) {
                    path12(tid, xn, L::oGp, [&](double zt) -> double {
                        double w = 0.0;
                        if (ownsRow) {
                            const double *pc = lds + L::oP12 + prow;
                            const double lgp = pc[0], ugp = pc[8], rr_ = pc[24], rri = pc[32];
                            const double zr = alpha * zt + (1.0 - alpha) * zg;
                            const double zn = clip(zr + yg * rri, lgp, ugp);
                            yg += rr_ * (zr - zn);
                            zg = zn;
                            w = rr_ * zg - yg;
                        }
                        return w;
                    });
                }
                if (bt == 127) {        // the shared variable T (state in misc)
                    const double xtv = misc[L::M_xtT], rb = misc[L::M_rbT];
                    double xx = misc[L::M_xT], zz = misc[L::M_zbT], yy = misc[L::M_ybT];
                    xx = alpha * xtv + (1.0 - alpha) * xx;
                    const double zr = alpha * xtv + (1.0 - alpha) * zz;
                    const double zn = clip(zr + yy * misc[L::M_rbiT], misc[L::M_lbT], misc[L::M_ubT]);
                    yy += rb * (zr - zn);
                    zz = zn;
                    misc[L::M_xT] = xx; misc[L::M_zbT] = zz; misc[L::M_ybT] = yy;
                    misc[L::M_baseT] = (sigma * xx - 1.0) + (rb * zz - yy);
                }
                QB4(6); __syncthreads(); QS4(6);
            }
            it += cnt;
            if (QP4_CHECK_ON && cnt == cfg.check_every) {
                // ---- termination test: r_prim = ||[A;I]x - z||inf, r_dual = ||Hx + q + [A;I]^T y||inf (oracle/ocp.c admm) ----
                const uint32_t *lp_ = lanes + tid;
                asm volatile("" : "+v"(lp_));
                const unsigned pkc = lp_[L::F_VPK * NT], dwc = lp_[L::F_DPK * NT];
                const int xposc = QP4_HI(lp_[L::F_VRX * NT]);
                double *xx = lds + L::oXx, *ys = lds + L::oYs;
                const double xTc = misc[L::M_xT];
                xx[xposc] = v_x;
                if (bt >= 128 && bt < 128 + N) xx[(bt - 128) * XS + 21] = xTc;
                if (bt >= 128 + 16 && bt < 128 + 16 + N) { xx[(bt - 144) * XS + 22] = 0.0; xx[(bt - 144) * XS + 23] = 0.0; }
                if (isDyn) ys[bt] = d_y;
                if (bt >= 168 && bt < 170) ys[bt] = 0.0;
                __syncthreads();
                double mp[3] = {0, 0, 0};                                // rp, |Ax|, |z|
                double s0 = isDyn ? d_cT * d_y : 0.0;                    // T row: sum coefT_r y_r
                if (isDyn) {
                    const double ax = row_dot_dyn(xx, dwc);
                    mp[0] = fabs(ax - d_z); mp[1] = fabs(ax); mp[2] = fabs(d_z);
                }
                if (bt >= 128) {
                    const double ax = path12(tid, xx, L::oGpy, [&](double) -> double { return ownsRow ? yg : 0.0; });
                    if (ownsRow) { s0 += lds[L::oP12 + 16 + prow] * yg; mp[0] = fmax(mp[0], fabs(ax - zg)); mp[1] = fmax(mp[1], fabs(ax)); mp[2] = fmax(mp[2], fabs(zg)); }
                }
                var_chk_p(mp);
                if (bt == 127) { const double zT = misc[L::M_zbT]; mp[0] = fmax(mp[0], fabs(xTc - zT)); mp[1] = fmax(mp[1], fabs(xTc)); mp[2] = fmax(mp[2], fabs(zT)); }
                block_reduce_dpp<6, 3, true>(mp, red, tid);             // (its barriers publish gpy)
                const double ha = var_ha(pkc);
                double sums[2] = {s0, ha * v_x};                         // sum coefT_r y_r, sum ha_i x_i
                block_reduce_dpp<6, 2, false>(sums, red, tid);
                double md[3] = {0, 0, 0};                                // rd, |Hx|, |A^T y|
                var_chk_d(pkc, xposc, ha, xTc, md);
                if (bt == 127) {
                    const double hxT = misc[L::M_hdT] * xTc + sums[1], atyT = sums[0] + misc[L::M_ybT];
                    md[0] = fmax(md[0], fabs(hxT + atyT + 1.0)); md[1] = fmax(md[1], fabs(hxT)); md[2] = fmax(md[2], fabs(atyT));
                }
                block_reduce_dpp<6, 3, true>(md, red, tid);
                const double ep = cfg.eps_abs + cfg.eps_rel * fmax(mp[1], mp[2]);
                const double ed = cfg.eps_abs + cfg.eps_rel * fmax(fmax(md[1], md[2]), 1.0);      // ||q||_inf = 1
                done = (mp[0] <= ep && md[0] <= ed) ? 1 : 0;
                __syncthreads();
            }
        }
#ifdef MPCMP_STAMPS
        if ((tid & 63) == 0) {
            unsigned long long *o = ws.dbg + (size_t)b * MPCMP_DBG_WORDS;
            const int w = tid >> 6;
            for (int k = 0; k < 8; k++) { o[32 + (w - 3) * 16 + k] = st_acc[k]; o[32 + (w - 3) * 16 + 8 + k] = st_busy[k]; }
        }
#endif
        // ---------------- results ----------------
        ws.p[(size_t)b * n_tot + vv] = v_x; ws.y[(size_t)b * mn_tot + ma + vv] = v_yb;
        if (isDyn) ws.y[(size_t)b * mn_tot + bt] = d_y;
        if (ownsRow) ws.y[(size_t)b * mn_tot + meq + 8 * (N - 1) + prow] = yg;
        if (bt == 127) {
            ws.p[(size_t)b * n_tot + na] = misc[L::M_xT];
            ws.y[(size_t)b * mn_tot + mn_tot - 1] = misc[L::M_ybT];
            ws.qpit[b] = it; ws.qp_total[b] += it;
            if (!done) atomicAdd(&ws.status[b], MPCMP_ST_CAP_ONE);
        }
    }
}

}  // namespace mpcmp
