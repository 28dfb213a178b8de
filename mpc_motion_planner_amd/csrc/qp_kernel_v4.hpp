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
    static constexpr int size = oRed + 48;
    static_assert(size * 8 + 512 <= 80 * 1024, "two workgroups per CU: 80 KB of LDS each");
    // termination-test overlay
    static constexpr int oXx = oRhsJ;                               // [N][XS] x node-major
    static constexpr int oGpy = oXx + D::N * XS;                    // [N][XS] path-row part of A^T y
    static constexpr int oYs = oGpy + D::N * XS;                    // [meq + 2] duals of the dynamics rows
    static_assert(oYs + D::meq + 2 <= oEndV, "termination-test vectors must fit over the solve vectors");
    static_assert(oS % 2 == 0 && oXn % 2 == 0 && oGp % 2 == 0 && oRhsJ % 2 == 0 && oTJ % 2 == 0 && oCJ % 2 == 0 && oRI % 2 == 0 && oYI % 2 == 0 &&
                  oKJC % 2 == 0 && oKX % 2 == 0 && oJ12 % 2 == 0 && oG48 % 2 == 0 && oZero % 2 == 0 && oMisc % 2 == 0 && oGpy % 2 == 0, "16-byte LDS accesses");
    // misc slots
    static constexpr int M_xtT = 0, M_xT = 1, M_zbT = 2, M_ybT = 3, M_baseT = 4, M_delta = 5, M_hdT = 6, M_rbT = 7, M_lbT = 8, M_ubT = 9,
                         M_sumha = 10, M_kap = 11, M_pad = 16 /* 16..31: write-only pad slots */;
};

template <int NSEG>
__global__ __launch_bounds__(384, 3) void k_qp4(mpcmp_config cfg, WS ws, const Qp3Pat *__restrict__ pat, int B, const double *__restrict__ fac) {
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
    if (tid < 48) lds[L::oZero + tid] = 0.0;
    if (tid < 32) lds[L::oCD + tid] = tid < 16 ? c_D[tid] : 0.0;
    if (tid < 32) {
        double v = 0.0;
        if (tid == L::M_baseT) v = -1.0;                             // sigma x_T - q_T + rho_T z_T - y_T with x = z = y = 0, q_T = 1 (cost = T)
        if (tid == L::M_lbT) v = cfg.lbT - T;
        if (tid == L::M_ubT) v = cfg.ubT - T;
        if (tid == L::M_rbT) v = (cfg.ubT - cfg.lbT < 1e-4) ? rho_eq : rho_in;
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
    auto node_slot = [&](int v) -> int { return v < 14 * N ? XS * (v / 14) + v % 14 : XS * ((v - 14 * N) / 7) + 14 + (v - 14 * N) % 7; };
    const int o_pad = L::oMisc + L::M_pad + (tid & 15);              // write-only slot for the lanes without an output
    // ---------------- G quad of this lane (both roles): rows 2 lp, 2 lp + 1 of segment gseg ----------------
    const int Q = tid >> 2, part = tid & 3, gseg = Q / 24, glp = Q % 24;
    double m1[2][14];
    {
        typedef const __attribute__((address_space(1))) double *gptr_t;
        gptr_t fg = (gptr_t)(fa + F::fG + tid);
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int j = 0; j < 14; j++) m1[a][j] = fg[(14 * a + j) * 384];
    }
    const int grow = 2 * glp + part;                                  // output row of this lane (lanes 0, 1 of the quad)
    const bool gout = part < 2;
    const int o_bj = L::oRhsJ + 56 * gseg + 14 * part;                // operand of t = G b_J; the operand of x_J = G c sits oCJ - oRhsJ further
    const int o_tw = gout ? L::oTJ + 56 * gseg + grow : o_pad;
    const int o_tdw = (gout && grow < 7) ? L::oCJ + L::TD * gseg + 7 * grow : o_pad;
    const int o_xw = gout ? L::oXn + node_slot(ws.ext_of_int[49 * gseg + (gout ? grow : 0)]) : o_pad;
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
    const double inv_eq = 1.0 / rho_eq, inv_in = 1.0 / rho_in;
    // ---------------- shares of P1b, P3, P4a (both roles) ----------------
    // P1b: lane 4 i + pp of entry i: pp = 0 opening segment, 1 closing segment, 2 dense block.  Role A: entries 0..47, role B: 48..69.
    const int h1 = tid < L::NA ? tid : tid - L::NA + 4 * 48;
    int o_kb = L::oKX, o_tb = L::oZero, o_rI = o_pad, o_bI = L::oRhsI;
    if (h1 < 4 * nI && (tid < L::NA || tid - L::NA < 4 * (nI - 48))) {
        const int i = h1 >> 2, pp = h1 & 3, nd = i / 14, c = i % 14;
        if (pp == 0 && nd < NSEG) { o_kb = L::oKX + nd * 7 * L::KX + c; o_tb = L::oTJ + 56 * nd + c - 7; }
        if (pp == 1 && nd >= 1) { o_kb = L::oKX + (nd - 1) * 7 * L::KX + 14 + c; o_tb = L::oTJ + 56 * (nd - 1) + c - 7; }
        if (pp == 2) { o_kb = L::oKX + nd * 7 * L::KX + 28 + c; o_tb = L::oCJ + L::TD * nd; }
        if (pp == 0) { o_rI = L::oRI + i; o_bI = L::oRhsI + i; }
    }
    auto ph_p1b = [&]() {
        const double *kb = lds + o_kb, *tb = lds + o_tb;
        double kq[7], tv[7];
#pragma unroll
        for (int d = 0; d < 7; d++) { kq[d] = kb[L::KX * d]; tv[d] = tb[7 * d]; }
        const double bI = lds[o_bI];
        double acc = ((kq[0] * tv[0] + kq[1] * tv[1]) + (kq[2] * tv[2] + kq[3] * tv[3])) + ((kq[4] * tv[4] + kq[5] * tv[5]) + kq[6] * tv[6]);
        acc = sum4(acc);
        lds[o_rI] = bI - acc;
    };
    // P3: 8 lanes (7 used) per row pair of S^-1.  Role B: row pairs 0..23, role A: 24..34.
    const int h3 = tid < L::NA ? tid + 8 * 24 : tid - L::NA;
    const int sg = h3 >> 3, scs = h3 & 7;
    const bool isS = (tid >= L::NA || tid < 8 * (nI / 2 - 24)) && scs < 7;
    const int o_so = L::oS + (isS ? (2 * sg + (scs & 1)) * L::SRS + 10 * scs : 0), o_sx = L::oS + (isS ? (2 * sg + 1 - (scs & 1)) * L::SRS + 10 * scs : 0);
    const int o_sr = isS ? L::oRI + 10 * scs : L::oZero;
    const bool sOut = isS && scs < 2;
    const int s_row = sOut ? 2 * sg + scs : 0;
    const int o_yw = sOut ? L::oYI + s_row : o_pad;
    const int o_xiw = sOut ? L::oXn + 3 * (s_row / 14) * XS + s_row % 14 : o_pad;
    double s_wb = 0.0;
    auto ph_p3 = [&]() {
        const double *so = lds + o_so, *sx = lds + o_sx, *sr = lds + o_sr;
        const double xT = misc[L::M_xtT];
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const D2 mo = lds2(so + 2 * j), mx_ = lds2(sx + 2 * j), rv = lds2(sr + 2 * j);
            a0 += mo.x * rv.x; a1 += mx_.x * rv.x;
            a0 += mo.y * rv.y; a1 += mx_.y * rv.y;
        }
        double yi = quad_sum2(a0, a1);
        yi += dpp_xor4(yi);
        lds[o_yw] = yi;
        lds[o_xiw] = yi - s_wb * xT;
    };
    // P4a: role B: the 35 rows with a dense block (4 lanes each: u_3s rows of the four segments, then u_{N-1}), then 52 further interior
    // rows; role A: the other 116 interior rows.  Interior rows 7..48 of segment s4 in the order sr = 42 s4 + (lr - 7).
    int o_kr = L::oKJC + NSEG * 196, o_yc = L::oYI, o_bs = L::oZero, o_dk = L::oZero, o_dy = L::oZero, o_cw = o_pad;
    {
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
                    o_yc = L::oYI + 14 * s4 + (int)(pat->jc[lr] & 255u);      // canonical slots [base, base + 14, base - 7, base + 7] (structure3.hpp)
                    o_bs = L::oRhsJ + 56 * s4 + lr;
                    o_cw = L::oCJ + 56 * s4 + lr;
                } else { o_bs = L::oRhsU + lr; o_cw = L::oCJ + 56 * NSEG + lr; }
            }
        }
    }
    const int o_xtc = (tid >= 128 && tid < 128 + N) ? L::oXn + (tid - 128) * XS + 21 : o_pad;      // x~_T copies (slot 21 of every node), made in P4a
    auto ph_p4a = [&]() {
        const D2 k01 = lds2(lds + o_kr), k23 = lds2(lds + o_kr + 2);
        const double *yc = lds + o_yc;
        const double y0 = yc[0], y1 = yc[14], y2 = yc[-7], y3 = yc[7];
        const D2 d01 = lds2(lds + o_dk), d23 = lds2(lds + o_dk + 2), e01 = lds2(lds + o_dy), e23 = lds2(lds + o_dy + 2);
        const double bs = lds[o_bs], xT = misc[L::M_xtT];
        double ad = (d01.x * e01.x + d01.y * e01.y) + (d23.x * e23.x + d23.y * e23.y);
        ad = sum4(ad);
        const double sp = (k01.x * y0 + k01.y * y1) + (k23.x * y2 + k23.y * y3);
        lds[o_cw] = (bs - sp) - ad;
        lds[o_xtc] = xT;
    };
    int it = 0, done = 0, until_check = cfg.check_every;
    double wrow = 0.0;                                                // border vector entry of the G output row
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
        const double rr_ = req ? rho_eq : rho_in, rri = req ? inv_eq : inv_in;
        const int o_xv = L::oXn + pk * XS + 6 * pq;
        const int o_gd = prp == 0 ? L::oGp + pk * XS + 6 * pq : L::oMisc + L::M_pad + 8;
        const int o_gdy = prp == 0 ? L::oGpy + pk * XS + 6 * pq : L::oMisc + L::M_pad + 8;
        auto solve_a = [&]() {
            {   // P1a
                const double tq = g_prod(lds + o_bj);
                lds[o_tw] = tq; lds[o_tdw] = tq;
            }
            __syncthreads();
            ph_p1b();
            __syncthreads();
            ph_p3();
            __syncthreads();
            ph_p4a();
            __syncthreads();
            {   // P4b
                const double xT = misc[L::M_xtT];
                const double tq = g_prod(lds + o_bj + (L::oCJ - L::oRhsJ));
                lds[o_xw] = tq - wrow * xT;
            }
            __syncthreads();
        };
        // border solve K_0 wbar = k
        solve_a();
        wrow = gout ? lds[o_xw] : 0.0;
        s_wb = sOut ? lds[o_xiw] : 0.0;
        __syncthreads();          // (role B: products k_v wbar_v)
        __syncthreads();          // (role B: delta)
        for (it = 1; it <= cfg.qp_iters; it++) {
            __syncthreads();      // A (role B)
            solve_a();
            // ---- E: path rows of nodes 0..11 ----
            path_rows(p0, p1, lds + o_xv, lds + o_gd, [&](double zt) -> double {
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
            __syncthreads();
            if (--until_check == 0) {
                until_check = cfg.check_every;
                __syncthreads();                                       // (role B publishes x, y)
                double sums[2] = {ownsRow ? cfT * yg : 0.0, 0.0};
                double mx[6] = {0, 0, 0, 0, 0, 0};
                const double ax = path_rows(p0, p1, lds + L::oXx + pk * XS + 6 * pq, lds + o_gdy, [&](double) -> double { return ownsRow ? yg : 0.0; });
                if (ownsRow) { mx[0] = fabs(ax - zg); mx[1] = fabs(ax); mx[2] = fabs(zg); }
                block_reduce_dpp<6, 2, false>(sums, red, tid);
                block_reduce_dpp<6, 6, true>(mx, red, tid);
                const double ep = cfg.eps_abs + cfg.eps_rel * fmax(mx[1], mx[2]);
                const double ed = cfg.eps_abs + cfg.eps_rel * fmax(fmax(mx[4], mx[5]), 1.0);
                done = (mx[0] <= ep && mx[3] <= ed) ? 1 : 0;
                __syncthreads();                                       // (the overlay is dead again)
                if (done) break;
            }
        }
        if (ownsRow) ws.y[(size_t)b * mn_tot + meq + 8 * pk + prow] = yg;
    } else {
        // =========================================== role B ===========================================
        const int bt = tid - L::NA;                                     // 0..191
        // ---- rows 48 of the four segments: lanes bt 176..191 (4 lanes per row) ----
        const bool is48 = bt >= 176;
        const int r48s = (bt - 176) >> 2;
        const int o_g48 = is48 ? L::oG48 + 56 * r48s + 14 * part : L::oZero;
        const int o_b48 = is48 ? L::oRhsJ + 56 * r48s + 14 * part : L::oZero;
        const bool out48 = is48 && part == 0;
        const int o_t48 = out48 ? L::oTJ + 56 * r48s + 48 : o_pad;
        const int o_x48 = out48 ? L::oXn + node_slot(ws.ext_of_int[49 * (out48 ? r48s : 0) + 48]) : o_pad;
        double w48 = 0.0;
        // ---- u_{N-1} block: lanes bt 74..80 (they own the variables u_{N-1} as their second variable) ----
        const bool isU = bt >= 74 && bt < 81;
        const int ur = isU ? bt - 74 : 0;
        const int o_gu = L::oGu + 8 * ur;
        const int o_tuw = isU ? L::oTU + ur : o_pad, o_tduw = isU ? L::oCJ + L::TD * NSEG + 7 * ur : o_pad;
        const int o_xuw = isU ? L::oXn + (N - 1) * XS + 14 + ur : o_pad;
        // ---- variables: lane bt owns bt and (bt < na - 192) 192 + bt, external arm order; T: state in misc, handled by lane 191 ----
        constexpr int NV2 = na - L::NB;                                  // lanes with a second variable (81)
        double v_x[2] = {0, 0}, v_zb[2] = {0, 0}, v_yb[2] = {0, 0}, v_lb[2] = {0, 0}, v_ub[2] = {0, 0}, v_wb[2] = {0, 0}, v_kt[2] = {0, 0}, v_cf[2] = {0, 0};
        bool v_req[2] = {false, false};
        int o_rhs[2] = {o_pad, o_pad}, o_xpos[2] = {0, 0}, o_wA[2] = {0, 0}, o_wB[2] = {0, 0}, o_wf[2] = {meq, meq}, o_cdA[2] = {16, 16}, o_cdB[2] = {16, 16};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int v = bt + L::NB * h;
            if (v < na) {
                const int ip = int3_of_ext(NSEG, v);
                o_rhs[h] = ip < D::nJ ? L::oRhsJ + 56 * (ip / 49) + ip % 49 : (ip < D::nJ + 7 ? L::oRhsU + (ip - D::nJ) : L::oRhsI + (ip - D::nJ - 7));
                v_kt[h] = fa[F::fKT + ip];
                double lo, hi;
                if (v < 14 * N) {
                    const int k = v / 14, c = v % 14;
                    if (k == 0) { lo = hi = x0e[c]; }
                    else if (k == N - 1) { lo = xfe[c] - cfg.eps_target; hi = xfe[c] + cfg.eps_target; }
                    else { lo = cfg.lbx[c]; hi = cfg.ubx[c]; }
                    o_xpos[h] = k * XS + c;
                    if (k % 3 != 0) { o_wA[h] = 14 * 3 * (k / 3) + c; o_cdA[h] = k % 3; }
                    else {
                        if (k < N - 1) { o_wA[h] = 14 * k + c; o_cdA[h] = 0; }
                        if (k > 0) { o_wB[h] = 14 * (k - 3) + c; o_cdB[h] = 3; }
                    }
                    if (c >= 7 && k <= N - 2) { o_wf[h] = 14 * k + (c - 7); v_cf[h] = -tsT; }
                } else {
                    const int k = (v - 14 * N) / 7, c = (v - 14 * N) % 7;
                    lo = cfg.lbu[c]; hi = cfg.ubu[c];
                    o_xpos[h] = k * XS + 14 + c;
                    if (k <= N - 2) { o_wf[h] = 14 * k + 7 + c; v_cf[h] = -tsT; }
                }
                v_req[h] = hi - lo < 1e-4;
                const double zv = zg_[v];
                v_lb[h] = lo - zv; v_ub[h] = hi - zv;
            }
        }
        const bool wave2 = (bt & ~63) < NV2;                             // some lane of this wave owns a second variable (waves 3, 4)
        // ---- dynamics row bt (bt < meq) ----
        const bool isDyn = bt < meq;
        double d_y = 0.0, d_z = 0.0, d_l = 0.0, d_cT = 0.0;
        int o_dx0 = 0, o_dxf = 0, o_dci = 16;
        if (isDyn) {
            const int r = bt, k = r / 14, rr = r % 14;
            o_dx0 = 3 * (k / 3) * XS + rr;
            o_dxf = k * XS + (rr < 7 ? 7 + rr : 14 + rr - 7);
            o_dci = 4 * (k % 3);
            d_l = -ws.ceq[(size_t)b * meq + r];
            d_cT = -ts * zg_[(rr < 7) ? 14 * k + 7 + rr : 14 * N + 7 * k + rr - 7];
        }
        const bool waveDyn = (bt & ~63) < meq;
        const int o_wgw = isDyn ? L::oWg + bt : o_pad, o_tpw = isDyn ? L::oTp + bt : o_pad;
        // ---- path rows of node N - 1: lanes bt 176..191 (one DPP row of wave 5), Jacobian in LDS ----
        const bool isP12 = bt >= 176;
        const int prp = (tid & 15) >> 2, pq = tid & 3;
        const bool ownsRow = isP12 && pq < 2;
        const int prow = 2 * prp + pq;
        double lgp = 0.0, ugp = 0.0, cfT = 0.0, zg = 0.0, yg = 0.0;
        bool preq = false;
        if (ownsRow) {
            const double gv = ws.g[(size_t)b * 8 * N + 8 * (N - 1) + prow];
            lgp = cfg.lbg[prow] - gv; ugp = cfg.ubg[prow] - gv;
            preq = ugp - lgp < 1e-4;
            cfT = Gkg[((N - 1) * 8 + prow) * 22 + 21];
        }
        const double rr_ = preq ? rho_eq : rho_in, rri = preq ? inv_eq : inv_in;
        const int o_j0 = L::oJ12 + (2 * prp + (pq & 1)) * XS + 6 * pq, o_j1 = L::oJ12 + (2 * prp + 1 - (pq & 1)) * XS + 6 * pq;
        const int o_xv12 = L::oXn + (N - 1) * XS + 6 * pq;
        const int o_gd12 = (isP12 && prp == 0) ? L::oGp + (N - 1) * XS + 6 * pq : L::oMisc + L::M_pad + 8;
        const int o_gdy12 = (isP12 && prp == 0) ? L::oGpy + (N - 1) * XS + 6 * pq : L::oMisc + L::M_pad + 8;
        // (A^T w)[v] of the dynamics rows (w: oWg or, at the tests, the duals) plus the path-row part gp[v]
        auto col_gather = [&](const double *w, const double *gp, int h) -> double {
            const double *cA = lds + L::oCD + o_cdA[h], *cB = lds + L::oCD + o_cdB[h];
            const double *wa = w + o_wA[h], *wb = w + o_wB[h];
            const double a0 = cA[0], a1 = cA[4], a2 = cA[8], b0 = cB[0], b1 = cB[4], b2 = cB[8];
            const double u0 = wa[0], u1 = wa[14], u2 = wa[28], q0 = wb[0], q1 = wb[14], q2 = wb[28];
            const double wf = w[o_wf[h]], g = gp[o_xpos[h]];
            return (g + v_cf[h] * wf) + ((a0 * u0 + a1 * u1) + (a2 * u2 + b0 * q0)) + (b1 * q1 + b2 * q2);
        };
        auto row_dot_dyn = [&](const double *xe) -> double {
            const double *cd = lds + L::oCD + o_dci, *x0 = xe + o_dx0;
            const double c0 = cd[0], c1 = cd[1], c2 = cd[2], c3 = cd[3];
            const double x_0 = x0[0], x_1 = x0[XS], x_2 = x0[2 * XS], x_3 = x0[3 * XS], xf_ = xe[o_dxf], xT_ = xe[21];
            return ((c0 * x_0 + c1 * x_1) + (c2 * x_2 + c3 * x_3)) + (d_cT * xT_ - tsT * xf_);
        };
        auto solve_b = [&](const bool first) {
            // ---- P1a: t = G b_J (segments 2, 3), rows 48, t_U ----
            {
                const double tq = g_prod(lds + o_bj);
                lds[o_tw] = tq; lds[o_tdw] = tq;
                if (bt >= 128) {                                          // wave 5: rows 48 (16 lanes)
                    const double *g48 = lds + o_g48, *b48 = lds + o_b48;
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < 7; j++) { const D2 gq = lds2(g48 + 2 * j), bq = lds2(b48 + 2 * j); acc += gq.x * bq.x; acc += gq.y * bq.y; }
                    lds[o_t48] = sum4(acc);
                } else if (bt >= 64) {                                    // wave 4: the U block (7 lanes)
                    double acc = 0.0;
#pragma unroll
                    for (int c = 0; c < 7; c++) acc += lds[o_gu + c] * lds[L::oRhsU + c];
                    lds[o_tuw] = acc; lds[o_tduw] = acc;
                }
            }
            __syncthreads();
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
            __syncthreads();
            ph_p3();
            __syncthreads();
            ph_p4a();
            __syncthreads();
            // ---- P4b: x~_J = G c - wbar x~_T; rows 48; x~_U ----
            {
                const double xT = misc[L::M_xtT];
                const double tq = g_prod(lds + o_bj + (L::oCJ - L::oRhsJ));
                lds[o_xw] = tq - wrow * xT;
                if (bt >= 128) {
                    const double *g48 = lds + o_g48, *c48 = lds + o_b48 + (L::oCJ - L::oRhsJ);
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < 7; j++) { const D2 gq = lds2(g48 + 2 * j), cq = lds2(c48 + 2 * j); acc += gq.x * cq.x; acc += gq.y * cq.y; }
                    lds[o_x48] = sum4(acc) - w48 * xT;
                } else if (bt >= 64) {
                    double acc = 0.0;
#pragma unroll
                    for (int c = 0; c < 7; c++) acc += lds[o_gu + c] * lds[L::oCJ + 56 * NSEG + c];
                    lds[o_xuw] = acc - v_wb[1] * xT;
                }
            }
            __syncthreads();
        };
        // border solve K_0 wbar = k (wbar = 0, x~_T = 0 so far), then delta = kappa + h_TT + sigma + rho_T - k^T wbar
        solve_b(true);
        wrow = gout ? lds[o_xw] : 0.0;
        s_wb = sOut ? lds[o_xiw] : 0.0;
        w48 = out48 ? lds[o_x48] : 0.0;
#pragma unroll
        for (int h = 0; h < 2; h++) v_wb[h] = (bt + L::NB * h < na) ? xn[o_xpos[h]] : 0.0;
        __syncthreads();
        bpl[bt] = v_kt[0] * v_wb[0];
        if (bt < 128) bpl[L::NB + bt] = v_kt[1] * v_wb[1];               // (zero beyond na)
        __syncthreads();
        if (bt >= 128) {
            const int ln = bt - 128;
            double sacc = ((bpl[ln] + bpl[ln + 64]) + (bpl[ln + 128] + bpl[ln + 192])) + bpl[ln + 256];
            sacc = wave_sum(sacc);
            if (ln == 0) {
                const double hdT = misc[L::M_sumha] + cfg.hess_reg;
                misc[L::M_hdT] = hdT;
                const double delta = (misc[L::M_kap] + (hdT + sigma + misc[L::M_rbT])) - sacc;
                misc[L::M_delta] = delta;
                if (!(delta > 0.0)) atomicOr(&ws.status[b], 2);
            }
        }
        for (it = 1; it <= cfg.qp_iters; it++) {
            // ---- A: rhs = sigma x - q + rho_b z_b - y_b + A^T w;  wbar_v rhs_v for the T border ----
            {
                const double r0 = (sigma * v_x[0] + ((v_req[0] ? rho_eq : rho_in) * v_zb[0] - v_yb[0])) + col_gather(wg, gpl, 0);
                lds[o_rhs[0]] = r0;
                bpl[bt] = v_wb[0] * r0;
                if (wave2) {
                    const double r1 = (sigma * v_x[1] + ((v_req[1] ? rho_eq : rho_in) * v_zb[1] - v_yb[1])) + col_gather(wg, gpl, 1);
                    lds[o_rhs[1]] = r1;
                    bpl[L::NB + bt] = v_wb[1] * r1;                     // (lanes without a second variable: wbar = 0, pad slot for rhs)
                }
            }
            __syncthreads();
            solve_b(false);
            // ---- E: dynamics rows, variables, T, path rows of node N - 1 ----
            {
                if (waveDyn) {
                    const double zt = row_dot_dyn(xn);
                    const double zr = alpha * zt + (1.0 - alpha) * d_z;
                    d_y += rho_eq * (zr - d_l);                          // the row is an equality: the projection of anything onto [l, l] is l
                    d_z = d_l;
                    const double w = rho_eq * d_l - d_y;
                    lds[o_wgw] = w; lds[o_tpw] = d_cT * w;
                }
                {
                    const double xtv = xn[o_xpos[0]];
                    v_x[0] = alpha * xtv + (1.0 - alpha) * v_x[0];
                    const double zr = alpha * xtv + (1.0 - alpha) * v_zb[0];
                    const double zn = clip(zr + v_yb[0] * (v_req[0] ? inv_eq : inv_in), v_lb[0], v_ub[0]);
                    v_yb[0] += (v_req[0] ? rho_eq : rho_in) * (zr - zn);
                    v_zb[0] = zn;
                }
                if (wave2) {
                    const double xtv = xn[o_xpos[1]];
                    v_x[1] = alpha * xtv + (1.0 - alpha) * v_x[1];
                    const double zr = alpha * xtv + (1.0 - alpha) * v_zb[1];
                    const double zn = clip(zr + v_yb[1] * (v_req[1] ? inv_eq : inv_in), v_lb[1], v_ub[1]);
                    v_yb[1] += (v_req[1] ? rho_eq : rho_in) * (zr - zn);
                    v_zb[1] = zn;
                }
                if (bt >= 128) {
                    D2 p0[3], p1[3];
#pragma unroll
                    for (int j = 0; j < 3; j++) { p0[j] = lds2(lds + o_j0 + 2 * j); p1[j] = lds2(lds + o_j1 + 2 * j); }
                    path_rows(p0, p1, lds + o_xv12, lds + o_gd12, [&](double zt) -> double {
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
                if (bt == 127) {        // the shared variable T (state in misc)
                    const double xtv = misc[L::M_xtT], rb = misc[L::M_rbT];
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
            __syncthreads();
            if (--until_check == 0) {
                until_check = cfg.check_every;
                // ---- termination test: r_prim = ||[A;I]x - z||inf, r_dual = ||Hx + q + [A;I]^T y||inf (oracle/ocp.c admm) ----
                double *xx = lds + L::oXx, *ys = lds + L::oYs, *gpy = lds + L::oGpy;
                const double xTc = misc[L::M_xT];
                xx[o_xpos[0]] = v_x[0];
                if (bt < NV2) xx[o_xpos[1]] = v_x[1];
                if (bt >= 128 && bt < 128 + N) xx[(bt - 128) * XS + 21] = xTc;
                if (bt >= 128 + 16 && bt < 128 + 16 + N) { xx[(bt - 144) * XS + 22] = 0.0; xx[(bt - 144) * XS + 23] = 0.0; }
                if (isDyn) ys[bt] = d_y;
                if (bt >= 168 && bt < 170) ys[bt] = 0.0;
                __syncthreads();
                double sums[2] = {isDyn ? d_cT * d_y : 0.0, 0.0};       // T row: sum coefT_r y_r, sum ha_i x_i
                double mx[6] = {0, 0, 0, 0, 0, 0};                       // rp, |Ax|, |z|, rd, |Hx|, |A^T y|
                double ha[2] = {0.0, 0.0};
#pragma unroll
                for (int h = 0; h < 2; h++) if (v_cf[h] != 0.0) { ha[h] = -ts * lam_rows[o_wf[h]]; sums[1] += ha[h] * v_x[h]; }
                if (isDyn) {
                    const double ax = row_dot_dyn(xx);
                    mx[0] = fabs(ax - d_z); mx[1] = fabs(ax); mx[2] = fabs(d_z);
                }
                if (bt >= 128) {
                    D2 p0[3], p1[3];
#pragma unroll
                    for (int j = 0; j < 3; j++) { p0[j] = lds2(lds + o_j0 + 2 * j); p1[j] = lds2(lds + o_j1 + 2 * j); }
                    const double ax = path_rows(p0, p1, lds + L::oXx + (N - 1) * XS + 6 * pq, lds + o_gdy12, [&](double) -> double { return ownsRow ? yg : 0.0; });
                    if (ownsRow) { sums[0] += cfT * yg; mx[0] = fmax(mx[0], fabs(ax - zg)); mx[1] = fmax(mx[1], fabs(ax)); mx[2] = fmax(mx[2], fabs(zg)); }
                }
                block_reduce_dpp<6, 2, false>(sums, red, tid);            // (its barriers publish gpy)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if (bt + L::NB * h < na) {
                        const double hx = (fabs(ha[h]) + cfg.hess_reg) * v_x[h] + ha[h] * xTc, aty = col_gather(ys, gpy, h) + v_yb[h];
                        mx[0] = fmax(mx[0], fabs(v_x[h] - v_zb[h])); mx[1] = fmax(mx[1], fabs(v_x[h])); mx[2] = fmax(mx[2], fabs(v_zb[h]));
                        mx[3] = fmax(mx[3], fabs(hx + aty)); mx[4] = fmax(mx[4], fabs(hx)); mx[5] = fmax(mx[5], fabs(aty));
                    }
                }
                if (bt == 127) {
                    const double zT = misc[L::M_zbT], yT = misc[L::M_ybT];
                    const double hxT = misc[L::M_hdT] * xTc + sums[1], atyT = sums[0] + yT;
                    mx[0] = fmax(mx[0], fabs(xTc - zT)); mx[1] = fmax(mx[1], fabs(xTc)); mx[2] = fmax(mx[2], fabs(zT));
                    mx[3] = fmax(mx[3], fabs(hxT + atyT + 1.0)); mx[4] = fmax(mx[4], fabs(hxT)); mx[5] = fmax(mx[5], fabs(atyT));
                }
                block_reduce_dpp<6, 6, true>(mx, red, tid);
                const double ep = cfg.eps_abs + cfg.eps_rel * fmax(mx[1], mx[2]);
                const double ed = cfg.eps_abs + cfg.eps_rel * fmax(fmax(mx[4], mx[5]), 1.0);      // ||q||_inf = 1
                done = (mx[0] <= ep && mx[3] <= ed) ? 1 : 0;
                __syncthreads();
                if (done) break;
            }
        }
        if (it > cfg.qp_iters) it = cfg.qp_iters;
        // ---------------- results ----------------
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int v = bt + L::NB * h;
            if (v < na) { ws.p[(size_t)b * n_tot + v] = v_x[h]; ws.y[(size_t)b * mn_tot + ma + v] = v_yb[h]; }
        }
        if (isDyn) ws.y[(size_t)b * mn_tot + bt] = d_y;
        if (ownsRow) ws.y[(size_t)b * mn_tot + meq + 8 * (N - 1) + prow] = yg;
        if (bt == 127) {
            ws.p[(size_t)b * n_tot + na] = misc[L::M_xT];
            ws.y[(size_t)b * mn_tot + mn_tot - 1] = misc[L::M_ybT];
            ws.qpit[b] = it; ws.qp_total[b] += it;
        }
    }
}

}  // namespace mpcmp
