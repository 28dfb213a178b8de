// structure3.hpp — static structure of ONE ARM's share of the QP for k_qp3 (qp_kernel_v3.hpp), shared by host and device.
//
// The reduced KKT matrix K = H + sigma I + diag(rho_box) + A^T diag(rho) A of an OCP with NARM independent arms is
//     [ K_0^(0)              k^(0) ]
//     [          K_0^(1)     k^(1) ]          the arms couple ONLY through the final time T (last row / column),
//     [ k^(0)T   k^(1)T      kappa ]
// so T is bordered out and every arm's K_0 is factorised by its own workgroup:
//     x_T = (b_T - sum_a w_a^T b_a) / delta,   x_a = K_0a^-1 b_a - w_a x_T,   w_a = K_0a^-1 k_a,   delta = kappa - sum_a k_a^T w_a.
// One arm's variables (external arm order: xs [N][14] | us [N][7], na = 21 N) in solver (internal) order:
//     interior  J_s (49) = [u_3s, x_3s+1, u_3s+1, x_3s+2, u_3s+2]        s = 0..NSEG-1     (mutually decoupled)
//     U         (7)      = u_{N-1}                                        (couples to x_{N-1} only)
//     interface I (nI)   = [x_0, x_3, .., x_3NSEG]                        (block tridiagonal, 14 x 14 blocks)
// J_s couples to the 28 interface entries C_s = [x_3s, x_3s+3]: a dense 7 x 14 block (u_3s x x_3s, the path rows of node 3s)
// plus at most four further entries per row (dynamics rows), stored sparsely — the solve never needs E_s = G_s K_JC explicitly:
//     t = G b_J;   r_I = b_I - K_CJ t;   y_I = S^-1 r_I;   x_J = G (b_J - K_JC y_I).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>
#include "structure.hpp"

namespace mpcmp {

template <int NSEG>
struct Dim3 {
    static constexpr int N = 3 * NSEG + 1;
    static constexpr int na = 21 * N;                       // variables of one arm (T excluded)
    static constexpr int meq = 14 * (N - 1), min_ = 8 * N, ma = meq + min_;
    static constexpr int nJ = 49 * NSEG, nI = 14 * (NSEG + 1);
    static constexpr int JP = 1225, UP = 28, SP = nI * (nI + 1) / 2;
    // assembled-entry index space of one arm: pass A (everything but K_II), then pass B (K_II packed)
    static constexpr int eKJJ = 0, eKUU = eKJJ + NSEG * JP, eKJC = eKUU + UP, eKUX = eKJC + NSEG * 196,
                         eKuX = eKUX + NSEG * 98, eKT = eKuX + 98, eKap = eKT + na, EA = eKap + 1, eS = EA, E = EA + SP;
};

// internal (solver) index of external arm variable v in [0, na)
MPCMP_HD inline int int3_of_ext(int nseg, int v) {
    const int N = 3 * nseg + 1, nJ = 49 * nseg;
    if (v < 14 * N) {
        const int k = v / 14, r = v % 14;
        if (k % 3 == 0) return nJ + 7 + 14 * (k / 3) + r;
        return 49 * (k / 3) + (k % 3 == 1 ? 7 : 28) + r;
    }
    const int u = v - 14 * N, k = u / 7, r = u % 7;
    if (k == N - 1) return nJ + r;
    return 49 * (k / 3) + (k % 3 == 0 ? 0 : (k % 3 == 1 ? 21 : 42)) + r;
}

// Sparse pattern of K_JC (identical for every segment), byte-packed:
//   jc[r]      : C-columns (0..27) of the up to four sparse entries of interior row r (0xFF = unused slot)
//   cjl/cjh[c] : up to eight (row << 2 | slot) references of C-column c (0xFF = end) — the transposed view
struct Qp3Pat {
    uint32_t jc[49];
    uint32_t cjl[28], cjh[28];
};

struct Tables3 {
    int nseg = 0;
    Qp3Pat pat;
    std::vector<int> ext_of_int;        // na
    std::vector<int> entry_ptr;         // E + 1
    std::vector<uint32_t> terms;        // row << 16 | a << 8 | b   (positions in the row's canonical nonzero list, structure.hpp row_vars)
};

inline bool build_tables3(int nseg, Tables3 &T) {
    const int N = 3 * nseg + 1, na = 21 * N, ma = 14 * (N - 1) + 8 * N, nJ = 49 * nseg, nI = 14 * (nseg + 1);
    const int eKJJ = 0, eKUU = eKJJ + nseg * 1225, eKJC = eKUU + 28, eKUX = eKJC + nseg * 196, eKuX = eKUX + nseg * 98,
              eKT = eKuX + 98, eKap = eKT + na, EA = eKap + 1, eS = EA, E = EA + nI * (nI + 1) / 2;
    T.nseg = nseg;
    T.ext_of_int.assign(na, -1);
    for (int v = 0; v < na; v++) {
        const int i = int3_of_ext(nseg, v);
        if (i < 0 || i >= na || T.ext_of_int[i] != -1) return false;
        T.ext_of_int[i] = v;
    }
    // classify a pair of arm variables (internal indices a >= b): 0 both J, 1 J-I, ...
    std::vector<std::vector<int>> slots(49);                  // sparse C-columns per interior row (from segment 0)
    std::vector<int> vars;
    auto cidx = [&](int s, int ia) -> int {                    // C-column of interface index ia (relative to I) for segment s, or -1
        const int ni = ia / 14, c = ia % 14;
        if (ni == s) return c;
        if (ni == s + 1) return 14 + c;
        return -1;
    };
    // pass 1: sparse pattern
    for (int pass = 0; pass < 2; pass++) {
        for (int r = 0; r < ma; r++) {
            row_vars(nseg, r, vars);
            for (size_t x = 0; x < vars.size(); x++)
                for (size_t y = 0; y < vars.size(); y++) {
                    if (vars[x] == na || vars[y] == na) continue;
                    const int a = int3_of_ext(nseg, vars[x]), b = int3_of_ext(nseg, vars[y]);
                    if (!(a < nJ && b >= nJ + 7)) continue;
                    const int s = a / 49, lr = a % 49, c = cidx(s, b - nJ - 7);
                    if (c < 0) return false;
                    if (lr < 7 && c < 14) continue;            // dense u_3s x x_3s block
                    bool found = false;
                    for (int q : slots[lr]) found |= (q == c);
                    if (pass == 0 && s == 0 && !found) slots[lr].push_back(c);
                    if (pass == 1 && !found) return false;     // a later segment has an entry outside segment 0's pattern
                }
        }
    }
    // canonical slot order: [base, base + 14, base - 7, base + 7] (-1: no such entry), so that the loop kernel addresses a row's
    // C-entries from ONE lane-dependent base plus immediate offsets.  u_3s rows (lr < 7) keep their x_3s entry in the dense
    // block: slot 0 names it as a placeholder (no term is ever assembled there, the coefficient stays 0).
    for (int lr = 0; lr < 49; lr++) {
        std::vector<int> v = slots[lr];
        std::sort(v.begin(), v.end());
        std::vector<int> canon;
        if (v.size() == 1 && lr < 7 && v[0] >= 14) canon = {v[0] - 14, v[0]};                                           // u_3s
        else if (v.size() == 2 && v[1] == v[0] + 14) canon = {v[0], v[1]};                                              // u_3s+1, u_3s+2
        else if (v.size() == 3 && v[1] == v[0] + 7 && v[2] == v[0] + 14) canon = {v[0], v[2], -1, v[1]};                // positions
        else if (v.size() == 4 && v[1] == v[0] + 7 && v[2] == v[0] + 14 && v[3] == v[0] + 21) canon = {v[1], v[3], v[0], v[2]};   // velocities
        else return false;
        slots[lr] = canon;
        // the column form (kcj) of the loop kernel assumes: an entry of C-column c sits in one of the rows c % 14 + 7 d, d = -1..5
        for (int col : canon) {
            if (col < 0) continue;
            const int d = lr - col % 14;
            if (d % 7 != 0 || d < -7 || d > 35) return false;
        }
    }
    for (int lr = 0; lr < 49; lr++) {
        if (slots[lr].size() > 4) return false;
        uint32_t w = 0xFFFFFFFFu;
        for (size_t q = 0; q < slots[lr].size(); q++) if (slots[lr][q] >= 0) w = (w & ~(0xFFu << (8 * q))) | ((uint32_t)slots[lr][q] << (8 * q));
        T.pat.jc[lr] = w;
    }
    for (int c = 0; c < 28; c++) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0xFFFFFFFFu;
        int cnt = 0;
        for (int lr = 0; lr < 49; lr++)
            for (size_t q = 0; q < slots[lr].size(); q++)
                if (slots[lr][q] == c) {
                    if (cnt >= 8) return false;
                    const uint32_t ref = ((uint32_t)lr << 2) | (uint32_t)q;
                    if (cnt < 4) lo = (lo & ~(0xFFu << (8 * cnt))) | (ref << (8 * cnt));
                    else hi = (hi & ~(0xFFu << (8 * (cnt - 4)))) | (ref << (8 * (cnt - 4)));
                    cnt++;
                }
        T.pat.cjl[c] = lo; T.pat.cjh[c] = hi;
    }
    // pass 2: entry lists
    auto entry_of = [&](int vi, int vj) -> int {
        if (vi == na && vj == na) return eKap;
        if (vi == na || vj == na) return eKT + int3_of_ext(nseg, vi == na ? vj : vi);
        int a = int3_of_ext(nseg, vi), b = int3_of_ext(nseg, vj);
        if (a < b) { const int t = a; a = b; b = t; }         // a >= b
        if (a < nJ) {                                          // both interior
            if (a / 49 != b / 49) return -1;
            return eKJJ + (a / 49) * 1225 + packed(a % 49, b % 49);
        }
        if (a < nJ + 7) {                                      // a in U
            if (b < nJ) return -1;
            return eKUU + packed(a - nJ, b - nJ);
        }
        const int ia = a - nJ - 7;                             // a in I
        if (b >= nJ + 7) return eS + packed(ia, b - nJ - 7);
        if (b >= nJ) {                                         // U x I: only x_{N-1}
            if (ia / 14 != nseg) return -1;
            return eKuX + (b - nJ) * 14 + ia % 14;
        }
        const int s = b / 49, lr = b % 49, c = cidx(s, ia);
        if (c < 0) return -1;
        if (lr < 7 && c < 14) return eKUX + s * 98 + lr * 14 + c;
        for (size_t q = 0; q < slots[lr].size(); q++)
            if (slots[lr][q] == c) return eKJC + s * 196 + lr * 4 + (int)q;
        return -1;
    };
    std::vector<std::vector<uint32_t>> lists(E);
    for (int r = 0; r < ma; r++) {
        row_vars(nseg, r, vars);
        const int nz = (int)vars.size();
        for (int a = 0; a < nz; a++)
            for (int b = 0; b <= a; b++) {
                const int e = entry_of(vars[a], vars[b]);
                if (e < 0 || e >= E) return false;
                lists[e].push_back(((uint32_t)r << 16) | ((uint32_t)a << 8) | (uint32_t)b);
            }
    }
    T.entry_ptr.assign(E + 1, 0);
    T.terms.clear();
    for (int e = 0; e < E; e++) {
        T.entry_ptr[e] = (int)T.terms.size();
        T.terms.insert(T.terms.end(), lists[e].begin(), lists[e].end());
    }
    T.entry_ptr[E] = (int)T.terms.size();
    return true;
}

}  // namespace mpcmp
