// structure.hpp — static structure of the collocation NLP / QP shared by host and device code.
//
// NLP variables (external order, the reference's layout motionPlanner.cpp:158-174):
//     z = [x_0 .. x_{N-1} | u_0 .. u_{N-1} | T],   x_k = [q_k; qd_k] (14),  u_k = qdd_k (7),  n = 21N+1
// General rows: 14 dynamics rows per collocated node k <= N-2 (row 14k+r), then 8 path rows per node
// (row meq + 8k + r).  Then the n box rows.
//
// Linear-solver (internal) order = nested dissection of the reduced KKT matrix
//     K = H + sigma I + diag(rho_box) + A^T diag(rho) A :
//   interior J_s (49) = [u_3s, x_3s+1, u_3s+1, x_3s+2, u_3s+2]     s = 0..NSEG-1   (mutually decoupled)
//   interface I (nI)  = [x_0, x_3, .., x_3NSEG, u_{N-1}, T]
//   interior s couples to the 29 interface entries C_s = [x_3s, x_3s+3, T].
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace mpcmp {

template <int NSEG>
struct Dim {
    static constexpr int N = 3 * NSEG + 1;
    static constexpr int n = 21 * N + 1;
    static constexpr int meq = 14 * (N - 1);
    static constexpr int min_ = 8 * N;
    static constexpr int m = meq + min_;
    static constexpr int mn = m + n;
    static constexpr int nJ = 49 * NSEG;
    static constexpr int nI = 14 * (NSEG + 1) + 8;
    static constexpr int SP = nI * (nI + 1) / 2;          // packed interface Schur complement
    static constexpr int JP = 49 * 50 / 2;                // packed interior block
    static constexpr int JC = 49 * 29;                    // interior x coupled-interface block
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int need = cmax(cmax(n, m), cmax(78 * NSEG, cmax(4 * nI, 22 * N)));
    static constexpr int NT = (need + 63) / 64 * 64;      // threads per workgroup (one problem)
    static constexpr int NW = NT / 64;
};

#if defined(__HIPCC__)
#define MPCMP_HD __host__ __device__
#else
#define MPCMP_HD
#endif

MPCMP_HD inline int packed(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// internal (solver) index of external variable v
MPCMP_HD inline int int_of_ext(int nseg, int v) {
    const int N = 3 * nseg + 1, nJ = 49 * nseg;
    if (v == 21 * N) return nJ + 14 * (nseg + 1) + 7;
    if (v < 14 * N) {
        const int k = v / 14, r = v % 14;
        if (k % 3 == 0) return nJ + 14 * (k / 3) + r;
        return 49 * (k / 3) + (k % 3 == 1 ? 7 : 28) + r;
    }
    const int u = v - 14 * N, k = u / 7, r = u % 7;
    if (k == N - 1) return nJ + 14 * (nseg + 1) + r;
    return 49 * (k / 3) + (k % 3 == 0 ? 0 : (k % 3 == 1 ? 21 : 42)) + r;
}

// Host-built tables (uploaded once per context).
struct StructureTables {
    int nseg = 0;
    std::vector<int> ext_of_int;        // n
    // assembly program: for every entry of [seg0: KJJ packed | KJC][seg1 ...][KII packed] the list of
    // (row, a, b) triples whose rho_row * val(row,a) * val(row,b) sum to that entry.
    std::vector<int> entry_ptr;         // E+1
    std::vector<uint32_t> terms;        // row<<16 | a<<8 | b
};

// canonical nonzero list of a general row: external variable ids in a fixed order
//   dynamics row (k, r): [X(3s+0,r), X(3s+1,r), X(3s+2,r), X(3s+3,r), fcol, T]
//   path row (k, q<7):   [x_k (14), u_k (7), T];   height row (k, 7): [q_k (7)]
inline void row_vars(int nseg, int row, std::vector<int> &vars) {
    const int N = 3 * nseg + 1, meq = 14 * (N - 1);
    vars.clear();
    if (row < meq) {
        const int k = row / 14, r = row % 14, s = k / 3;
        for (int j = 0; j < 4; j++) vars.push_back(14 * (3 * s + j) + r);
        vars.push_back(r < 7 ? 14 * k + 7 + r : 14 * N + 7 * k + (r - 7));
        vars.push_back(21 * N);
    } else {
        const int k = (row - meq) / 8, q = (row - meq) % 8;
        if (q < 7) {
            for (int c = 0; c < 14; c++) vars.push_back(14 * k + c);
            for (int c = 0; c < 7; c++) vars.push_back(14 * N + 7 * k + c);
            vars.push_back(21 * N);
        } else {
            for (int c = 0; c < 7; c++) vars.push_back(14 * k + c);
        }
    }
}

// location of K(vi, vj) in the blocked storage; returns global entry id
inline int entry_of(int nseg, int vi, int vj) {
    const int nJ = 49 * nseg, nI = 14 * (nseg + 1) + 8;
    const int JP = 1225, JC = 1421;
    int a = int_of_ext(nseg, vi), b = int_of_ext(nseg, vj);
    if (a < b) { int t = a; a = b; b = t; }               // a >= b
    if (a < nJ) {                                           // both interior
        const int s = a / 49;
        if (b / 49 != s) return -1;
        return s * (JP + JC) + packed(a % 49, b % 49);
    }
    if (b < nJ) {                                           // interior b, interface a
        const int s = b / 49, ia = a - nJ;
        int c;
        if (ia == nI - 1) c = 28;
        else { c = ia - 14 * s; if (c < 0 || c >= 28) return -1; }
        return s * (JP + JC) + JP + (b % 49) * 29 + c;
    }
    return nseg * (JP + JC) + packed(a - nJ, b - nJ);
}

inline bool build_tables(int nseg, StructureTables &T) {
    const int N = 3 * nseg + 1, n = 21 * N + 1, m = 14 * (N - 1) + 8 * N;
    const int nI = 14 * (nseg + 1) + 8;
    const int E = nseg * (1225 + 1421) + nI * (nI + 1) / 2;
    T.nseg = nseg;
    T.ext_of_int.assign(n, -1);
    for (int v = 0; v < n; v++) {
        const int i = int_of_ext(nseg, v);
        if (i < 0 || i >= n || T.ext_of_int[i] != -1) return false;
        T.ext_of_int[i] = v;
    }
    std::vector<std::vector<uint32_t>> lists(E);
    std::vector<int> vars;
    for (int r = 0; r < m; r++) {
        row_vars(nseg, r, vars);
        const int nz = (int)vars.size();
        for (int a = 0; a < nz; a++)
            for (int b = 0; b <= a; b++) {
                const int e = entry_of(nseg, vars[a], vars[b]);
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


// ------------------------------------------------------------------------------------------------
// Load-balanced assembly streams for k_qp2.  The entries of one assembly pass are distributed over the NT threads
// (longest-processing-time first) so that every thread evaluates about the same number of terms; thread t reads
// word w at words[off + w*NT + t] (coalesced).  Word formats:
//   term : bit31 = 0, bits 0-13 offA, bits 14-27 offB, bit 28 = equality row (rho_eq instead of rho)
//          value = rho * V[offA] * V[offB], V = [dynamics-row coefficients (meq x RS) | path Jacobians (8N x GS)]
//   store: bit31 = 1, bits 0-19 destination (doubles, relative to the factor area); flushes the accumulator
//   nop  : 0xFFFFFFFF
struct AsmStreams {
    static constexpr int MAXPASS = 8;
    int npass = 0;
    int off[MAXPASS] = {0}, W[MAXPASS] = {0};
    // entries with very many terms (the T-T diagonal: one term per general row) are split into chunks whose partial
    // sums go to scratch slots; the owner adds them up in fixed order afterwards (deterministic)
    int split_dst = -1, split_scr = 0, split_n = 0;
    std::vector<uint32_t> words;
};

inline bool build_streams(int nseg, int NT, int HS, int GS, int RS, AsmStreams &A) {
    const int N = 3 * nseg + 1, m = 14 * (N - 1) + 8 * N, meq = 14 * (N - 1);
    const int nI = 14 * (nseg + 1) + 8, SP = nI * (nI + 1) / 2, JP = 1225, JC = 1421;
    const int E = nseg * (JP + JC) + SP;
    std::vector<std::vector<uint32_t>> lists(E);
    std::vector<int> vars;
    for (int r = 0; r < m; r++) {
        row_vars(nseg, r, vars);
        const int nz = (int)vars.size();
        const int vbase = r < meq ? r * RS : meq * RS + (r - meq) * GS;
        for (int a = 0; a < nz; a++)
            for (int b = 0; b <= a; b++) {
                const int e = entry_of(nseg, vars[a], vars[b]);
                if (e < 0 || e >= E) return false;
                const uint32_t oa = vbase + a, ob = vbase + b;
                if (oa >= (1u << 14) || ob >= (1u << 14)) return false;
                lists[e].push_back(oa | (ob << 14) | ((r < meq ? 1u : 0u) << 28));
            }
    }
    // factor-area layout (doubles, relative): S | KJJ[nseg] | KJC[HS] | ...
    const int oKJJ = SP, oKJC = SP + nseg * JP;
    const int scr_base = SP + nseg * JP + 2 * HS * JC;      // scratch slots behind KJC[HS], Eh[HS]
    struct Ent { int e; uint32_t dst; };
    std::vector<std::vector<Ent>> passes;
    {   // pass 0: interface block + every interior diagonal block
        std::vector<Ent> p0;
        for (int i = 0; i < SP; i++) p0.push_back({nseg * (JP + JC) + i, (uint32_t)i});
        for (int s = 0; s < nseg; s++)
            for (int i = 0; i < JP; i++) p0.push_back({s * (JP + JC) + i, (uint32_t)(oKJJ + s * JP + i)});
        passes.push_back(p0);
    }
    for (int s0 = 0; s0 < nseg; s0 += HS) {   // one pass per segment group: coupling blocks
        std::vector<Ent> pg;
        for (int h = 0; h < HS && s0 + h < nseg; h++)
            for (int i = 0; i < JC; i++) pg.push_back({(s0 + h) * (JP + JC) + JP + i, (uint32_t)(oKJC + h * JC + i)});
        passes.push_back(pg);
    }
    if ((int)passes.size() > AsmStreams::MAXPASS) return false;
    A.npass = (int)passes.size();
    A.words.clear();
    for (int p = 0; p < A.npass; p++) {
        std::vector<Ent> ents = passes[p];
        std::stable_sort(ents.begin(), ents.end(), [&](const Ent &x, const Ent &y) { return lists[x.e].size() > lists[y.e].size(); });
        std::vector<std::vector<uint32_t>> th(NT);
        // LPT: next entry goes to the currently shortest stream (ties: lowest thread id) — deterministic
        auto shortest = [&]() { int best = 0; for (int t = 1; t < NT; t++) if (th[t].size() < th[best].size()) best = t; return best; };
        for (const Ent &en : ents) {
            const std::vector<uint32_t> &L = lists[en.e];
            if (L.size() > 48) {            // split (only ever the T-T entry)
                if (A.split_dst >= 0) return false;
                const int chunk = 16, nc = ((int)L.size() + chunk - 1) / chunk;
                A.split_dst = (int)en.dst; A.split_scr = scr_base; A.split_n = nc;
                for (int c = 0; c < nc; c++) {
                    const int t = shortest();
                    for (int i = c * chunk; i < (c + 1) * chunk && i < (int)L.size(); i++) th[t].push_back(L[i]);
                    th[t].push_back(0x80000000u | (uint32_t)(scr_base + c));
                }
                continue;
            }
            const int best = shortest();
            for (uint32_t w : L) th[best].push_back(w);
            th[best].push_back(0x80000000u | en.dst);
        }
        size_t W = 0;
        for (int t = 0; t < NT; t++) W = std::max(W, th[t].size());
        A.off[p] = (int)A.words.size(); A.W[p] = (int)W;
        A.words.resize(A.words.size() + W * NT, 0xFFFFFFFFu);
        for (int t = 0; t < NT; t++)
            for (size_t w = 0; w < th[t].size(); w++) A.words[A.off[p] + w * NT + t] = th[t][w];
    }
    return true;
}

}  // namespace mpcmp
