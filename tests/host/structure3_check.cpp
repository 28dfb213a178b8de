// Host-side check of the static structure tables of k_qp3f / k_qp3 (mpc_motion_planner_amd/csrc/structure3.hpp), compiled and
// run by tests/test_structure3_host.py.  Prints "ok <nseg>" per discretisation or a diagnostic and exits non-zero.
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#include "../../mpc_motion_planner_amd/csrc/structure3.hpp"

using namespace mpcmp;

static int fail(const char *what, int nseg, long a = 0, long b = 0) { std::printf("FAIL nseg %d: %s (%ld, %ld)\n", nseg, what, a, b); return 1; }

static int check(int nseg) {
    Tables3 T;
    if (!build_tables3(nseg, T)) return fail("build_tables3", nseg);
    const int N = 3 * nseg + 1, na = 21 * N, ma = 14 * (N - 1) + 8 * N, nJ = 49 * nseg, nI = 14 * (nseg + 1);
    // the internal order is a permutation of the arm's variables
    std::set<int> seen(T.ext_of_int.begin(), T.ext_of_int.end());
    if ((int)T.ext_of_int.size() != na || (int)seen.size() != na || *seen.begin() != 0 || *seen.rbegin() != na - 1) return fail("ext_of_int is not a permutation", nseg);
    for (int v = 0; v < na; v++) if (T.ext_of_int[int3_of_ext(nseg, v)] != v) return fail("int3_of_ext / ext_of_int disagree", nseg, v);
    // canonical slots of the sparse K_JC: [base, base + 14, base - 7, base + 7], 0xFF where there is no entry
    for (int r = 0; r < 49; r++) {
        const uint32_t w = T.pat.jc[r];
        const int b0 = w & 255, b1 = (w >> 8) & 255, b2 = (w >> 16) & 255, b3 = (w >> 24) & 255;
        if (b0 == 255 || b1 != b0 + 14) return fail("slots 0 / 1", nseg, r, w);
        if (b2 != 255 && b2 != b0 - 7) return fail("slot 2", nseg, r, w);
        if (b3 != 255 && b3 != b0 + 7) return fail("slot 3", nseg, r, w);
        for (int q = 0; q < 4; q++) {
            const int c = (w >> (8 * q)) & 255;
            if (c == 255) continue;
            const int d = r - c % 14;
            if (c >= 28 || d % 7 != 0 || d < -7 || d > 35) return fail("column form assumption", nseg, r, c);
        }
    }
    // every product A[r][a] A[r][b] (a >= b, positions in the row's nonzero list) is a term of exactly one entry
    const int E = (int)T.entry_ptr.size() - 1;
    long expect = 0;
    std::vector<int> vars;
    for (int r = 0; r < ma; r++) { row_vars(nseg, r, vars); expect += (long)vars.size() * ((long)vars.size() + 1) / 2; }
    if ((long)T.terms.size() != expect || T.entry_ptr[E] != (int)T.terms.size()) return fail("number of terms", nseg, (long)T.terms.size(), expect);
    std::set<uint32_t> uniq(T.terms.begin(), T.terms.end());
    if (uniq.size() != T.terms.size()) return fail("a term appears twice", nseg);
    // entry space: sizes as Dim3 says; kappa (the (T, T) entry) has one term per row; no other entry is that long
    const int eKap = nseg * 1225 + 28 + nseg * 196 + nseg * 98 + 98 + na, EA = eKap + 1, SP = nI * (nI + 1) / 2;
    if (E != EA + SP) return fail("entry count", nseg, E, EA + SP);
    // (every row but the N tool-height rows, which do not depend on T: robot_ocp.hpp:91)
    if (T.entry_ptr[eKap + 1] - T.entry_ptr[eKap] != ma - N) return fail("kappa term count", nseg, T.entry_ptr[eKap + 1] - T.entry_ptr[eKap], ma - N);
    int tmax = 0;
    for (int e = 0; e < E; e++) if (e != eKap) { const int c = T.entry_ptr[e + 1] - T.entry_ptr[e]; if (c > tmax) tmax = c; }
    if (tmax > 64) return fail("an entry other than kappa has a long term list (ELL table size)", nseg, tmax);
    // interior blocks are decoupled from each other: a term of a K_JJ entry belongs to a row that touches only that segment's nodes
    (void)nJ;
    std::printf("ok %d (entries %d, terms %ld, longest list but kappa %d)\n", nseg, E, (long)T.terms.size(), tmax);
    return 0;
}

int main() { return check(6) | check(8); }
