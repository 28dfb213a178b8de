"""The oracle under AddressSanitizer + UBSan (CPU build only; GPU sanitizers are not available on the pool)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = r'''
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
int main(void) {
    orc_model m; orc_default_model(&m);
    static const int segs[4] = {1, 2, 4, 6};
    for (int is = 0; is < 4; is++) {
        const int nseg = segs[is], N = orc_num_nodes(nseg);
        orc_config c; orc_default_config(&c, nseg, 2); orc_set_margins(&c, 0.9, 0.9, 0.5, 0.9); c.qp_iters = 60;
        double x0[14] = {-2.57762, 0.0202198, 0.522866, -1.41521, -0.636309, 3.04483, -1.01866, 0.525372, 1.60065, -1.82775, 1.9492, -0.681813, -0.809268, -0.655321};
        double xf[14] = {-2.0756, -0.278874, -0.748365, -0.979504, 1.33928, 2.31974, 0.613684, -0.757468, 0.896376, 0.901192, 0.389052, -0.124456, -0.918855, -1.1499};
        double *xg = calloc(14 * N, 8), *ug = calloc(7 * N, 8), *xs = calloc(14 * N, 8), *us = calloc(7 * N, 8), T, Tg;
        orc_info info;
        orc_warm_start(&c, c.ubu, x0, xf, xg, ug, &Tg);
        orc_solve(&m, &c, x0, xf, xg, ug, Tg, xs, us, &T, &info);
        double out[74], pt[28];
        orc_traj_stats(&m, nseg, xs, us, T, xf, 50, out);
        orc_mpc_point(&m, nseg, xs, us, T, 0.05, pt);
        orc_mpc_point(&m, nseg, xs, us, T, 99.0, pt);
        double *smp = calloc(29 * 51, 8); orc_sample(&m, nseg, xs, us, T, 50, smp);
        int n = 21 * N + 1, mm = 14 * (N - 1) + 8 * N;
        double *p = calloc(n, 8), *y = calloc(mm + n, 8);
        orc_debug_qp(&m, &c, x0, xf, xg, ug, Tg, 0, p, y);
        printf("nseg %d T %.6f status %d\n", nseg, T, info.status & 7);
        free(xg); free(ug); free(xs); free(us); free(smp); free(p); free(y);
    }
    /* threaded batch helper */
    {
        orc_config c; orc_default_config(&c, 2, 1); c.qp_iters = 25;
        const int N = orc_num_nodes(2), B = 5;
        double *x0 = calloc(14 * B, 8), *xf = calloc(14 * B, 8), *xg = calloc(14 * N * B, 8), *ug = calloc(7 * N * B, 8), *Tg = calloc(B, 8);
        double *xs = calloc(14 * N * B, 8), *us = calloc(7 * N * B, 8), *T = calloc(B, 8);
        orc_info *info = calloc(B, sizeof(orc_info));
        for (int b = 0; b < B; b++) { xf[14 * b] = 0.3 + 0.1 * b; x0[14 * b + 3] = xf[14 * b + 3] = -1.5; x0[14 * b + 5] = xf[14 * b + 5] = 1.5;
            orc_warm_start(&c, c.ubu, x0 + 14 * b, xf + 14 * b, xg + 14 * N * b, ug + 7 * N * b, Tg + b); }
        orc_solve_batch(&m, &c, B, x0, xf, xg, ug, Tg, xs, us, T, info, 3);
        printf("batch T0 %.6f\n", T[0]);
        free(x0); free(xf); free(xg); free(ug); free(Tg); free(xs); free(us); free(T); free(info);
    }
    return 0;
}
'''


def test_oracle_asan_ubsan(tmp_path):
    src = tmp_path / "san.c"; src.write_text(PROG)
    exe = str(tmp_path / "san")
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-std=c99", "-I" + os.path.join(ROOT, "oracle"), "-o", exe, str(src),
                           os.path.join(ROOT, "oracle", "rbd.c"), os.path.join(ROOT, "oracle", "ocp.c"), "-lm", "-lpthread"])
    r = subprocess.run([exe], capture_output=True, text=True, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
    assert r.stdout.count("status 0") == 4
