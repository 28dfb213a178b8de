// kinematics_host.hpp — host-side scenario helpers of the reference's robot wrapper (robot_utils/pandaWrapper.cpp:14-107):
// tool-frame Jacobian, task <-> joint velocity maps and the damped least-squares inverse kinematics that
// examples/benchmark.cpp:19-42 uses to draw target states.  These run once per scenario on the host in the reference and
// do so here; the batched solve never calls them.
#pragma once
#include <cmath>
#include "../../include/mpcmp.h"

namespace mpcmp_host {

// forward kinematics of the tool frame; zax/org: joint axes and origins in the world frame
inline void fk_chain(const mpcmp_model &M, const double *q, double R[9], double p[3], double zax[7][3], double org[7][3]) {
    double Rw[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pw[3] = {0, 0, 0};
    for (int i = 0; i < 7; i++) {
        for (int r = 0; r < 3; r++) pw[r] += Rw[3 * r] * M.p[i][0] + Rw[3 * r + 1] * M.p[i][1] + Rw[3 * r + 2] * M.p[i][2];
        const double c = std::cos(q[i]), s = std::sin(q[i]);
        double Rj[9], Rn[9];
        for (int r = 0; r < 3; r++) {
            Rj[3 * r] = M.R0[i][3 * r] * c + M.R0[i][3 * r + 1] * s;
            Rj[3 * r + 1] = -M.R0[i][3 * r] * s + M.R0[i][3 * r + 1] * c;
            Rj[3 * r + 2] = M.R0[i][3 * r + 2];
        }
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) Rn[3 * r + k] = Rw[3 * r] * Rj[k] + Rw[3 * r + 1] * Rj[3 + k] + Rw[3 * r + 2] * Rj[6 + k];
        for (int k = 0; k < 9; k++) Rw[k] = Rn[k];
        for (int r = 0; r < 3; r++) { zax[i][r] = Rw[3 * r + 2]; org[i][r] = pw[r]; }
    }
    for (int r = 0; r < 3; r++) p[r] = pw[r] + Rw[3 * r] * M.tool[0] + Rw[3 * r + 1] * M.tool[1] + Rw[3 * r + 2] * M.tool[2];
    for (int k = 0; k < 9; k++) R[k] = Rw[k];
}

// world-aligned Jacobian of the tool frame, rows [linear(3); angular(3)], row-major 6x7: what the reference obtains by
// rotating Pinocchio's LOCAL frame Jacobian with blockdiag(R, R) (pandaWrapper.cpp:70-75,97-101)
inline void tool_jacobian(const mpcmp_model &M, const double *q, double J[42], double p[3], double R[9]) {
    double zax[7][3], org[7][3];
    fk_chain(M, q, R, p, zax, org);
    for (int i = 0; i < 7; i++) {
        const double d[3] = {p[0] - org[i][0], p[1] - org[i][1], p[2] - org[i][2]};
        J[0 * 7 + i] = zax[i][1] * d[2] - zax[i][2] * d[1];
        J[1 * 7 + i] = zax[i][2] * d[0] - zax[i][0] * d[2];
        J[2 * 7 + i] = zax[i][0] * d[1] - zax[i][1] * d[0];
        for (int r = 0; r < 3; r++) J[(3 + r) * 7 + i] = zax[i][r];
    }
}

// x = (J J^T + damp I)^-1 b for a 6x7 J (Cholesky of the 6x6 normal matrix; it is SPD for damp > 0)
inline bool solve_normal(const double J[42], double damp, const double b[6], double x[6]) {
    double A[36];
    for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) {
            double s = 0;
            for (int k = 0; k < 7; k++) s += J[r * 7 + k] * J[c * 7 + k];
            A[r * 6 + c] = s + (r == c ? damp : 0.0);
        }
    double L[36] = {0};
    for (int r = 0; r < 6; r++)
        for (int c = 0; c <= r; c++) {
            double s = A[r * 6 + c];
            for (int k = 0; k < c; k++) s -= L[r * 6 + k] * L[c * 6 + k];
            if (r == c) { if (!(s > 0.0)) return false; L[r * 6 + r] = std::sqrt(s); }
            else L[r * 6 + c] = s / L[c * 6 + c];
        }
    double y[6];
    for (int r = 0; r < 6; r++) { double s = b[r]; for (int k = 0; k < r; k++) s -= L[r * 6 + k] * y[k]; y[r] = s / L[r * 6 + r]; }
    for (int r = 5; r >= 0; r--) { double s = y[r]; for (int k = r + 1; k < 6; k++) s -= L[k * 6 + r] * x[k]; x[r] = s / L[r * 6 + r]; }
    return true;
}

// rotation vector of R (SO(3) logarithm)
inline void log3(const double R[9], double w[3], double *theta) {
    const double tr = R[0] + R[4] + R[8];
    double ct = 0.5 * (tr - 1.0);
    ct = ct > 1.0 ? 1.0 : (ct < -1.0 ? -1.0 : ct);
    const double th = std::acos(ct);
    const double vx = R[7] - R[5], vy = R[2] - R[6], vz = R[3] - R[1];
    *theta = th;
    if (th < 1e-8) { w[0] = 0.5 * vx; w[1] = 0.5 * vy; w[2] = 0.5 * vz; return; }
    if (M_PI - th > 1e-4) { const double f = th / (2.0 * std::sin(th)); w[0] = f * vx; w[1] = f * vy; w[2] = f * vz; return; }
    // near pi: the axis comes from the diagonal, its sign from the skew part
    const double d[3] = {R[0], R[4], R[8]};
    for (int k = 0; k < 3; k++) {
        const double a = std::sqrt(std::fmax(0.0, (d[k] - ct) / (1.0 - ct)));
        const double sgn = (k == 0 ? vx : (k == 1 ? vy : vz)) < 0.0 ? -1.0 : 1.0;
        w[k] = th * sgn * a;
    }
}

// se(3) logarithm of (R, p), output [v(3); w(3)] (the ordering of pinocchio::log6(...).toVector())
inline void log6(const double R[9], const double p[3], double out[6]) {
    double w[3], th;
    log3(R, w, &th);
    double alpha, beta;
    if (th < 1e-4) {
        const double t2 = th * th;
        alpha = 1.0 - t2 / 12.0 - t2 * t2 / 720.0;
        beta = 1.0 / 12.0 + t2 / 720.0 + t2 * t2 / 30240.0;
    } else {
        const double st = std::sin(th), ct = std::cos(th);
        alpha = th * st / (2.0 * (1.0 - ct));
        beta = (1.0 - alpha) / (th * th);
    }
    const double wp = w[0] * p[0] + w[1] * p[1] + w[2] * p[2];
    const double cx = w[1] * p[2] - w[2] * p[1], cy = w[2] * p[0] - w[0] * p[2], cz = w[0] * p[1] - w[1] * p[0];
    out[0] = alpha * p[0] - 0.5 * cx + beta * wp * w[0];
    out[1] = alpha * p[1] - 0.5 * cy + beta * wp * w[1];
    out[2] = alpha * p[2] - 0.5 * cz + beta * wp * w[2];
    out[3] = w[0]; out[4] = w[1]; out[5] = w[2];
}

}  // namespace mpcmp_host
