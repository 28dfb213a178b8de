// rbd_device.hpp — rigid-body layer for gfx950: RNEA with forward-mode analytic derivatives, FK and the
// tool-height Jacobian row for a 7-joint revolute-z chain.  Replaces the Pinocchio calls on the reference's
// hot path (robot_ocp.hpp:87-91,118-122,145-155; motionPlanner.hpp:92,111,127,141).
//
// One thread evaluates one (collocation node, derivative direction) pair: the Newton-Euler recursion and
// its tangent are fused in a single fully unrolled forward/backward sweep whose per-joint spatial
// quantities (F,N and their tangents) stay in VGPRs; sin/cos of the joint angles are staged once per node in
// LDS and shared by the 22 threads of the node.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mpcmp.h"

namespace mpcmp {

struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// s * (z_hat x v)
__device__ __forceinline__ V3 zcross(V3 v, double s) { return mk(-s * v.y, s * v.x, 0.0); }

struct M3 {
    double a[9];
};
__device__ __forceinline__ V3 mul(const M3 &R, V3 v) {
    return mk(R.a[0] * v.x + R.a[1] * v.y + R.a[2] * v.z, R.a[3] * v.x + R.a[4] * v.y + R.a[5] * v.z,
              R.a[6] * v.x + R.a[7] * v.y + R.a[8] * v.z);
}
__device__ __forceinline__ V3 mulT(const M3 &R, V3 v) {
    return mk(R.a[0] * v.x + R.a[3] * v.y + R.a[6] * v.z, R.a[1] * v.x + R.a[4] * v.y + R.a[7] * v.z,
              R.a[2] * v.x + R.a[5] * v.y + R.a[8] * v.z);
}
__device__ __forceinline__ V3 mulI(const double *I, V3 v) {
    return mk(I[0] * v.x + I[1] * v.y + I[2] * v.z, I[3] * v.x + I[4] * v.y + I[5] * v.z,
              I[6] * v.x + I[7] * v.y + I[8] * v.z);
}
// R_i(q) = R0_i * Rz(q), with (s,c) = (sin q, cos q)
__device__ __forceinline__ M3 joint_rot(const double *A, double s, double c) {
    M3 R;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        R.a[3 * r + 0] = A[3 * r + 0] * c + A[3 * r + 1] * s;
        R.a[3 * r + 1] = -A[3 * r + 0] * s + A[3 * r + 1] * c;
        R.a[3 * r + 2] = A[3 * r + 2];
    }
    return R;
}

// tau = RNEA(q,v,a); if TANGENT also dtau = d tau / d theta with theta = (q|v|a)_j  (type 0|1|2).
// sc: [7][2] = sin, cos of the joint angles.
// tw (optional, TANGENT only): lane-private LDS area for the tangent wrenches dF, dN (42 doubles, element e of this
// lane at tw[e * tws]); keeps the per-thread VGPR footprint below the spill threshold in the 320-thread kernels.
//
// RELOAD: the model constants of a joint are read (scalar loads) through an opaque copy of the model pointer right where the joint is
// processed.  Without it the compiler loads all ~180 constants of the chain once per kernel, ahead of everything, runs out of SGPRs and parks
// them in VGPR lanes: k_step<4> had 3,216 v_readlane_b32 for 4,600 FP64 instructions (round 5, from the ISA).  The model must be readable
// through the constant address space (a kernel argument by value, or global memory nobody writes during the kernel).
typedef const __attribute__((address_space(4))) mpcmp_model *cmodel_t;
struct JointK { double R0[9], p[3], com[3], I[9], mass; };
template <bool RELOAD, bool FWD>
__device__ __forceinline__ void joint_consts(const mpcmp_model *__restrict__ M, int i, int ic, JointK &k) {      // (R0, p, I, mass of joint i; com of joint ic)
    if (RELOAD) {
        const mpcmp_model *Mo = M;
        asm volatile("" : "+s"(Mo));
        cmodel_t Mc = (cmodel_t)Mo;
#pragma unroll
        for (int q = 0; q < 9; q++) k.R0[q] = Mc->R0[i][q];
#pragma unroll
        for (int q = 0; q < 3; q++) { k.p[q] = Mc->p[i][q]; k.com[q] = Mc->com[ic][q]; }
        if (FWD) {
#pragma unroll
            for (int q = 0; q < 9; q++) k.I[q] = Mc->I[i][q];
            k.mass = Mc->mass[i];
        }
    } else {
#pragma unroll
        for (int q = 0; q < 9; q++) k.R0[q] = M->R0[i][q];
#pragma unroll
        for (int q = 0; q < 3; q++) { k.p[q] = M->p[i][q]; k.com[q] = M->com[ic][q]; }
        if (FWD) {
#pragma unroll
            for (int q = 0; q < 9; q++) k.I[q] = M->I[i][q];
            k.mass = M->mass[i];
        }
    }
}
template <bool TANGENT, bool TW_LDS = false, bool RELOAD = false>
__device__ __forceinline__ void rnea_dir(const mpcmp_model *__restrict__ M, const double *sc, const double *v,
                                         const double *a, int type, int j, double *tau, double *dtau,
                                         double *tw = nullptr, int tws = 0) {
    V3 F[7], N[7], dF[TW_LDS ? 1 : 7], dN[TW_LDS ? 1 : 7];
    V3 w = mk(0, 0, 0), wd = mk(0, 0, 0), al = mk(-M->gravity[0], -M->gravity[1], -M->gravity[2]);
    V3 dw = mk(0, 0, 0), dwd = mk(0, 0, 0), dal = mk(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 7; i++) {
        JointK K;
        joint_consts<RELOAD, true>(M, i, i, K);
        const M3 R = joint_rot(K.R0, sc[2 * i], sc[2 * i + 1]);
        const V3 p = ld3(K.p), c = ld3(K.com);
        const double vi = v[i], ai = a[i];
        const V3 u = mulT(R, w), ud = mulT(R, wd);
        const V3 b = al + cross(wd, p) + cross(w, cross(w, p));
        const V3 wn = mk(u.x, u.y, u.z + vi);
        const V3 wdn = mk(ud.x + u.y * vi, ud.y - u.x * vi, ud.z + ai);
        const V3 aln = mulT(R, b);
        const V3 wxc = cross(wn, c);
        const V3 ac = aln + cross(wdn, c) + cross(wn, wxc);
        const V3 Iw = mulI(K.I, wn);
        F[i] = K.mass * ac;
        N[i] = mulI(K.I, wdn) + cross(wn, Iw);
        if (TANGENT) {
            const double dq = (type == 0 && i == j) ? 1.0 : 0.0;
            const double dv = (type == 1 && i == j) ? 1.0 : 0.0;
            const double da = (type == 2 && i == j) ? 1.0 : 0.0;
            const V3 du = mulT(R, dw) - zcross(u, dq);
            const V3 dud = mulT(R, dwd) - zcross(ud, dq);
            const V3 db = dal + cross(dwd, p) + cross(dw, cross(w, p)) + cross(w, cross(dw, p));
            const V3 dwn = mk(du.x, du.y, du.z + dv);
            const V3 dwdn = mk(dud.x + du.y * vi + u.y * dv, dud.y - du.x * vi - u.x * dv, dud.z + da);
            const V3 daln = mulT(R, db) - zcross(aln, dq);
            const V3 dac = daln + cross(dwdn, c) + cross(dwn, wxc) + cross(wn, cross(dwn, c));
            const V3 dFi = K.mass * dac;
            const V3 dNi = mulI(K.I, dwdn) + cross(dwn, Iw) + cross(wn, mulI(K.I, dwn));
            if (TW_LDS) {
                tw[(6 * i + 0) * tws] = dFi.x; tw[(6 * i + 1) * tws] = dFi.y; tw[(6 * i + 2) * tws] = dFi.z;
                tw[(6 * i + 3) * tws] = dNi.x; tw[(6 * i + 4) * tws] = dNi.y; tw[(6 * i + 5) * tws] = dNi.z;
            } else { dF[i] = dFi; dN[i] = dNi; }
            dw = dwn; dwd = dwdn; dal = daln;
        }
        w = wn; wd = wdn; al = aln;
        __builtin_amdgcn_sched_barrier(0);      // keep each joint's model-constant loads next to their uses
    }
    V3 f = mk(0, 0, 0), n = mk(0, 0, 0), df = mk(0, 0, 0), dn = mk(0, 0, 0);
#pragma unroll
    for (int i = 6; i >= 0; i--) {
        JointK K;
        joint_consts<RELOAD, false>(M, i < 6 ? i + 1 : 6, i, K);      // R0, p of joint i + 1 (i = 6: unused), com of joint i
        const V3 c = ld3(K.com);
        V3 fi = F[i], ni = N[i] + cross(c, F[i]);
        V3 dfi = mk(0, 0, 0), dni = mk(0, 0, 0);
        if (TANGENT) {
            V3 dFi, dNi;
            if (TW_LDS) {
                dFi = mk(tw[(6 * i + 0) * tws], tw[(6 * i + 1) * tws], tw[(6 * i + 2) * tws]);
                dNi = mk(tw[(6 * i + 3) * tws], tw[(6 * i + 4) * tws], tw[(6 * i + 5) * tws]);
            } else { dFi = dF[i]; dNi = dN[i]; }
            dfi = dFi; dni = dNi + cross(c, dFi);
        }
        if (i < 6) {
            const M3 R = joint_rot(K.R0, sc[2 * (i + 1)], sc[2 * (i + 1) + 1]);
            const V3 p = ld3(K.p);
            const V3 gf = mul(R, f), gn = mul(R, n);
            fi = fi + gf; ni = ni + gn + cross(p, gf);
            if (TANGENT) {
                const double dq = (type == 0 && i + 1 == j) ? 1.0 : 0.0;
                const V3 dgf = mul(R, df + zcross(f, dq)), dgn = mul(R, dn + zcross(n, dq));
                dfi = dfi + dgf; dni = dni + dgn + cross(p, dgf);
            }
        }
        f = fi; n = ni; df = dfi; dn = dni;
        tau[i] = n.z;
        if (TANGENT) dtau[i] = dn.z;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Column j of the joint-space inertia matrix, d tau / d a_j: rnea_dir's tangent for type 2, where the tangent angular velocity is zero and the
// nominal motion (w, wd, al, F, N) drops out of every term — a quarter of the generic recursion's instructions.  Every term left out is an exact
// zero in rnea_dir<true> (products with dq = dv = 0 and dw = 0), so the two agree to the sign of zeros (finite inputs).
template <bool TW_LDS = false, bool RELOAD = false>
__device__ __forceinline__ void rnea_mcol(const mpcmp_model *__restrict__ M, const double *sc, int j, double *dtau, double *tw = nullptr, int tws = 0) {
    V3 dF[TW_LDS ? 1 : 7], dN[TW_LDS ? 1 : 7];
    V3 dwd = mk(0, 0, 0), dal = mk(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 7; i++) {
        JointK K;
        joint_consts<RELOAD, true>(M, i, i, K);
        const M3 R = joint_rot(K.R0, sc[2 * i], sc[2 * i + 1]);
        const V3 p = ld3(K.p), c = ld3(K.com);
        const double da = (i == j) ? 1.0 : 0.0;
        const V3 dud = mulT(R, dwd);
        const V3 db = dal + cross(dwd, p);
        const V3 dwdn = mk(dud.x, dud.y, dud.z + da);
        const V3 daln = mulT(R, db);
        const V3 dac = daln + cross(dwdn, c);
        const V3 dFi = K.mass * dac;
        const V3 dNi = mulI(K.I, dwdn);
        if (TW_LDS) {
            tw[(6 * i + 0) * tws] = dFi.x; tw[(6 * i + 1) * tws] = dFi.y; tw[(6 * i + 2) * tws] = dFi.z;
            tw[(6 * i + 3) * tws] = dNi.x; tw[(6 * i + 4) * tws] = dNi.y; tw[(6 * i + 5) * tws] = dNi.z;
        } else { dF[i] = dFi; dN[i] = dNi; }
        dwd = dwdn; dal = daln;
        __builtin_amdgcn_sched_barrier(0);
    }
    V3 df = mk(0, 0, 0), dn = mk(0, 0, 0);
#pragma unroll
    for (int i = 6; i >= 0; i--) {
        JointK K;
        joint_consts<RELOAD, false>(M, i < 6 ? i + 1 : 6, i, K);
        const V3 c = ld3(K.com);
        V3 dFi, dNi;
        if (TW_LDS) {
            dFi = mk(tw[(6 * i + 0) * tws], tw[(6 * i + 1) * tws], tw[(6 * i + 2) * tws]);
            dNi = mk(tw[(6 * i + 3) * tws], tw[(6 * i + 4) * tws], tw[(6 * i + 5) * tws]);
        } else { dFi = dF[i]; dNi = dN[i]; }
        V3 dfi = dFi, dni = dNi + cross(c, dFi);
        if (i < 6) {
            const M3 R = joint_rot(K.R0, sc[2 * (i + 1)], sc[2 * (i + 1) + 1]);
            const V3 p = ld3(K.p);
            const V3 dgf = mul(R, df), dgn = mul(R, dn);
            dfi = dfi + dgf; dni = dni + dgn + cross(p, dgf);
        }
        df = dfi; dn = dni;
        dtau[i] = dn.z;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Forward kinematics of the chain: tool position and the z-row of the world-aligned tool Jacobian
// (robot_ocp.hpp:145-160: J = blockdiag(R,R) * J_local, row 2).  Also returns joint-7 origin height
// (MotionPlanner::sample_random_state rejects on oMi[7].z, motionPlanner.cpp:111) and link8 position.
__device__ __forceinline__ void fk_tool(const mpcmp_model *__restrict__ M, const double *sc, V3 *p_tool, double *Jz,
                                        V3 *p_joint7, V3 *p_link8) {
    M3 Rw; V3 pw = mk(0, 0, 0);
#pragma unroll
    for (int k = 0; k < 9; k++) Rw.a[k] = (k % 4 == 0) ? 1.0 : 0.0;
    V3 zax[7], org[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const M3 R = joint_rot(M->R0[i], sc[2 * i], sc[2 * i + 1]);
        pw = pw + mul(Rw, ld3(M->p[i]));
        M3 Rn;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
                Rn.a[3 * r + c] = Rw.a[3 * r + 0] * R.a[0 + c] + Rw.a[3 * r + 1] * R.a[3 + c] + Rw.a[3 * r + 2] * R.a[6 + c];
        Rw = Rn;
        zax[i] = mk(Rw.a[2], Rw.a[5], Rw.a[8]);
        org[i] = pw;
    }
    const V3 pt = pw + mul(Rw, ld3(M->tool));
    *p_tool = pt;
    if (p_joint7) *p_joint7 = pw;
    if (p_link8) *p_link8 = pw + mul(Rw, ld3(M->link8));
    if (Jz) {
#pragma unroll
        for (int i = 0; i < 7; i++) Jz[i] = cross(zax[i], pt - org[i]).z;
    }
}


// World-aligned velocity of the tool frame, J(q) qd with J = blockdiag(R,R) J_local (PandaWrapper::forward_velocities,
// robot_utils/pandaWrapper.cpp:90-107), and the tool position (table-collision check, examples/benchmark.cpp:106-113).
__device__ __forceinline__ void fk_task_velocity(const mpcmp_model *__restrict__ M, const double *sc, const double *qd,
                                                 V3 *p_tool, V3 *vlin, V3 *vang) {
    M3 Rw; V3 pw = mk(0, 0, 0);
#pragma unroll
    for (int k = 0; k < 9; k++) Rw.a[k] = (k % 4 == 0) ? 1.0 : 0.0;
    V3 zax[7], org[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const M3 R = joint_rot(M->R0[i], sc[2 * i], sc[2 * i + 1]);
        pw = pw + mul(Rw, ld3(M->p[i]));
        M3 Rn;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
                Rn.a[3 * r + c] = Rw.a[3 * r + 0] * R.a[0 + c] + Rw.a[3 * r + 1] * R.a[3 + c] + Rw.a[3 * r + 2] * R.a[6 + c];
        Rw = Rn;
        zax[i] = mk(Rw.a[2], Rw.a[5], Rw.a[8]);
        org[i] = pw;
    }
    const V3 pt = pw + mul(Rw, ld3(M->tool));
    V3 vl = mk(0, 0, 0), va = mk(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 7; i++) { vl = vl + qd[i] * cross(zax[i], pt - org[i]); va = va + qd[i] * zax[i]; }
    *p_tool = pt; *vlin = vl; *vang = va;
}

}  // namespace mpcmp
