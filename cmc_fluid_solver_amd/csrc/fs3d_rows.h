// Device-side construction of one tridiagonal row per cell -- the dense restatement
// of AdiSolver3D::ApplyBC0/ApplyBC1/BuildMatrix (AdiSolver3D.cpp:732-852).
//
// The reference solves each segment separately (SolveSegment, :687-705).  Here a
// whole grid line is one dense recurrence in which
//   START rows have a = 0, END rows have c = 0, SKIP rows are identity and never written,
// so the Thomas recurrence decouples at exactly the places where the reference
// starts/ends a segment, and every floating-point operation on a segment's cells is
// the one the reference performs, in the same order (no FMA contraction: the file is
// compiled with -ffp-contract=off; divisions are IEEE-correct).
#pragma once
#include "fs3d_common.h"

template <typename R>
struct RowUVWT {
    // U,V,W share (a,b,c): they depend only on the advecting component of temp and on v_vis
    // (AdiSolver3D.cpp:739-762); T has its own because of t_vis and its own BC kind.
    R a_v, b_v, c_v;
    R a_t, b_t, c_t;
    R d[4];
};

template <int DIR>
__device__ __forceinline__ long long sweep_stride(long long plane, int dimz)
{
    return DIR == 0 ? plane : (DIR == 1 ? (long long)dimz : 1LL);
}

// Interior row (kind == ROW_INTERIOR): BuildMatrix, AdiSolver3D.cpp:755-801, with the
// stencils of TimeLayer3D.h:338-340 (d_x,d_y,d_z) and :554-588 (DissFuncX/Y/Z).
template <typename R, int DIR>
__device__ __forceinline__ void build_interior_row(const SweepParams<R> &p, long long idx, RowUVWT<R> &r)
{
    const long long sx = p.plane, sy = p.dimz, sz = 1;
    const long long ss = DIR == 0 ? sx : (DIR == 1 ? sy : sz);
    constexpr int M1 = DIR == 0 ? 1 : 0;
    constexpr int M2 = DIR == 2 ? 1 : 2;
    const long long sm1 = M1 == 0 ? sx : sy;
    const long long sm2 = M2 == 1 ? sy : sz;
    const R *tS = p.temp(DIR);
    const R two_ds = p.two_ds[DIR];

    const R q = tS[idx] / two_ds;               // temp->Vs->elem / (2*ds)
    r.a_v = -q - p.vis_v;  r.b_v = p.b_v;  r.c_v = q - p.vis_v;
    r.a_t = -q - p.vis_t;  r.b_t = p.b_t;  r.c_t = q - p.vis_t;

    // derivatives along the sweep axis of U,V,W (for DissFunc) and of T (momentum RHS)
    const R g0 = (p.temp(0)[idx + ss] - p.temp(0)[idx - ss]) / two_ds;
    const R g1 = (p.temp(1)[idx + ss] - p.temp(1)[idx - ss]) / two_ds;
    const R g2 = (p.temp(2)[idx + ss] - p.temp(2)[idx - ss]) / two_ds;
    const R gT = (p.temp(3)[idx + ss] - p.temp(3)[idx - ss]) / two_ds;
    // derivatives of the advecting component along the two other axes
    const R x1 = (tS[idx + sm1] - tS[idx - sm1]) / p.two_ds[M1];
    const R x2 = (tS[idx + sm2] - tS[idx - sm2]) / p.two_ds[M2];

    const R t0 = DIR == 0 ? (R(2) * g0) * g0 : g0 * g0;
    const R t1 = DIR == 1 ? (R(2) * g1) * g1 : g1 * g1;
    const R t2 = DIR == 2 ? (R(2) * g2) * g2 : g2 * g2;
    const R gm1 = M1 == 0 ? g0 : g1;
    const R gm2 = M2 == 1 ? g1 : g2;
    const R diss = (((t0 + t1) + t2) + gm1 * x1) + gm2 * x2;

    r.d[0] = p.cur(0)[idx] * R(3) / p.dt;
    r.d[1] = p.cur(1)[idx] * R(3) / p.dt;
    r.d[2] = p.cur(2)[idx] * R(3) / p.dt;
    r.d[DIR] = r.d[DIR] - p.v_T * gT;
    r.d[3] = p.cur(3)[idx] * R(3) / p.dt + p.t_phi * diss;
}

// START / END rows: ApplyBC0 / ApplyBC1, AdiSolver3D.cpp:804-852.
template <typename R>
__device__ __forceinline__ void build_bc_row(const SweepParams<R> &p, long long idx, int code, RowUVWT<R> &r)
{
    const bool is_start = (code & 3) == ROW_START;
    const bool vfree = code & ROW_VELFREE, tfree = code & ROW_TEMPFREE;
    if (vfree) {
        r.a_v = is_start ? R(0) : R(-1); r.b_v = R(2); r.c_v = is_start ? R(-1) : R(0);
        r.d[0] = r.d[1] = r.d[2] = R(0);
    } else {
        r.a_v = R(0); r.b_v = R(1); r.c_v = R(0);
        r.d[0] = p.node(0)[idx]; r.d[1] = p.node(1)[idx]; r.d[2] = p.node(2)[idx];
    }
    if (tfree) {
        r.a_t = is_start ? R(0) : R(-1); r.b_t = R(2); r.c_t = is_start ? R(-1) : R(0);
        r.d[3] = R(0);
    } else {
        r.a_t = R(0); r.b_t = R(1); r.c_t = R(0);
        r.d[3] = p.node(3)[idx];
    }
}

// One forward-elimination step of Common::SolveTridiagonal (Algorithms.h:23-32) for the
// two systems / four right-hand sides.  cp_* / dp_* are c'[i-1], d'[i-1] on entry and
// c'[i], d'[i] on return.  START rows take the i == 0 form (c0/b0, d0/b0); SKIP rows
// reset the state to zero.
template <typename R>
__device__ __forceinline__ void thomas_forward(int kind, const RowUVWT<R> &r, R &cp_v, R &cp_t, R dp[4])
{
    if (kind == ROW_SKIP) {
        cp_v = R(0); cp_t = R(0); dp[0] = dp[1] = dp[2] = dp[3] = R(0);
        return;
    }
    R den_v, den_t, n0, n1, n2, n3;
    if (kind == ROW_START) {
        den_v = r.b_v; den_t = r.b_t;
        n0 = r.d[0]; n1 = r.d[1]; n2 = r.d[2]; n3 = r.d[3];
    } else {
        den_v = r.b_v - r.a_v * cp_v;
        den_t = r.b_t - r.a_t * cp_t;
        n0 = r.d[0] - dp[0] * r.a_v;
        n1 = r.d[1] - dp[1] * r.a_v;
        n2 = r.d[2] - dp[2] * r.a_v;
        n3 = r.d[3] - dp[3] * r.a_t;
    }
    // END rows: the reference sets c[num-1] = 0 before dividing (Algorithms.h:23)
    cp_v = (kind == ROW_END ? R(0) : r.c_v) / den_v;
    cp_t = (kind == ROW_END ? R(0) : r.c_t) / den_t;
    dp[0] = n0 / den_v; dp[1] = n1 / den_v; dp[2] = n2 / den_v; dp[3] = n3 / den_t;
}
