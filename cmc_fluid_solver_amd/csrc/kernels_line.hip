// Thread-per-line Thomas sweep (FS3D_SWEEP_LINE): the fully general kernel.
//
// One thread owns one grid line of the sweep direction and runs the reference's
// sequential recurrence (Algorithms.h:21-38) over it: forward elimination writes
// c'(2) and d'(4) per cell to an HBM scratch, back-substitution reads them in reverse
// and scatters x into `next` (UpdateSegment, AdiSolver3D.cpp:707-730), optionally
// followed by the merge into temp (MergeLayerTo, TimeLayer3D.h:664-683).
// X and Y sweeps are coalesced (lanes along k); the Z sweep is not (lanes along j) --
// this kernel is the correctness baseline and the fallback for dims the pipelined
// kernel does not cover.  HBM traffic: 8 (in) + 6 (scratch out) + 6 (scratch in)
// + 4 (temp again) + 8 (out) = 32 words/cell against the 16-word algorithmic figure.
#include "fs3d_rows.h"

template <typename R, int DIR>
__global__ void __launch_bounds__(256) k_sweep_line(SweepParams<R> p)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long nlines, base, stride;
    int n;
    if (DIR == 0) { nlines = p.plane; base = tid; stride = p.plane; n = p.dimx; }
    else if (DIR == 1) {
        nlines = (long long)p.dimx * p.dimz;
        base = (tid / p.dimz) * p.plane + (tid % p.dimz); stride = p.dimz; n = p.dimy;
    } else { nlines = (long long)p.dimx * p.dimy; base = tid * p.dimz; stride = 1; n = p.dimz; }
    if (tid >= nlines) return;

    R cp_v = R(0), cp_t = R(0), dp[4] = {R(0), R(0), R(0), R(0)};
    for (int s = 0; s < n; s++) {
        const long long idx = base + s * stride;
        const int code = (p.code[idx] >> (4 * DIR)) & 0xF;
        const int kind = code & 3;
        RowUVWT<R> r;
        if (kind == ROW_INTERIOR) build_interior_row<R, DIR>(p, idx, r);
        else if (kind != ROW_SKIP) build_bc_row<R>(p, idx, code, r);
        thomas_forward<R>(kind, r, cp_v, cp_t, dp);
        p.scr(0)[idx] = cp_v; p.scr(1)[idx] = cp_t;
        p.scr(2)[idx] = dp[0]; p.scr(3)[idx] = dp[1]; p.scr(4)[idx] = dp[2]; p.scr(5)[idx] = dp[3];
    }
    // back-substitution: x[n-1] = d'[n-1]; x[i] = d'[i] - c'[i]*x[i+1]
    R x[4] = {R(0), R(0), R(0), R(0)};
    for (int s = n - 1; s >= 0; s--) {
        const long long idx = base + s * stride;
        const int cw = p.code[idx];
        const int kind = (cw >> (4 * DIR)) & 3;
        const R c_v = p.scr(0)[idx], c_t = p.scr(1)[idx];
        const R d0 = p.scr(2)[idx], d1 = p.scr(3)[idx], d2 = p.scr(4)[idx], d3 = p.scr(5)[idx];
        if (kind == ROW_END || kind == ROW_SKIP) { x[0] = d0; x[1] = d1; x[2] = d2; x[3] = d3; }
        else {
            x[0] = d0 - c_v * x[0]; x[1] = d1 - c_v * x[1];
            x[2] = d2 - c_v * x[2]; x[3] = d3 - c_t * x[3];
        }
        if (kind != ROW_SKIP) {
            p.next(0)[idx] = x[0]; p.next(1)[idx] = x[1]; p.next(2)[idx] = x[2]; p.next(3)[idx] = x[3];
        }
        if (p.merge) {
            const bool is_in = ((cw >> CODE_TYPE_SHIFT) & 3) == FS3D_NODE_IN;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                R t = p.temp(v)[idx];
                if (is_in) {
                    // NODE_IN cell outside every segment (run without a closing cell,
                    // Grid3D.cpp:87-117): the reference merges the stale `next` value
                    const R xv = kind != ROW_SKIP ? x[v] : p.next(v)[idx];
                    t = (t + xv) / R(2);
                    if (p.merge == 2) t = (t + xv) / R(2);
                }
                p.temp_out(v)[idx] = t;
            }
        }
    }
}

template <typename R>
void launch_sweep_line(fs3d_ctx *c, int dir, const SweepParams<R> &p)
{
    long long nlines = dir == 0 ? p.plane : (dir == 1 ? (long long)p.dimx * p.dimz : (long long)p.dimx * p.dimy);
    const int bs = 64;   // few lines exist (N^2): small blocks spread them over all CUs
    const unsigned grid = (unsigned)((nlines + bs - 1) / bs);
    switch (dir) {
    case 0: hipLaunchKernelGGL((k_sweep_line<R, 0>), dim3(grid), dim3(bs), 0, c->stream, p); break;
    case 1: hipLaunchKernelGGL((k_sweep_line<R, 1>), dim3(grid), dim3(bs), 0, c->stream, p); break;
    default: hipLaunchKernelGGL((k_sweep_line<R, 2>), dim3(grid), dim3(bs), 0, c->stream, p); break;
    }
}

template void launch_sweep_line<float>(fs3d_ctx *, int, const SweepParams<float> &);
template void launch_sweep_line<double>(fs3d_ctx *, int, const SweepParams<double> &);

// ---------------------------------------------------------------------------------------------
// Cross-slab X sweep (multi-GPU): the same recurrence cut at slab boundaries.
// Rank r eliminates planes [0, dimx) of its slab starting from the (c', d') that rank r-1 reached
// at its last plane (carry_in, 6 values per line; none on rank 0), leaves c', d' of its cells in the
// HBM scratch and its own last (c', d') in carry_out.  Back-substitution runs the other way with
// the x of the neighbour's first plane (xcarry_in, 4 values per line; none on the last rank).
// This is the reference's pipelined Thomas (AdiSolver3D.cu:524-640) with CPU-ordering semantics;
// cell for cell it performs the single-GPU arithmetic, so slabbed results are bit-identical.
// Carry layout: [value][line], line = j*dimz + k (the caller packs/sends the [l0,l1) part of each value row).
template <typename R>
__global__ void __launch_bounds__(256) k_xsweep_fwd(SweepParams<R> p, const R *carry_in, R *carry_out, long long l0, long long l1)
{
    const long long tid = l0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;   // lines [l0, l1) of the plane
    const long long nlines = p.plane;
    if (tid >= l1) return;
    R cp_v = R(0), cp_t = R(0), dp[4] = {R(0), R(0), R(0), R(0)};
    if (carry_in) {
        cp_v = carry_in[0 * nlines + tid]; cp_t = carry_in[1 * nlines + tid];
        dp[0] = carry_in[2 * nlines + tid]; dp[1] = carry_in[3 * nlines + tid];
        dp[2] = carry_in[4 * nlines + tid]; dp[3] = carry_in[5 * nlines + tid];
    }
    for (int s = 0; s < p.dimx; s++) {
        const long long idx = tid + s * p.plane;
        const int code = p.code[idx] & 0xF;
        const int kind = code & 3;
        RowUVWT<R> r;
        if (kind == ROW_INTERIOR) build_interior_row<R, 0>(p, idx, r);
        else if (kind != ROW_SKIP) build_bc_row<R>(p, idx, code, r);
        thomas_forward<R>(kind, r, cp_v, cp_t, dp);
        p.scr(0)[idx] = cp_v; p.scr(1)[idx] = cp_t;
        p.scr(2)[idx] = dp[0]; p.scr(3)[idx] = dp[1]; p.scr(4)[idx] = dp[2]; p.scr(5)[idx] = dp[3];
    }
    carry_out[0 * nlines + tid] = cp_v; carry_out[1 * nlines + tid] = cp_t;
    carry_out[2 * nlines + tid] = dp[0]; carry_out[3 * nlines + tid] = dp[1];
    carry_out[4 * nlines + tid] = dp[2]; carry_out[5 * nlines + tid] = dp[3];
}

template <typename R>
__global__ void __launch_bounds__(256) k_xsweep_bwd(SweepParams<R> p, const R *xcarry_in, R *xcarry_out, long long l0, long long l1)
{
    const long long tid = l0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nlines = p.plane;
    if (tid >= l1) return;
    R x[4] = {R(0), R(0), R(0), R(0)};
    if (xcarry_in) { x[0] = xcarry_in[tid]; x[1] = xcarry_in[nlines + tid]; x[2] = xcarry_in[2 * nlines + tid]; x[3] = xcarry_in[3 * nlines + tid]; }
    for (int s = p.dimx - 1; s >= 0; s--) {
        const long long idx = tid + s * p.plane;
        const int cw = p.code[idx];
        const int kind = cw & 3;
        const R c_v = p.scr(0)[idx], c_t = p.scr(1)[idx];
        const R d0 = p.scr(2)[idx], d1 = p.scr(3)[idx], d2 = p.scr(4)[idx], d3 = p.scr(5)[idx];
        if (kind == ROW_END || kind == ROW_SKIP) { x[0] = d0; x[1] = d1; x[2] = d2; x[3] = d3; }
        else {
            x[0] = d0 - c_v * x[0]; x[1] = d1 - c_v * x[1];
            x[2] = d2 - c_v * x[2]; x[3] = d3 - c_t * x[3];
        }
        if (kind != ROW_SKIP) {
            p.next(0)[idx] = x[0]; p.next(1)[idx] = x[1]; p.next(2)[idx] = x[2]; p.next(3)[idx] = x[3];
        }
        if (p.merge) {
            const bool is_in = ((cw >> CODE_TYPE_SHIFT) & 3) == FS3D_NODE_IN;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                R t = p.temp(v)[idx];
                if (is_in) {
                    const R xv = kind != ROW_SKIP ? x[v] : p.next(v)[idx];
                    t = (t + xv) / R(2);
                    if (p.merge == 2) t = (t + xv) / R(2);
                }
                p.temp_out(v)[idx] = t;
            }
        }
    }
    xcarry_out[tid] = x[0]; xcarry_out[nlines + tid] = x[1]; xcarry_out[2 * nlines + tid] = x[2]; xcarry_out[3 * nlines + tid] = x[3];
}

template <typename R>
void launch_xsweep_fwd(fs3d_ctx *c, const SweepParams<R> &p, const void *carry_in, void *carry_out, long long l0, long long l1)
{
    const unsigned grid = (unsigned)((l1 - l0 + 63) / 64);
    hipLaunchKernelGGL((k_xsweep_fwd<R>), dim3(grid), dim3(64), 0, c->stream, p, (const R *)carry_in, (R *)carry_out, l0, l1);
}
template <typename R>
void launch_xsweep_bwd(fs3d_ctx *c, const SweepParams<R> &p, const void *xcarry_in, void *xcarry_out, long long l0, long long l1)
{
    const unsigned grid = (unsigned)((l1 - l0 + 63) / 64);
    hipLaunchKernelGGL((k_xsweep_bwd<R>), dim3(grid), dim3(64), 0, c->stream, p, (const R *)xcarry_in, (R *)xcarry_out, l0, l1);
}
template void launch_xsweep_fwd<float>(fs3d_ctx *, const SweepParams<float> &, const void *, void *, long long, long long);
template void launch_xsweep_fwd<double>(fs3d_ctx *, const SweepParams<double> &, const void *, void *, long long, long long);
template void launch_xsweep_bwd<float>(fs3d_ctx *, const SweepParams<float> &, const void *, void *, long long, long long);
template void launch_xsweep_bwd<double>(fs3d_ctx *, const SweepParams<double> &, const void *, void *, long long, long long);

// ---------------------------------------------------------------------------------------------
// Cross-slab X sweep, reduced-interface form (all ranks at once; cmc_fluid_solver_amd/partition.py with one chunk
// per x-slab).  The reference pipelines the recurrence rank by rank (AdiSolver3D.cu:524-640): rank r cannot start
// before rank r-1 has finished.  Here every rank eliminates the planes of its slab at the same time:
//   k_xiface    per line: down- and up-sweep over the slab's cells before its last plane -> the interface row of
//               the last plane (A, Bp, cl, Dp) and x_first = Gf - Vf X_{r-1} - Wf X_r, 18 words per line;
//   all-gather  of those 18 words per line (the only exchange of the sweep);
//   k_xreduce   per line: the R x R interface system (R = ranks), then for THIS rank the value just below its slab
//               (X_{r-1}) and just above it (x_first of rank r+1) -- written in the carry layout of the halves;
//   the existing halves (k_xsweep_fwd / k_xsweep_bwd, or the pipe kernel's MODE 1 / 2) solve the slab with those
//   two values given, all lines at once, no further exchange.
// Same equations as the pipelined form, algebraically exact; the rounding of the interface solve differs (results
// equal the single-GPU fields to rounding, not bit for bit; the pipelined form stays available, FS3D_OPT_XSOLVE).
template <typename R>
__device__ __forceinline__ void xrow(const SweepParams<R> &p, long long idx, RowUVWT<R> &r)
{
    const int code = p.code[idx] & 0xF, kind = code & 3;
    if (kind == ROW_INTERIOR) build_interior_row<R, 0>(p, idx, r);
    else if (kind != ROW_SKIP) build_bc_row<R>(p, idx, code, r);
    else { r.a_v = r.c_v = r.a_t = r.c_t = R(0); r.b_v = r.b_t = R(1); r.d[0] = r.d[1] = r.d[2] = r.d[3] = R(0); }
    if (kind == ROW_END) { r.c_v = R(0); r.c_t = R(0); }       // Algorithms.h:23
    if (kind == ROW_START) { r.a_v = R(0); r.a_t = R(0); }
}

#define XIFACE_WORDS 18
template <typename R>
__global__ void __launch_bounds__(256) k_xiface(SweepParams<R> p, R *out)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nl = p.plane;
    if (tid >= nl) return;
    const int n = p.dimx;
    RowUVWT<R> r;
    // Rows are built four at a time: their loads do not depend on the recurrence, so four cells' worth are in flight
    // together (a thin slab gives every thread a short, latency-bound walk: 31 cells of a 32-plane slab).
    constexpr int B = 4;
    RowUVWT<R> rb[B];
    // down: x[n-2] = dp - lp X_{r-1} - cp X_r
    R cpv = R(0), lpv = R(-1), cpt = R(0), lpt = R(-1), dp[4] = {R(0), R(0), R(0), R(0)};
    for (int s0 = 0; s0 < n - 1; s0 += B) {
#pragma unroll
        for (int b = 0; b < B; b++) xrow<R>(p, tid + (long long)(s0 + b < n - 1 ? s0 + b : n - 2) * p.plane, rb[b]);
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (s0 + b >= n - 1) break;
            const RowUVWT<R> &r = rb[b];
            const R dv = r.b_v - r.a_v * cpv, dt = r.b_t - r.a_t * cpt;
            dp[0] = (r.d[0] - r.a_v * dp[0]) / dv; dp[1] = (r.d[1] - r.a_v * dp[1]) / dv; dp[2] = (r.d[2] - r.a_v * dp[2]) / dv;
            dp[3] = (r.d[3] - r.a_t * dp[3]) / dt;
            lpv = -r.a_v * lpv / dv; lpt = -r.a_t * lpt / dt;
            cpv = r.c_v / dv; cpt = r.c_t / dt;
        }
    }
    // up: x[0] = ep - ap X_{r-1} - up X_r
    R apv = R(0), upv = R(-1), apt = R(0), upt = R(-1), ep[4] = {R(0), R(0), R(0), R(0)};
    for (int s0 = n - 2; s0 >= 0; s0 -= B) {
#pragma unroll
        for (int b = 0; b < B; b++) xrow<R>(p, tid + (long long)(s0 - b >= 0 ? s0 - b : 0) * p.plane, rb[b]);
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (s0 - b < 0) break;
            const RowUVWT<R> &r = rb[b];
            const R dv = r.b_v - r.c_v * apv, dt = r.b_t - r.c_t * apt;
            ep[0] = (r.d[0] - r.c_v * ep[0]) / dv; ep[1] = (r.d[1] - r.c_v * ep[1]) / dv; ep[2] = (r.d[2] - r.c_v * ep[2]) / dv;
            ep[3] = (r.d[3] - r.c_t * ep[3]) / dt;
            upv = -r.c_v * upv / dv; upt = -r.c_t * upt / dt;
            apv = r.a_v / dv; apt = r.a_t / dt;
        }
    }
    xrow<R>(p, tid + (long long)(n - 1) * p.plane, r);
    R *o = out + tid;
    o[0 * nl] = -r.a_v * lpv; o[1 * nl] = r.b_v - r.a_v * cpv; o[2 * nl] = r.c_v; o[3 * nl] = apv; o[4 * nl] = upv;
    o[5 * nl] = -r.a_t * lpt; o[6 * nl] = r.b_t - r.a_t * cpt; o[7 * nl] = r.c_t; o[8 * nl] = apt; o[9 * nl] = upt;
    o[10 * nl] = r.d[0] - r.a_v * dp[0]; o[11 * nl] = r.d[1] - r.a_v * dp[1]; o[12 * nl] = r.d[2] - r.a_v * dp[2]; o[13 * nl] = r.d[3] - r.a_t * dp[3];
    o[14 * nl] = ep[0]; o[15 * nl] = ep[1]; o[16 * nl] = ep[2]; o[17 * nl] = ep[3];
}

#define XREDUCE_MAXR FS3D_XREDUCE_MAX_RANKS      // larger groups: refused (explicit) / pipelined form (auto), fs3d_hip.hip
template <typename R>
__global__ void __launch_bounds__(256) k_xreduce(const R *all, long long nl, int nranks, int me, R *carry_in, R *xcarry_in)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= nl) return;
    const long long rs = (long long)XIFACE_WORDS * nl;          // one rank's block
    auto W = [&](int r, int w) { return all[r * rs + w * nl + tid]; };
    R xl[4], xr[4];
    for (int sys = 0; sys < 4; sys++) {
        const int m = sys == 3 ? 5 : 0;
        R cp[XREDUCE_MAXR], dq[XREDUCE_MAXR];
        R c_ = R(0), d_ = R(0);
        for (int r = 0; r < nranks; r++) {
            const R lo = W(r, m + 0), cl = W(r, m + 2);
            R di = W(r, m + 1), up = R(0), rhs = W(r, 10 + sys);
            if (r + 1 < nranks) { di = di - cl * W(r + 1, m + 3); up = -cl * W(r + 1, m + 4); rhs = rhs - cl * W(r + 1, 14 + sys); }
            const R den = di - lo * c_;
            c_ = up / den; d_ = (rhs - lo * d_) / den;
            cp[r] = c_; dq[r] = d_;
        }
        R x = dq[nranks - 1], xme = R(0), xabove = R(0), xbelow = R(0);
        for (int r = nranks - 1; r >= 0; r--) {
            if (r < nranks - 1) x = dq[r] - cp[r] * x;
            if (r == me + 1) xabove = x;
            if (r == me) xme = x;
            if (r == me - 1) xbelow = x;
        }
        xl[sys] = me > 0 ? xbelow : R(0);
        // the first cell of the slab above: Gf - Vf X_me - Wf X_{me+1}
        xr[sys] = me + 1 < nranks ? W(me + 1, 14 + sys) - W(me + 1, m + 3) * xme - W(me + 1, m + 4) * xabove : R(0);
    }
    // carry layouts of the halves: forward [c'_uvw, c'_T, d'_U, d'_V, d'_W, d'_T] = [0, 0, X_{r-1}]; backward [x_U .. x_T]
    carry_in[0 * nl + tid] = R(0); carry_in[1 * nl + tid] = R(0);
    for (int k = 0; k < 4; k++) { carry_in[(2 + k) * nl + tid] = xl[k]; xcarry_in[k * nl + tid] = xr[k]; }
}

// ---- (r3) the interface solve DISTRIBUTED over the ranks (FS3D_XSOLVE_REDUCED_A2A; xGMI is point-to-point: an all-gather delivers
// every rank's 18 words of every line to every rank, (R-1) x 18 words per line received, and every rank then solves every line's
// R x R system redundantly).  Here rank r owns the lines [r Lp, (r+1) Lp) of the plane:
//   k_xpack        this slab's words [word][line] -> blocks [owner][word][line - owner Lp]      (one contiguous send per peer)
//   all-to-all #1  rank r receives the words of ITS lines from every rank                       ((R-1)/R x 18 words per line)
//   k_xreduce_a2a  per owned line: the R x R system once, then for EVERY rank the value below / above its slab -> [rank][8][line]
//   all-to-all #2  the 8 boundary words per line back to the rank they belong to                 ((R-1)/R x 8 words per line)
//   k_xunpack      blocks [owner][8][line] -> the carry layout of the slab solve
// Same operations on the same values as k_xreduce: bit-identical to the all-gather form (tests/test_gpu_slabs.py).
template <typename R>
__global__ void __launch_bounds__(256) k_xpack(const R *in, long long nl, long long lp, R *out)
{
    const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nl) return;
    const long long r = l / lp, t = l - r * lp;
    R *o = out + r * XIFACE_WORDS * lp + t;
#pragma unroll
    for (int w = 0; w < XIFACE_WORDS; w++) o[w * lp] = in[w * nl + l];
}

template <typename R>
__global__ void __launch_bounds__(256) k_xreduce_a2a(const R *all, long long nl, long long lp, int nranks, int me, R *out)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lp || (long long)me * lp + t >= nl) return;
    const long long rs = (long long)XIFACE_WORDS * lp;          // one rank's block
    auto W = [&](int r, int w) { return all[r * rs + w * lp + t]; };
    for (int sys = 0; sys < 4; sys++) {
        const int m = sys == 3 ? 5 : 0;
        R cp[XREDUCE_MAXR], dq[XREDUCE_MAXR], X[XREDUCE_MAXR];
        R c_ = R(0), d_ = R(0);
        for (int r = 0; r < nranks; r++) {
            const R lo = W(r, m + 0), cl = W(r, m + 2);
            R di = W(r, m + 1), up = R(0), rhs = W(r, 10 + sys);
            if (r + 1 < nranks) { di = di - cl * W(r + 1, m + 3); up = -cl * W(r + 1, m + 4); rhs = rhs - cl * W(r + 1, 14 + sys); }
            const R den = di - lo * c_;
            c_ = up / den; d_ = (rhs - lo * d_) / den;
            cp[r] = c_; dq[r] = d_;
        }
        R x = dq[nranks - 1];
        for (int r = nranks - 1; r >= 0; r--) {
            if (r < nranks - 1) x = dq[r] - cp[r] * x;
            X[r] = x;
        }
        for (int r = 0; r < nranks; r++) {
            const R xl = r > 0 ? X[r - 1] : R(0);
            const R xr = r + 1 < nranks ? W(r + 1, 14 + sys) - W(r + 1, m + 3) * X[r] - W(r + 1, m + 4) * X[r + 1] : R(0);
            out[(long long)r * 8 * lp + sys * lp + t] = xl;
            out[(long long)r * 8 * lp + (4 + sys) * lp + t] = xr;
        }
    }
}

template <typename R>
__global__ void __launch_bounds__(256) k_xunpack(const R *in, long long nl, long long lp, R *carry_in, R *xcarry_in)
{
    const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nl) return;
    const long long s = l / lp, t = l - s * lp;
    const R *b = in + s * 8 * lp + t;
    carry_in[0 * nl + l] = R(0); carry_in[1 * nl + l] = R(0);
#pragma unroll
    for (int k = 0; k < 4; k++) { carry_in[(2 + k) * nl + l] = b[k * lp]; xcarry_in[k * nl + l] = b[(4 + k) * lp]; }
}

template <typename R>
void launch_xpack(fs3d_ctx *c, const void *in, long long nl, long long lp, void *out)
{
    hipLaunchKernelGGL((k_xpack<R>), dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, c->stream, (const R *)in, nl, lp, (R *)out);
}
template <typename R>
void launch_xreduce_a2a(fs3d_ctx *c, const void *all, long long nl, long long lp, int nranks, int me, void *out)
{
    hipLaunchKernelGGL((k_xreduce_a2a<R>), dim3((unsigned)((lp + 255) / 256)), dim3(256), 0, c->stream, (const R *)all, nl, lp, nranks, me, (R *)out);
}
template <typename R>
void launch_xunpack(fs3d_ctx *c, const void *in, long long nl, long long lp, void *carry_in, void *xcarry_in)
{
    hipLaunchKernelGGL((k_xunpack<R>), dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, c->stream, (const R *)in, nl, lp, (R *)carry_in, (R *)xcarry_in);
}
template void launch_xpack<float>(fs3d_ctx *, const void *, long long, long long, void *);
template void launch_xpack<double>(fs3d_ctx *, const void *, long long, long long, void *);
template void launch_xreduce_a2a<float>(fs3d_ctx *, const void *, long long, long long, int, int, void *);
template void launch_xreduce_a2a<double>(fs3d_ctx *, const void *, long long, long long, int, int, void *);
template void launch_xunpack<float>(fs3d_ctx *, const void *, long long, long long, void *, void *);
template void launch_xunpack<double>(fs3d_ctx *, const void *, long long, long long, void *, void *);

template <typename R>
void launch_xiface(fs3d_ctx *c, const SweepParams<R> &p, void *out)
{
    hipLaunchKernelGGL((k_xiface<R>), dim3((unsigned)((p.plane + 63) / 64)), dim3(64), 0, c->stream, p, (R *)out);
}
template <typename R>
void launch_xreduce(fs3d_ctx *c, const void *all, long long nl, int nranks, int me, void *carry_in, void *xcarry_in)
{
    hipLaunchKernelGGL((k_xreduce<R>), dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, c->stream, (const R *)all, nl, nranks, me, (R *)carry_in, (R *)xcarry_in);
}
template void launch_xiface<float>(fs3d_ctx *, const SweepParams<float> &, void *);
template void launch_xiface<double>(fs3d_ctx *, const SweepParams<double> &, void *);
template void launch_xreduce<float>(fs3d_ctx *, const void *, long long, int, int, void *, void *);
template void launch_xreduce<double>(fs3d_ctx *, const void *, long long, int, int, void *, void *);
