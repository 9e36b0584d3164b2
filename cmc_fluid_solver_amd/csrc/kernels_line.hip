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
