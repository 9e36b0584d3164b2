// Wave-pipelined exact Thomas sweep (FS3D_SWEEP_PIPE) for CDNA4.
//
// Work decomposition
//   bundle    = 64 neighbouring grid lines of the sweep direction (lanes of a wave),
//               whole length n = dim along the sweep axis.  One workgroup per bundle.
//   workgroup = NW = 8 waves; wave w owns cells [w*CH, (w+1)*CH) of every line of the bundle.
//   X, Y sweeps: lanes run along k (unit stride)  -> every global access is a coalesced row.
//   Z sweep    : lanes run along j, a thread's CH cells are contiguous in memory.
//
// Phases (bit-exact w.r.t. the sequential reference, Algorithms.h:21-38)
//   P  all waves, in parallel: load cur/temp (+ neighbours), build the rows (fs3d_rows.h),
//      keep per cell q,dU,dV,dW in registers and dT in LDS.
//   F  forward elimination as a relay: wave 0 eliminates its chunk, hands (c',d') of its
//      last cell to wave 1 through LDS, ... The recurrence is the reference's, cell by cell;
//      c'_uvw,d'_U,d'_V,d'_W overwrite the row data in registers, c'_T,d'_T live in LDS.
//   B  back-substitution as the reverse relay, registers/LDS only: x overwrites c',d'.
//   O  all waves, in parallel and off the relay's critical path: scatter x to `next`
//      (UpdateSegment, AdiSolver3D.cpp:707-730) and apply the merge into temp
//      (TimeLayer3D.h:415-436) in the same pass.
// Nothing but the 8 input and 8 output words per cell (+2-byte cell code) moves to/from HBM:
// the 6 words/cell of c',d' that a thread-per-line kernel spills stay on chip (128 VGPRs
// per lane + 128 KiB LDS per workgroup for a 256-cell fp32 line).
#include "fs3d_rows.h"

#define PIPE_NW 8

template <typename R> struct PipeLds { };

// one workgroup = one bundle
template <typename R, int DIR, int CH>
__global__ void __launch_bounds__(PIPE_NW * 64, 2) k_sweep_pipe(SweepParams<R> p, int n_o, int n_tiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave id, kept scalar

    // ---- XCD-aware bundle id: blocks b, b+8, b+16.. share an XCD (round-robin dispatch);
    // give each XCD a contiguous range of logical ids so that the +-1 planes a bundle reads
    // are being streamed by sibling CUs of the same L2 (speed only, never correctness).
    const int nb = gridDim.x;
    int lb = blockIdx.x;
    {
        const int q = nb >> 3, r = nb & 7, x = lb & 7, slot = lb >> 3;
        lb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + slot;
    }
    const int tile = lb / n_o, o = lb - tile * n_o;      // tile-major: consecutive ids = consecutive planes

    const int n = DIR == 0 ? p.dimx : (DIR == 1 ? p.dimy : p.dimz);
    const int la_len = DIR == 2 ? p.dimy : p.dimz;       // length of the lane axis
    const int l = tile * 64 + lane;
    const bool lane_valid = l < la_len;
    const long long ss = DIR == 0 ? p.plane : (DIR == 1 ? (long long)p.dimz : 1LL);
    const long long so = DIR == 0 ? (long long)p.dimz : p.plane;
    // every address = (field + wave-uniform element offset)[32-bit per-lane offset]: the uniform part
    // lives in SGPRs, one VGPR serves all accesses (global_load v, v_off, s[base:base+1])
    const long long ub = DIR == 0 ? (long long)o * p.dimz + tile * 64
                       : (DIR == 1 ? (long long)o * p.plane + tile * 64 : (long long)o * p.plane + (long long)tile * 64 * p.dimz);
    const int vo = DIR == 2 ? lane * p.dimz : lane;
    const int vsl = DIR == 2 ? p.dimz : 1;               // per-lane offset step of a lane-axis neighbour
    const bool hi_edge = lane == 63 || l + 1 >= la_len;  // right lane neighbour not in this wave
    const bool lo_edge = lane == 0;

    // LDS: [n][64] d_T / d'_T, [n][64] c'_T, relay slots
    R *ldsD = (R *)smem_raw;
    R *ldsC = ldsD + (size_t)PIPE_NW * CH * 64;
    R *relay = ldsC + (size_t)PIPE_NW * CH * 64;       // 6 x 64 forward, reused 4 x 64 backward

    const int s0 = w * CH;
    const R *tS = p.temp(DIR);

    // per-cell register storage: q -> c'_uvw ; dU,dV,dW -> d'_U,d'_V,d'_W
    R st0[CH], st1[CH], st2[CH], st3[CH];
    unsigned cpack[(CH + 7) / 8];
    unsigned inmask = 0;
#pragma unroll
    for (int i = 0; i < (CH + 7) / 8; i++) cpack[i] = 0;

    // ------------------------------------------------------------------ P: rows
    {
        R wU[3], wV[3], wW[3], wT[3];
        auto ld = [&](const R *f, int s) -> R {
            return (lane_valid && s >= 0 && s < n) ? (f + (ub + (long long)s * ss))[vo] : R(0);
        };
        wU[0] = ld(p.temp(0), s0 - 1); wV[0] = ld(p.temp(1), s0 - 1); wW[0] = ld(p.temp(2), s0 - 1); wT[0] = ld(p.temp(3), s0 - 1);
        wU[1] = ld(p.temp(0), s0);     wV[1] = ld(p.temp(1), s0);     wW[1] = ld(p.temp(2), s0);     wT[1] = ld(p.temp(3), s0);
#pragma unroll
        for (int t = 0; t < CH; t++) {
            const int s = s0 + t;
            const long long us = ub + (long long)s * ss;      // wave-uniform offset of cell s
            wU[2] = ld(p.temp(0), s + 1); wV[2] = ld(p.temp(1), s + 1); wW[2] = ld(p.temp(2), s + 1); wT[2] = ld(p.temp(3), s + 1);
            int cw = 0;
            if (lane_valid && s < n) cw = (p.code + us)[vo];
            const int code = (cw >> (4 * DIR)) & 0xF;
            const int kind = code & 3;
            cpack[t >> 3] |= (unsigned)code << (4 * (t & 7));
            if (lane_valid && s < n && ((cw >> CODE_TYPE_SHIFT) & 3) == FS3D_NODE_IN) inmask |= 1u << t;
            const R tc = DIR == 0 ? wU[1] : (DIR == 1 ? wV[1] : wW[1]);     // advecting component at the cell
            // lane-axis neighbours of tc: neighbouring lanes, real loads at the wave's edges
            R l_lo = __shfl_up(tc, 1, 64), l_hi = __shfl_down(tc, 1, 64);
            R q = R(0), d0 = R(0), d1 = R(0), d2 = R(0), d3 = R(0);
            if (kind == ROW_INTERIOR) {
                if (lo_edge) l_lo = (tS + us)[vo - vsl];
                if (hi_edge) l_hi = (tS + us)[vo + vsl];
                const R o_lo = (tS + (us - so))[vo], o_hi = (tS + (us + so))[vo];
                const R two_ds = p.two_ds[DIR];
                constexpr int M1 = DIR == 0 ? 1 : 0;          // axis of the `o` neighbours
                constexpr int M2 = DIR == 2 ? 1 : 2;          // axis of the lane neighbours
                q = tc / two_ds;
                const R g0 = (wU[2] - wU[0]) / two_ds;
                const R g1 = (wV[2] - wV[0]) / two_ds;
                const R g2 = (wW[2] - wW[0]) / two_ds;
                const R gT = (wT[2] - wT[0]) / two_ds;
                const R x1 = (o_hi - o_lo) / p.two_ds[M1];
                const R x2 = (l_hi - l_lo) / p.two_ds[M2];
                const R t0 = DIR == 0 ? (R(2) * g0) * g0 : g0 * g0;
                const R t1 = DIR == 1 ? (R(2) * g1) * g1 : g1 * g1;
                const R t2 = DIR == 2 ? (R(2) * g2) * g2 : g2 * g2;
                const R gm1 = M1 == 0 ? g0 : g1;
                const R gm2 = M2 == 1 ? g1 : g2;
                const R diss = (((t0 + t1) + t2) + gm1 * x1) + gm2 * x2;
                d0 = (p.cur(0) + us)[vo] * R(3) / p.dt;
                d1 = (p.cur(1) + us)[vo] * R(3) / p.dt;
                d2 = (p.cur(2) + us)[vo] * R(3) / p.dt;
                if (DIR == 0) d0 = d0 - p.v_T * gT;
                if (DIR == 1) d1 = d1 - p.v_T * gT;
                if (DIR == 2) d2 = d2 - p.v_T * gT;
                d3 = (p.cur(3) + us)[vo] * R(3) / p.dt + p.t_phi * diss;
            } else if (kind != ROW_SKIP) {
                // ApplyBC0/ApplyBC1 right-hand sides (AdiSolver3D.cpp:804-852): node value or 0
                if (!(code & ROW_VELFREE)) { d0 = (p.node(0) + us)[vo]; d1 = (p.node(1) + us)[vo]; d2 = (p.node(2) + us)[vo]; }
                if (!(code & ROW_TEMPFREE)) d3 = (p.node(3) + us)[vo];
            }
            st0[t] = q; st1[t] = d0; st2[t] = d1; st3[t] = d2;
            ldsD[(size_t)s * 64 + lane] = d3;
            wU[0] = wU[1]; wU[1] = wU[2]; wV[0] = wV[1]; wV[1] = wV[2];
            wW[0] = wW[1]; wW[1] = wW[2]; wT[0] = wT[1]; wT[1] = wT[2];
            // keep the scheduler from hoisting every cell's loads to the top (it would spill):
            // one cell's ~13 loads x 8 waves x 64 lanes is already ample memory-level parallelism
            if ((t & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ------------------------------------------------------------------ F: forward relay
    // Every wave executes exactly PIPE_NW barriers: w of them waiting for its turn, the rest
    // after its own chunk (straight-line code: the chunk body exists once, outside any
    // data-dependent control flow, so the register arrays are never copied).
    for (int i = 0; i < w; i++) __syncthreads();
    {
        R cp_v = R(0), cp_t = R(0), dp[4] = {R(0), R(0), R(0), R(0)};
        if (w > 0) {
            cp_v = relay[0 * 64 + lane]; cp_t = relay[1 * 64 + lane];
            dp[0] = relay[2 * 64 + lane]; dp[1] = relay[3 * 64 + lane];
            dp[2] = relay[4 * 64 + lane]; dp[3] = relay[5 * 64 + lane];
        }
        // Branch-free chain body: the row kinds only steer selects, so the serial critical
        // path is the reference's arithmetic (2 mul, 2 sub, 6 div per cell) and nothing else.
        //   INTERIOR a = -q - vis, b = 3/dt + 2 vis, c = q - vis      (AdiSolver3D.cpp:760-762)
        //   START    a = 0,  FREE: b = 2, c = -1 ; NOSLIP: b = 1, c = 0 (ApplyBC0, :804-827)
        //   END      c = 0,  FREE: a = -1, b = 2 ; NOSLIP: a = 0, b = 1 (ApplyBC1, :829-852)
        //   SKIP     identity row.  START and SKIP rows restart the recurrence: the carried
        //   c', d' are zeroed first, which makes the general step equal to c0/b0, d0/b0 exactly.
#pragma unroll
        for (int t = 0; t < CH; t++) {
            const int s = s0 + t;
            const int code = (cpack[t >> 3] >> (4 * (t & 7))) & 0xF;
            const int kind = code & 3;
            const bool is_int = kind == ROW_INTERIOR;
            const bool restart = kind == ROW_START || kind == ROW_SKIP;
            const bool vfree = (code & ROW_VELFREE) != 0, tfree = (code & ROW_TEMPFREE) != 0;
            const bool is_start = kind == ROW_START, is_end = kind == ROW_END;
            const R q = st0[t];
            const R a_v = is_int ? (-q - p.vis_v) : ((is_end && vfree) ? R(-1) : R(0));
            const R c_v = is_int ? (q - p.vis_v) : ((is_start && vfree) ? R(-1) : R(0));
            const R b_v = is_int ? p.b_v : (vfree ? R(2) : R(1));
            const R a_t = is_int ? (-q - p.vis_t) : ((is_end && tfree) ? R(-1) : R(0));
            const R c_t = is_int ? (q - p.vis_t) : ((is_start && tfree) ? R(-1) : R(0));
            const R b_t = is_int ? p.b_t : (tfree ? R(2) : R(1));
            const R dT = ldsD[(size_t)s * 64 + lane];
            if (restart) { cp_v = R(0); cp_t = R(0); dp[0] = R(0); dp[1] = R(0); dp[2] = R(0); dp[3] = R(0); }
            const R den_v = b_v - a_v * cp_v;                 // Algorithms.h:30-31
            const R den_t = b_t - a_t * cp_t;
            const R n0 = st1[t] - dp[0] * a_v;
            const R n1 = st2[t] - dp[1] * a_v;
            const R n2 = st3[t] - dp[2] * a_v;
            const R n3 = dT - dp[3] * a_t;
            cp_v = c_v / den_v; cp_t = c_t / den_t;
            dp[0] = n0 / den_v; dp[1] = n1 / den_v; dp[2] = n2 / den_v; dp[3] = n3 / den_t;
            st0[t] = cp_v; st1[t] = dp[0]; st2[t] = dp[1]; st3[t] = dp[2];
            ldsD[(size_t)s * 64 + lane] = dp[3];
            ldsC[(size_t)s * 64 + lane] = cp_t;
        }
        relay[0 * 64 + lane] = cp_v; relay[1 * 64 + lane] = cp_t;
        relay[2 * 64 + lane] = dp[0]; relay[3 * 64 + lane] = dp[1];
        relay[4 * 64 + lane] = dp[2]; relay[5 * 64 + lane] = dp[3];
    }
    for (int i = w; i < PIPE_NW; i++) __syncthreads();

    // ------------------------------------------------------------------ B: backward relay (registers/LDS only)
    for (int i = 0; i < PIPE_NW - 1 - w; i++) __syncthreads();
    {
        R x[4] = {R(0), R(0), R(0), R(0)};
        if (w < PIPE_NW - 1) {
            x[0] = relay[0 * 64 + lane]; x[1] = relay[1 * 64 + lane];
            x[2] = relay[2 * 64 + lane]; x[3] = relay[3 * 64 + lane];
        }
#pragma unroll
        for (int t = CH - 1; t >= 0; t--) {
            const int s = s0 + t;
            const int kind = (cpack[t >> 3] >> (4 * (t & 7))) & 3;
            const R c_v = st0[t], c_t = ldsC[(size_t)s * 64 + lane];
            const R e0 = st1[t], e1 = st2[t], e2 = st3[t], e3 = ldsD[(size_t)s * 64 + lane];
            // x[num-1] = d[num-1] (Algorithms.h:34): END and SKIP rows do not look at x[i+1]
            if (kind == ROW_END || kind == ROW_SKIP) { x[0] = R(0); x[1] = R(0); x[2] = R(0); x[3] = R(0); }
            x[0] = e0 - c_v * x[0]; x[1] = e1 - c_v * x[1];   // Algorithms.h:36-37
            x[2] = e2 - c_v * x[2]; x[3] = e3 - c_t * x[3];
            st0[t] = x[3]; st1[t] = x[0]; st2[t] = x[1]; st3[t] = x[2];   // x replaces c',d'
        }
        relay[0 * 64 + lane] = x[0]; relay[1 * 64 + lane] = x[1];
        relay[2 * 64 + lane] = x[2]; relay[3 * 64 + lane] = x[3];
    }
    for (int i = PIPE_NW - 1 - w; i < PIPE_NW - 1; i++) __syncthreads();

    // ------------------------------------------------------------------ O: scatter + merge (all waves in parallel)
    if (lane_valid) {
#pragma unroll
        for (int t = 0; t < CH; t++) {
            const int s = s0 + t;
            if (s < n) {
                const long long us = ub + (long long)s * ss;
                const int kind = (cpack[t >> 3] >> (4 * (t & 7))) & 3;
                const R x0 = st1[t], x1 = st2[t], x2 = st3[t], x3 = st0[t];
                if (kind != ROW_SKIP) {
                    (p.next(0) + us)[vo] = x0; (p.next(1) + us)[vo] = x1; (p.next(2) + us)[vo] = x2; (p.next(3) + us)[vo] = x3;
                }
                if (p.merge) {
                    const bool is_in = (inmask >> t) & 1u;
                    R tv[4] = {(p.temp(0) + us)[vo], (p.temp(1) + us)[vo], (p.temp(2) + us)[vo], (p.temp(3) + us)[vo]};
                    if (is_in) {
                        R xv[4] = {x0, x1, x2, x3};
                        // NODE_IN cell outside every segment (run without a closing cell,
                        // Grid3D.cpp:87-117): the reference merges the stale `next` value
                        if (kind == ROW_SKIP) { xv[0] = (p.next(0) + us)[vo]; xv[1] = (p.next(1) + us)[vo]; xv[2] = (p.next(2) + us)[vo]; xv[3] = (p.next(3) + us)[vo]; }
#pragma unroll
                        for (int v = 0; v < 4; v++) {
                            tv[v] = (tv[v] + xv[v]) / R(2);
                            if (p.merge == 2) tv[v] = (tv[v] + xv[v]) / R(2);
                        }
                    }
                    (p.temp_out(0) + us)[vo] = tv[0]; (p.temp_out(1) + us)[vo] = tv[1]; (p.temp_out(2) + us)[vo] = tv[2]; (p.temp_out(3) + us)[vo] = tv[3];
                }
            }
            if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // 4 cells = 16 loads in flight per lane, no more
        }
    }
}

template <typename R, int DIR, int CH>
static bool launch_one(fs3d_ctx *c, const SweepParams<R> &p)
{
    const int la_len = DIR == 2 ? p.dimy : p.dimz;
    const int n_o = DIR == 0 ? p.dimy : p.dimx;
    const int n_tiles = (la_len + 63) / 64;
    const size_t lds = ((size_t)2 * PIPE_NW * CH * 64 + 6 * 64) * sizeof(R);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)k_sweep_pipe<R, DIR, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return false;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_sweep_pipe<R, DIR, CH>), dim3((unsigned)(n_o * n_tiles)), dim3(PIPE_NW * 64), lds, c->stream, p, n_o, n_tiles);
    return true;
}

template <typename R, int CH>
static bool launch_dir(fs3d_ctx *c, int dir, const SweepParams<R> &p)
{
    switch (dir) {
    case 0: return launch_one<R, 0, CH>(c, p);
    case 1: return launch_one<R, 1, CH>(c, p);
    default: return launch_one<R, 2, CH>(c, p);
    }
}

// false: the line is longer than NW*CH cells for every instantiated CH -> caller falls back to the LINE kernel
template <>
bool launch_sweep_pipe<float>(fs3d_ctx *c, int dir, const SweepParams<float> &p)
{
    const int n = dir == 0 ? p.dimx : (dir == 1 ? p.dimy : p.dimz);
    if (n <= PIPE_NW * 16) return launch_dir<float, 16>(c, dir, p);
    if (n <= PIPE_NW * 32) return launch_dir<float, 32>(c, dir, p);
    return false;
}

template <>
bool launch_sweep_pipe<double>(fs3d_ctx *c, int dir, const SweepParams<double> &p)
{
    const int n = dir == 0 ? p.dimx : (dir == 1 ? p.dimy : p.dimz);
    if (n <= PIPE_NW * 16) return launch_dir<double, 16>(c, dir, p);
    return false;
}
