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
//   P  all waves in parallel, FIELD-MAJOR: one field of the chunk at a time is pulled into a
//      register array (CH+2 loads in flight per lane), consumed, and its registers recycled.
//      Z sweep: the chunk is moved as a [64 lines][CH cells] tile in whole 128-byte rows
//      (16 bytes per lane) and transposed through a padded LDS tile, so HBM sees cache lines.
//      Result per cell: q, dU, dV, dW in registers, dT in LDS (rows of fs3d_rows.h).
//   F  forward elimination as a relay: wave 0 eliminates its chunk, hands (c',d') of its
//      last cell to wave 1 through LDS, ... The recurrence is the reference's, cell by cell;
//      c'_uvw,d'_U,d'_V,d'_W overwrite the row data in registers, c'_T,d'_T live in LDS.
//   B  back-substitution as the reverse relay, registers/LDS only: x overwrites c',d'.
//   O  all waves in parallel, field-major: scatter x to `next` (UpdateSegment,
//      AdiSolver3D.cpp:707-730) and apply the merge into temp (TimeLayer3D.h:415-436).
// Nothing but the 8 input and 8 output words per cell (+2-byte cell code) moves to/from HBM:
// the 6 words/cell of c',d' that a thread-per-line kernel spills stay on chip (128 VGPRs
// per lane + 128 KiB LDS per workgroup for a 256-cell fp32 line).
#include <algorithm>
#include <type_traits>
#include <utility>
#include "fs3d_rows.h"

#define PIPE_NW 8
#ifndef FS3D_Z_TILE_STORE
#define FS3D_Z_TILE_STORE 1   // Z sweep: scatter through the LDS tile in whole 64-byte row pieces
#endif


// Compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>).  The sub-pass loops of the
// P and O phases are too large for `#pragma unroll` (the unroller gives up past its size threshold, the cell
// indices of the register arrays turn dynamic and the arrays land in scratch memory): instantiate them instead.
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// Raw buffer access: address = descriptor base (4 SGPRs) + soffset (1 SGPR, wave-uniform row) +
// voffset (1 VGPR, per-lane byte offset).  One SGPR per row instead of a 64-bit pointer, no 64-bit
// VALU address arithmetic, and a store is masked by an out-of-range voffset instead of a branch.
template <typename R> struct Buf;
template <> struct Buf<float> {
    static __device__ __forceinline__ float ld(rsrc_t r, unsigned vo, unsigned so) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0)); }
    static __device__ __forceinline__ void st(rsrc_t r, unsigned vo, unsigned so, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, vo, so, 0); }
};
template <> struct Buf<double> {
    static __device__ __forceinline__ double ld(rsrc_t r, unsigned vo, unsigned so) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0)); }
    static __device__ __forceinline__ void st(rsrc_t r, unsigned vo, unsigned so, double v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, vo, so, 0); }
};
#define BUF_OOB 0xFFFFFFFFu      // voffset >= num_records: the hardware drops the store

// Geometry of one wave's chunk and the accessors for "one field of PC consecutive cells of the chunk".
// P and O phases walk the chunk in sub-passes of PC cells so that their transient register arrays stay small.
template <typename R, int DIR, int CH>
struct Chunk {
    static constexpr int VW = 16 / sizeof(R);          // elements per 16-byte vector
#ifndef FS3D_PC_BYTES_XY
#define FS3D_PC_BYTES_XY 64
#endif
    static constexpr int PCB = DIR == 2 ? 64 : FS3D_PC_BYTES_XY;       // bytes of a line per sub-pass (Z: 64-byte row pieces)
    static constexpr int PC = PCB / sizeof(R) < CH ? PCB / sizeof(R) : CH;   // cells per sub-pass
    static constexpr int NPASS = CH / PC;
    static constexpr int PR = PC / VW;                 // 16-byte pieces per tile row
    static constexpr int RPI = 64 / PR;                // tile rows covered by one wave-wide vector access
    static constexpr int TSTRIDE = PC + 1;             // padded LDS row (conflict-free column access)
    static constexpr int TILE_ELEMS = 66 * TSTRIDE;    // 64 lines + the line below + the line above

    int n, s0, lane;
    bool lane_valid;
    unsigned row0;             // byte offset (inside a field incl. its leading halo plane) of (lane 0, cell 0)
    unsigned ssb;              // byte stride along the sweep
    unsigned vob;              // per-lane byte offset, clamped into the tile for lanes past the lane axis: loads are
                               // UNCONDITIONAL (a predicated load costs a branch and an s_waitcnt vmcnt(0) at its join)
    unsigned fbytes;           // bytes of one field incl. both halo planes (descriptor range)
    int dimz, rows_valid;      // Z: row pitch (elements), number of valid tile rows
    bool zvec;                 // Z: vector/LDS-transposed path usable (dimz % VW == 0)
    R *tile;                   // Z: this wave's [66][PC+1] LDS tile

    // ONE descriptor per layer (4 SGPRs) instead of one per field: a field is addressed by adding its byte
    // offset fo = v * fsb to the wave-uniform offset.  21 descriptors would not fit the SGPR file.
    unsigned fsb;              // bytes between consecutive fields of a layer
    __device__ __forceinline__ rsrc_t layer(const R *first_owned, long long plane) const
    {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(first_owned - plane), 0, (int)(4u * fsb), 0x00020000);
    }
    // byte offset of cell s0+t of lane 0, cell index clamped into the line
    __device__ __forceinline__ unsigned soff(int t) const
    {
        int s = s0 + t;
        s = s < 0 ? 0 : (s > n - 1 ? n - 1 : s);
        return row0 + (unsigned)s * ssb;
    }
    __device__ __forceinline__ bool cell_ok(int t) const { return s0 + t < n; }

    // values of a field (+ uniform byte offset dub) at cells [c0, c0+PC) of the chunk -> out[0..PC).
    // Cells past the end of the line and lanes past the lane axis receive clamped (valid, meaningless) data.
    // EDGES (Z only): also fetch the lines just below/above the tile into tile rows 64 and 65.
    template <bool EDGES = false>
    __device__ __forceinline__ void load(rsrc_t f, int dub, int c0, R (&out)[PC]) const
    {
        __builtin_amdgcn_sched_barrier(0);   // the previous field's consumers stay above this field's loads (register budget)
        if (DIR == 2 && zvec) {
            const int piece = lane % PR, rsub = lane / PR;
            int pos = s0 + c0 + piece * VW;                 // first cell of this lane's 16-byte piece
            pos = pos > n - VW ? n - VW : pos;
            u32x4 v[PR], ve;
#pragma unroll
            for (int r = 0; r < PR; r++) {
                int row = r * RPI + rsub;
                row = row > rows_valid - 1 ? rows_valid - 1 : row;
                v[r] = __builtin_amdgcn_raw_buffer_load_b128(f, (unsigned)(row * dimz + pos) * (unsigned)sizeof(R), row0 + dub, 0);
            }
            if (EDGES) {
                // lanes [0,PR): the line below the tile (row -1); lanes [PR,2PR): the line above (row rows_valid)
                const int erow = rsub == 0 ? 0 : rows_valid + 1;
                ve = __builtin_amdgcn_raw_buffer_load_b128(f, (unsigned)(erow * dimz + pos) * (unsigned)sizeof(R),
                                                           row0 + dub - (unsigned)dimz * (unsigned)sizeof(R), 0);
            }
            __builtin_amdgcn_sched_barrier(0);   // every load of the field is in flight before the first consumer
            R *const trow = tile + (rsub * TSTRIDE + piece * VW);      // + r*RPI*TSTRIDE + k: immediate offsets
            R *const tcol = tile + lane * TSTRIDE;
#pragma unroll
            for (int r = 0; r < PR; r++) {
                R e[VW];
                __builtin_memcpy(e, &v[r], 16);
#pragma unroll
                for (int k = 0; k < VW; k++) trow[r * RPI * TSTRIDE + k] = e[k];
            }
            if (EDGES && rsub < 2) {
                R e[VW];
                __builtin_memcpy(e, &ve, 16);
#pragma unroll
                for (int k = 0; k < VW; k++) trow[64 * TSTRIDE + k] = e[k];
            }
#pragma unroll
            for (int t = 0; t < PC; t++) out[t] = tcol[t];
        } else {
#pragma unroll
            for (int t = 0; t < PC; t++) out[t] = Buf<R>::ld(f, vob, soff(c0 + t) + dub);
            __builtin_amdgcn_sched_barrier(0);   // every load of the field is in flight before the first consumer
        }
    }
    // the two cells just outside [c0, c0+PC) (clamped into the line)
    __device__ __forceinline__ void load_halo(rsrc_t f, unsigned fo, int c0, R &lo, R &hi) const
    {
        lo = Buf<R>::ld(f, vob, soff(c0 - 1) + fo);
        hi = Buf<R>::ld(f, vob, soff(c0 + PC) + fo);
    }
    // one cell of this lane (any field-relative uniform byte offset)
    __device__ __forceinline__ R at(rsrc_t f, unsigned so) const { return Buf<R>::ld(f, vob, so); }

    struct Keep { int unused; };
    static __device__ __forceinline__ void pin(Keep &) {}
    // scatter in[0..PC) to cells [c0, c0+PC) of a field where wmask bit t is set (all: every valid cell is written).
    // Z sweep, all cells written: transpose through the LDS tile and store element-wide, each wave-instruction
    // covering 64/PC whole tile rows of PC contiguous cells (full 64-byte segments).
    __device__ __forceinline__ void store(rsrc_t f, unsigned fo, int c0, const R (&in)[PC], unsigned wmask, bool all, Keep &) const
    {
        if (DIR == 2 && zvec && all && FS3D_Z_TILE_STORE) {
            constexpr int RPS = 64 / PC;                        // tile rows per store instruction
            const int col = lane % PC, rsub = lane / PC;
            R *const tcol = tile + lane * TSTRIDE;
            R *const trow = tile + (rsub * TSTRIDE + col);
#pragma unroll
            for (int t = 0; t < PC; t++) tcol[t] = in[t];
            const bool col_ok = s0 + c0 + col < n;
#pragma unroll
            for (int r = 0; r < PC; r++) {
                const int row = r * RPS + rsub;
                const R val = trow[r * RPS * TSTRIDE];
                const bool ok = col_ok && row < rows_valid;
                Buf<R>::st(f, ok ? (unsigned)(row * dimz + s0 + c0 + col) * (unsigned)sizeof(R) : BUF_OOB, row0 + fo, val);
            }
        } else {
#pragma unroll
            for (int t = 0; t < PC; t++) {
                const bool ok = lane_valid && cell_ok(c0 + t) && ((wmask >> t) & 1u);
                Buf<R>::st(f, ok ? vob : BUF_OOB, soff(c0 + t) + fo, in[t]);
            }
        }
    }
    // central difference along the sweep, in place: a[t] <- (a[t+1] - a[t-1]) / two_ds   (TimeLayer3D.h:338-340)
    __device__ __forceinline__ static void deriv_inplace(R (&a)[PC], R a_lo, R a_hi, R two_ds)
    {
        R prev = a_lo;
#pragma unroll
        for (int t = 0; t < PC; t++) {
            const R cur = a[t];
            const R nxt = t == PC - 1 ? a_hi : a[t == PC - 1 ? t : t + 1];
            a[t] = (nxt - prev) / two_ds;
            prev = cur;
        }
    }
};

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
    const int tile_id = lb / n_o, o = lb - tile_id * n_o;   // tile-major: consecutive ids = consecutive planes

    const int n = DIR == 0 ? p.dimx : (DIR == 1 ? p.dimy : p.dimz);
    const int la_len = DIR == 2 ? p.dimy : p.dimz;          // length of the lane axis
    const int l = tile_id * 64 + lane;
    const long long so = DIR == 0 ? (long long)p.dimz : p.plane;
    const int vsl = DIR == 2 ? p.dimz : 1;                  // per-lane offset step of a lane-axis neighbour
    const bool hi_edge = lane == 63 || l + 1 >= la_len;     // right lane neighbour not in this wave
    const bool lo_edge = lane == 0;

    // LDS: [NW*CH][64] d_T / d'_T | [NW*CH][64] c'_T (P and O phases: per-wave transposition tiles) | relay
    constexpr size_t LDS_D = (size_t)PIPE_NW * CH * 64;
    // Per-wave transposition tiles (Z sweep, P and O phases).  A tile must not overlap the c'_T rows of
    // ANOTHER wave: wave w starts writing its own c'_T rows in its forward turn while later waves may
    // still be building rows.  If a tile fits into the wave's own c'_T row range it lives there,
    // otherwise the tiles get a region of their own behind the c'_T rows.
    constexpr size_t TILE = Chunk<R, DIR, CH>::TILE_ELEMS;
    constexpr bool TILE_IN_ROWS = TILE <= (size_t)CH * 64;
    constexpr size_t LDS_C = (size_t)PIPE_NW * CH * 64 + (TILE_IN_ROWS ? 0 : (size_t)PIPE_NW * TILE);
    R *ldsD = (R *)smem_raw;
    R *ldsC = ldsD + LDS_D;
    R *relay = ldsC + LDS_C;                                // 8 x 64 forward (c', d' per pass), reused 4 x 64 backward

    Chunk<R, DIR, CH> ck;
    ck.n = n; ck.s0 = w * CH; ck.lane = lane; ck.lane_valid = l < la_len;
    {
        // wave-uniform element offset of (lane 0, cell 0) from the first owned cell, then + the halo plane
        const long long ub = DIR == 0 ? (long long)o * p.dimz + tile_id * 64
                           : (DIR == 1 ? (long long)o * p.plane + tile_id * 64 : (long long)o * p.plane + (long long)tile_id * 64 * p.dimz);
        ck.row0 = (unsigned)((ub + p.plane) * (long long)sizeof(R));
        const long long ss = DIR == 0 ? p.plane : (DIR == 1 ? (long long)p.dimz : 1LL);
        ck.ssb = (unsigned)(ss * (long long)sizeof(R));
        const int lc = l < la_len ? lane : la_len - 1 - tile_id * 64;   // clamp lanes past the lane axis
        ck.vob = (unsigned)(DIR == 2 ? lc * p.dimz : lc) * (unsigned)sizeof(R);
        ck.fbytes = (unsigned)((p.nstride + 2 * p.plane) * (long long)sizeof(R));
        ck.fsb = (unsigned)(p.fstride * (long long)sizeof(R));
    }
    ck.dimz = p.dimz;
    ck.rows_valid = la_len - tile_id * 64 < 64 ? la_len - tile_id * 64 : 64;
    ck.zvec = (p.dimz % Chunk<R, DIR, CH>::VW) == 0;
    ck.tile = TILE_IN_ROWS ? ldsC + (size_t)w * CH * 64 : ldsC + (size_t)PIPE_NW * CH * 64 + (size_t)w * TILE;
    const int s0 = ck.s0;
    const bool lane_valid = ck.lane_valid;
    // this thread's column of the c'_T / d_T arrays: cell t of the chunk is at myX[t * 64] (immediate DS offsets)
    R *const myD = ldsD + ((size_t)s0 * 64 + lane);
    R *const myC = ldsC + ((size_t)s0 * 64 + lane);
    const int sob = (int)(so * (long long)sizeof(R));           // byte step to the neighbouring `o` plane/row
    const unsigned vslb = (unsigned)vsl * (unsigned)sizeof(R);  // byte step to a lane-axis neighbour
    // node values / cell codes have no halo plane: same offsets minus one plane
    const unsigned nsb = (unsigned)(p.nstride * (long long)sizeof(R));      // bytes between the node-value fields
    const rsrc_t rNode = __builtin_amdgcn_make_buffer_rsrc((void *)(p.node_ - p.plane), 0, (int)(3u * nsb + ck.fbytes), 0x00020000);
    const unsigned fsb = ck.fsb;
    const rsrc_t Lcur = ck.layer(p.cur_, p.plane), Ltmp = ck.layer(p.temp_, p.plane);
    const rsrc_t Lnext = ck.layer(p.next_, p.plane), Ltout = ck.layer(p.temp_out_, p.plane);
    const rsrc_t rCode = __builtin_amdgcn_make_buffer_rsrc((void *)(p.code - p.plane), 0, (int)(ck.fbytes / (sizeof(R) / 2)), 0x00020000);

    // per-cell register storage: q -> c'_uvw -> x_T ; dU,dV,dW -> d'_U,d'_V,d'_W -> x_U,x_V,x_W
    R st0[CH], st1[CH], st2[CH], st3[CH];
    unsigned cpack[(CH + 7) / 8];
    unsigned inmask = 0, segmask = 0, intmask = 0;         // NODE_IN cells; cells on a segment; INTERIOR rows (CH <= 32)
    unsigned umask = 0;                                    // wave-uniform: INTERIOR on all lines of the bundle
#pragma unroll
    for (int i = 0; i < (CH + 7) / 8; i++) cpack[i] = 0;

    // measurement only (fs3d_profile_sweep): 8 s_memtime stamps per wave
    unsigned long long *stamp = p.stamps ? p.stamps + ((size_t)blockIdx.x * 8 + w) * 8 : nullptr;
#define STAMP(k) do { if (stamp && lane == 0) stamp[k] = __builtin_amdgcn_s_memtime(); } while (0)
    STAMP(0);

    // ------------------------------------------------------------------ P: rows, field by field, PC cells per pass
    typedef Chunk<R, DIR, CH> CK;
    constexpr int PC = CK::PC;
    {
        // cell codes of the whole chunk
#pragma unroll
        for (int t = 0; t < CH; t++) {
            // unconditional load, then masked arithmetically (a select would be turned back into a branch)
            int cw = __builtin_amdgcn_raw_buffer_load_b16(rCode, ck.vob / (sizeof(R) / 2), ck.soff(t) / (sizeof(R) / 2), 0);
            cw &= -(int)(lane_valid && s0 + t < n);
            const int code = (cw >> (4 * DIR)) & 0xF;
            cpack[t >> 3] |= (unsigned)code << (4 * (t & 7));
            if (((cw >> CODE_TYPE_SHIFT) & 3) == FS3D_NODE_IN && lane_valid && s0 + t < n) inmask |= 1u << t;
            if ((code & 3) != ROW_SKIP) segmask |= 1u << t;
            if ((code & 3) == ROW_INTERIOR) intmask |= 1u << t;
        }
        {
            // cells that are INTERIOR rows on every line of the bundle: AND over the lanes (lanes past the lane axis
            // do not care).  Wave-uniform, so the fast paths below are plain scalar branches.
            unsigned m = lane_valid ? intmask : 0xFFFFFFFFu;
#pragma unroll
            for (int k = 1; k < 64; k <<= 1) m &= (unsigned)__shfl_xor((int)m, k, 64);
            umask = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
        }
        const R two_ds = p.two_ds[DIR];
        constexpr int M1 = DIR == 0 ? 1 : 0;          // axis of the `o` neighbours
        constexpr int M2 = DIR == 2 ? 1 : 2;          // axis of the lane neighbours
        const rsrc_t tS = Ltmp;
        const unsigned foS = (unsigned)DIR * fsb;
        // One sub-pass of PC cells: the INTERIOR row of every cell (BuildMatrix, AdiSolver3D.cpp:732-802), no
        // row-kind tests and no node-value loads; p_fix below replaces the rows of the other kinds.
        auto p_pass = [&](const int c0) __attribute__((always_inline)) {
            R gS[PC], x1[PC], x2[PC];                  // d(Vs)/ds, d(Vs)/d(o axis), d(Vs)/d(lane axis)
            R q[PC];
            {
                // advecting component Vs = temp[DIR]: q, its s-derivative and its lane-axis derivative
                R a_lo, a_hi;
                ck.template load<true>(tS, foS, c0, gS);
                ck.load_halo(tS, foS, c0, a_lo, a_hi);
#pragma unroll
                for (int t = 0; t < PC; t++) {
                    // lane-axis neighbours of Vs.  X/Y: the same rows one element left/right (same cache lines
                    // as the centre load).  Z: neighbouring lanes; the wave's two edge lanes read the edge rows.
                    R l_lo, l_hi;
                    if (DIR == 2 && ck.zvec) {
                        l_lo = __shfl_up(gS[t], 1, 64); l_hi = __shfl_down(gS[t], 1, 64);
                        const R e_lo = ck.tile[64 * CK::TSTRIDE + t], e_hi = ck.tile[65 * CK::TSTRIDE + t];
                        l_lo = lo_edge ? e_lo : l_lo;
                        l_hi = hi_edge ? e_hi : l_hi;
                    } else {
                        l_lo = ck.at(tS, ck.soff(c0 + t) + foS - vslb);
                        l_hi = ck.at(tS, ck.soff(c0 + t) + foS + vslb);
                    }
                    q[t] = gS[t] / two_ds;                          // temp->Vs / (2*ds)
                    x2[t] = (l_hi - l_lo) / p.two_ds[M2];
                }
                CK::deriv_inplace(gS, a_lo, a_hi, two_ds);
            }
            {
                R c[PC];
                ck.load(tS, (int)foS + sob, c0, x1);
                ck.load(tS, (int)foS - sob, c0, c);
#pragma unroll
                for (int t = 0; t < PC; t++) x1[t] = (x1[t] - c[t]) / p.two_ds[M1];
            }
            // DissFunc{X,Y,Z} (TimeLayer3D.h:554-588): (((tU + tV) + tW) + g_M1*x1) + g_M2*x2, summed in that order
            R acc[PC];
#pragma unroll
            for (int v = 0; v < 3; v++) {
                R g[PC];
                if (v == DIR) {
#pragma unroll
                    for (int t = 0; t < PC; t++) g[t] = gS[t];
                } else {
                    R a_lo, a_hi;
                    ck.load(Ltmp, (int)(v * fsb), c0, g);
                    ck.load_halo(Ltmp, v * fsb, c0, a_lo, a_hi);
                    CK::deriv_inplace(g, a_lo, a_hi, two_ds);
                }
#pragma unroll
                for (int t = 0; t < PC; t++) {
                    const R term = v == DIR ? (R(2) * g[t]) * g[t] : g[t] * g[t];
                    acc[t] = v == 0 ? term : acc[t] + term;
                    if (v == M1) x1[t] = g[t] * x1[t];
                    if (v == M2) x2[t] = g[t] * x2[t];
                }
            }
#pragma unroll
            for (int t = 0; t < PC; t++) acc[t] = p.t_phi * ((acc[t] + x1[t]) + x2[t]);   // t_phi * DissFunc
            // temperature: gradient along s (momentum RHS, AdiSolver3D.cpp:766/781/796)
            R gT[PC];
            {
                R a_lo, a_hi;
                ck.load(Ltmp, (int)(3 * fsb), c0, gT);
                ck.load_halo(Ltmp, 3 * fsb, c0, a_lo, a_hi);
                CK::deriv_inplace(gT, a_lo, a_hi, two_ds);
#pragma unroll
                for (int t = 0; t < PC; t++) gT[t] = p.v_T * gT[t];
            }
            {
                // T right-hand side -> LDS
                R cT[PC];
                ck.load(Lcur, (int)(3 * fsb), c0, cT);
#pragma unroll
                for (int t = 0; t < PC; t++) {
                    myD[(c0 + t) * 64] = cT[t] * R(3) / p.dt + acc[t];
                }
            }
#pragma unroll
            for (int v = 0; v < 3; v++) {
                R cV[PC];
                ck.load(Lcur, (int)(v * fsb), c0, cV);
#pragma unroll
                for (int t = 0; t < PC; t++) {
                    R d = cV[t] * R(3) / p.dt;
                    if (v == DIR) d = d - gT[t];
                    if (v == 0) st1[c0 + t] = d;
                    if (v == 1) st2[c0 + t] = d;
                    if (v == 2) st3[c0 + t] = d;
                }
            }
#pragma unroll
            for (int t = 0; t < PC; t++) st0[c0 + t] = q[t];
            __builtin_amdgcn_sched_barrier(0);   // pass boundary
        };
        // Rows that are not INTERIOR (segment ends, cells off every segment): replace what p_pass computed.
        //   START/END  d = node value (NOSLIP) or 0 (FREE) (ApplyBC0/1, AdiSolver3D.cpp:804-852);  SKIP  d = 0;  q = 0.
        // Rare (first/last wave of a line, obstacles): node values are fetched for the whole sub-pass unconditionally.
        auto p_fix = [&](const int c0) __attribute__((always_inline)) {
#pragma unroll
            for (int v = 0; v < 4; v++) {
                R nv[PC];
#pragma unroll
                for (int t = 0; t < PC; t++) nv[t] = ck.at(rNode, ck.soff(c0 + t) + v * nsb);
#pragma unroll
                for (int t = 0; t < PC; t++) {
                    const int code = (cpack[(c0 + t) >> 3] >> (4 * ((c0 + t) & 7))) & 0xF;
                    const int kind = code & 3;
                    const bool is_int = kind == ROW_INTERIOR;
                    const bool noslip = kind != ROW_SKIP && !(code & (v == 3 ? ROW_TEMPFREE : ROW_VELFREE));
                    const R d = noslip ? nv[t] : R(0);
                    if (v == 0) st1[c0 + t] = is_int ? st1[c0 + t] : d;
                    if (v == 1) st2[c0 + t] = is_int ? st2[c0 + t] : d;
                    if (v == 2) { st3[c0 + t] = is_int ? st3[c0 + t] : d; st0[c0 + t] = is_int ? st0[c0 + t] : R(0); }
                    if (v == 3) { if (!is_int) myD[(c0 + t) * 64] = d; }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        static_for<CK::NPASS>([&](auto pass_c) __attribute__((always_inline)) {
            constexpr int c0 = decltype(pass_c)::value * PC;
            constexpr unsigned PCM = PC >= 32 ? 0xFFFFFFFFu : ((1u << PC) - 1u);
            p_pass(c0);                                              // INTERIOR rows for every cell
            if (((umask >> c0) & PCM) != PCM) p_fix(c0);            // wave-uniform: some line has another row kind here
        });
    }

    // ------------------------------------------------------------------ F: forward relays, staggered
    // The four right-hand sides (T, U, V, W) are four independent first-order recurrences once each pass
    // carries its own c' (U, V, W repeat the shared c'_uvw recurrence: same inputs, same operations, same
    // rounding -> identical values).  Each is a relay over the waves; wave w runs pass k in turn w + k, so
    // up to four waves (on different SIMDs) advance at the same time and the forward phase takes
    // PIPE_NW + 3 turns of one short pass instead of PIPE_NW turns of one long one.
    // Chain body per cell and pass, branch-free (the row kinds only steer selects):
    //   INTERIOR a = -q - vis, b = 3/dt + 2 vis, c = q - vis      (AdiSolver3D.cpp:760-762)
    //   START    a = 0,  FREE: b = 2, c = -1 ; NOSLIP: b = 1, c = 0 (ApplyBC0, :804-827)
    //   END      c = 0,  FREE: a = -1, b = 2 ; NOSLIP: a = 0, b = 1 (ApplyBC1, :829-852)
    //   SKIP     identity row.  START and SKIP rows have a = 0, so the general step
    //   c' = c/(b - a c'), d' = (d - d' a)/(b - a c')  (Algorithms.h:28-32) gives c0/b0, d0/b0 for them whatever
    //   (finite) c', d' the previous segment left behind: b - 0*c' = b and d - d'*0 = d exactly (the sign of a
    //   zero d aside).  Likewise END and SKIP rows have c = 0 -> c' = 0 -> x = d' in the back-substitution.
    // Live-range split by hand: every stored row value passes through a register here.  The rows that the
    // register allocator parked in scratch during the P phase (its pressure peak) are reloaded now, in bulk and
    // while the wave waits for its turn anyway, instead of one by one inside the serial chain.
#define SPLIT8(a, o) asm volatile("" : "+v"(a[o]), "+v"(a[o + 1]), "+v"(a[o + 2]), "+v"(a[o + 3]), "+v"(a[o + 4]), "+v"(a[o + 5]), "+v"(a[o + 6]), "+v"(a[o + 7]))
#pragma unroll
    for (int o = 0; o < CH; o += 8) { SPLIT8(st0, o); SPLIT8(st1, o); SPLIT8(st2, o); SPLIT8(st3, o); }
    STAMP(1);
    for (int i = 0; i < w; i++) __syncthreads();
    STAMP(2);
#define FWD_COEF(VAR, FASTC)                                                                              \
    {                                                                                                     \
        const R q = st0[g + i];                                                                           \
        if (FASTC) { a4[i] = -q - vis; c4[i] = q - vis; b4[i] = bb; }                                     \
        else {                                                                                            \
            unsigned cw = cpack[(g + i) >> 3];                                                            \
            asm volatile("" : "+v"(cw));      /* opaque: no decode results carried from pass to pass */  \
            const int code = (cw >> (4 * ((g + i) & 7))) & 0xF;                                           \
            const int kind = code & 3;                                                                    \
            const bool is_int = kind == ROW_INTERIOR;                                                     \
            const bool fr = (code & (VAR == 3 ? ROW_TEMPFREE : ROW_VELFREE)) != 0;                        \
            a4[i] = is_int ? (-q - vis) : ((kind == ROW_END && fr) ? R(-1) : R(0));                       \
            c4[i] = is_int ? (q - vis) : ((kind == ROW_START && fr) ? R(-1) : R(0));                      \
            b4[i] = is_int ? bb : (fr ? R(2) : R(1));                                                     \
        }                                                                                                 \
    }
#define FWD_PASS(VAR, DREAD, DWRITE, CWRITE)                                                              \
    {                                                                                                     \
        R cp = R(0), dp = R(0);                                                                           \
        if (w > 0) { cp = relay[(2 * VAR) * 64 + lane]; dp = relay[(2 * VAR + 1) * 64 + lane]; }          \
        R vis = VAR == 3 ? p.vis_t : p.vis_v, bb = VAR == 3 ? p.b_t : p.b_v;                              \
        /* opaque per pass: otherwise U computes every a, c once and parks them in scratch for V and W */ \
        asm volatile("" : "+s"(vis), "+s"(bb));                                                           \
        _Pragma("unroll") for (int g = 0; g < CH; g += 4) {                                               \
            R a4[4], b4[4], c4[4];                                                                        \
            /* the group's q values become available only with the chain state of the previous group:   \
               otherwise the coefficients of the whole chunk are computed up front (96 live registers) */ \
            asm volatile("" : "+v"(cp), "+v"(dp), "+v"(st0[g]), "+v"(st0[g + 1]), "+v"(st0[g + 2]), "+v"(st0[g + 3])); \
            /* only the coefficients differ between the two arms; the chain below is common code */      \
            if (((umask >> g) & 0xFu) == 0xFu) {                                                          \
                _Pragma("unroll") for (int i = 0; i < 4; i++) FWD_COEF(VAR, true)                         \
            } else {                                                                                      \
                _Pragma("unroll") for (int i = 0; i < 4; i++) FWD_COEF(VAR, false)                        \
            }                                                                                             \
            _Pragma("unroll") for (int t = g; t < g + 4; t++) {                                           \
                const R a = a4[t - g], b = b4[t - g], c = c4[t - g];                                      \
                const R d = DREAD;                                                                        \
                const R den = b - a * cp;                                                                 \
                const R num = d - dp * a;                                                                 \
                cp = c / den;                                                                             \
                dp = num / den;                                                                           \
                DWRITE;                                                                                   \
                CWRITE;                                                                                   \
            }                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                            \
        }                                                                                                 \
        relay[(2 * VAR) * 64 + lane] = cp; relay[(2 * VAR + 1) * 64 + lane] = dp;                         \
    }
    FWD_PASS(3, myD[t * 64], myD[t * 64] = dp, myC[t * 64] = cp)
    __syncthreads();
    FWD_PASS(0, st1[t], st1[t] = dp, (void)0)
    __syncthreads();
    FWD_PASS(1, st2[t], st2[t] = dp, (void)0)
    __syncthreads();
    FWD_PASS(2, st3[t], st3[t] = dp, st0[t] = cp)          // last pass over the cell: c'_uvw replaces q
#undef FWD_PASS
#undef FWD_COEF
    STAMP(3);
    for (int i = w + 3; i < PIPE_NW + 3; i++) __syncthreads();

    // ------------------------------------------------------------------ B: backward relay (registers/LDS only)
    for (int i = 0; i < PIPE_NW - 1 - w; i++) __syncthreads();
    STAMP(4);
    {
        R x[4] = {R(0), R(0), R(0), R(0)};
        if (w < PIPE_NW - 1) {
            x[0] = relay[0 * 64 + lane]; x[1] = relay[1 * 64 + lane];
            x[2] = relay[2 * 64 + lane]; x[3] = relay[3 * 64 + lane];
        }
#pragma unroll
        for (int t = CH - 1; t >= 0; t--) {
            const R c_v = st0[t], c_t = myC[t * 64];
            const R e0 = st1[t], e1 = st2[t], e2 = st3[t], e3 = myD[t * 64];
            // x[num-1] = d[num-1] (Algorithms.h:34): END and SKIP rows carry c' = 0 and so do not look at x[i+1]
            x[0] = e0 - c_v * x[0]; x[1] = e1 - c_v * x[1];   // Algorithms.h:36-37
            x[2] = e2 - c_v * x[2]; x[3] = e3 - c_t * x[3];
            st0[t] = x[3]; st1[t] = x[0]; st2[t] = x[1]; st3[t] = x[2];   // x replaces c',d'
            if ((t & 3) == 0) __builtin_amdgcn_sched_barrier(0);
        }
        relay[0 * 64 + lane] = x[0]; relay[1 * 64 + lane] = x[1];
        relay[2 * 64 + lane] = x[2]; relay[3 * 64 + lane] = x[3];
    }
    STAMP(5);
    // the PIPE_NW-1 hand-off barriers, plus one more: ldsC becomes the transposition tiles again
    for (int i = PIPE_NW - 1 - w; i < PIPE_NW; i++) __syncthreads();
    STAMP(6);

    // ------------------------------------------------------------------ O: scatter + merge, field by field, PC cells per pass
    {
        // does every valid cell of the chunk sit on a segment?  (then whole tiles can be stored)
        const int len = n - s0 < 0 ? 0 : (n - s0 > CH ? CH : n - s0);
        const unsigned chunk_mask = len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u);
        const bool all_seg = __all((!lane_valid) || ((segmask & chunk_mask) == chunk_mask));
        static_for<CK::NPASS>([&](auto pass_c) __attribute__((always_inline)) {
            constexpr int c0 = decltype(pass_c)::value * PC;
            const unsigned seg_p = segmask >> c0, in_p = inmask >> c0;
            typename CK::Keep keepN, keepT;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                R xv[PC];
#pragma unroll
                for (int t = 0; t < PC; t++) xv[t] = v == 0 ? st1[c0 + t] : (v == 1 ? st2[c0 + t] : (v == 2 ? st3[c0 + t] : st0[c0 + t]));
                ck.store(Lnext, v * fsb, c0, xv, seg_p, all_seg, keepN);
                if (p.merge) {
                    R tv[PC];
                    ck.load(Ltmp, (int)(v * fsb), c0, tv);
                    CK::pin(keepN); CK::pin(keepT);      // the load returned: every older store has fetched its data
                    if (((in_p & ~seg_p) & (PC >= 32 ? 0xFFFFFFFFu : ((1u << PC) - 1u))) != 0) {
                        // NODE_IN cell outside every segment (run without a closing cell,
                        // Grid3D.cpp:87-117): the reference merges the stale `next` value
#pragma unroll
                        for (int t = 0; t < PC; t++)
                            if ((in_p >> t) & ~(seg_p >> t) & 1u) xv[t] = ck.at(Lnext, ck.soff(c0 + t) + v * fsb);
                    }
#pragma unroll
                    for (int t = 0; t < PC; t++) {
                        if ((in_p >> t) & 1u) {
                            tv[t] = (tv[t] + xv[t]) / R(2);
                            if (p.merge == 2) tv[t] = (tv[t] + xv[t]) / R(2);
                        }
                    }
                    ck.store(Ltout, v * fsb, c0, tv, 0xFFFFFFFFu, true, keepT);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    }
    STAMP(7);
#undef STAMP
}

template <typename R, int DIR, int CH>
static bool launch_one(fs3d_ctx *c, const SweepParams<R> &p)
{
    const int la_len = DIR == 2 ? p.dimy : p.dimz;
    const int n_o = DIR == 0 ? p.dimy : p.dimx;
    const int n_tiles = (la_len + 63) / 64;
    const size_t tile = Chunk<R, DIR, CH>::TILE_ELEMS;
    const size_t lds_c = (size_t)PIPE_NW * CH * 64 + (tile <= (size_t)CH * 64 ? 0 : (size_t)PIPE_NW * tile);
    const size_t lds = ((size_t)PIPE_NW * CH * 64 + lds_c + 8 * 64) * sizeof(R);
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
    if ((unsigned long long)p.fstride * 4ull * sizeof(float) >= (1ull << 32)) return false;   // 32-bit buffer offsets span a layer
    if (n <= PIPE_NW * 16) return launch_dir<float, 16>(c, dir, p);
    if (n <= PIPE_NW * 32) return launch_dir<float, 32>(c, dir, p);
    return false;
}

template <>
bool launch_sweep_pipe<double>(fs3d_ctx *c, int dir, const SweepParams<double> &p)
{
    const int n = dir == 0 ? p.dimx : (dir == 1 ? p.dimy : p.dimz);
    if ((unsigned long long)p.fstride * 4ull * sizeof(double) >= (1ull << 32)) return false;
    if (n <= PIPE_NW * 16) return launch_dir<double, 16>(c, dir, p);
    return false;
}
