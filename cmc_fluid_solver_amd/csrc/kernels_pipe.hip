// Wave-pipelined exact Thomas sweep (FS3D_SWEEP_PIPE) for CDNA4.
//
// Work decomposition
//   bundle    = 64 neighbouring grid lines of the sweep direction (the lanes of a wave), whole length n = dim along
//               the sweep axis (or a segment / slab piece of it, MODE 1/2).  One workgroup per bundle.
//   workgroup = NW = 8 waves; wave w owns CH cells of every line of the bundle as CH/PC pieces of PC = 16 cells:
//               piece h of wave w = cells [h*NW*PC + w*PC, +PC).  Consecutive pieces belong to consecutive waves.
//   X, Y sweeps: lanes run along k (unit stride)  -> every global access is a coalesced row.
//   Z sweep    : lanes run along j, a piece is 64 contiguous bytes of a row: [64 lines][16 cells] tiles move in
//                16-byte lane accesses and are transposed through a padded per-wave LDS tile.
//
// Phases (every floating-point operation on a cell is the sequential reference's, Algorithms.h:21-38)
//   P  all waves in parallel, one sub-pass per piece, software pipeline over nine field groups (group k+1 issued
//      before group k is consumed): the INTERIOR row of every cell -> q, dU, dV, dW in registers, dT in LDS;
//      a wave-uniform, rare fix-up pass for the other row kinds (segment ends, cells off every segment).
//   F  forward elimination: four staggered passes (T, U, V, W) over the pieces in line order; a pass of a piece
//      starts when the same pass of the piece before it has finished (LDS flag, no workgroup barrier).
//      c'_uvw, d'_U, d'_V, d'_W overwrite the row data in registers, c'_T, d'_T live in LDS.
//      fp32: divisions by the scaling-free core of the IEEE expansion, per-bundle redo flag (FM).
//   B  back-substitution over the pieces in reverse order, registers/LDS only: x overwrites c', d'.
//   O  all waves in parallel: scatter x to `next` (UpdateSegment, AdiSolver3D.cpp:707-730) and apply the merge
//      into temp (TimeLayer3D.h:415-436).
// A whole sweep moves nothing but the 8 input and 8 output words per cell (+2-byte cell code) to/from HBM: the
// 6 words/cell of c', d' that a thread-per-line kernel spills stay on chip (128 VGPRs per lane + 128 KiB LDS per
// workgroup for 256 fp32 cells).  The halves (MODE 1/2) pass them through an HBM scratch instead: lines of any
// length, x-slabs of a multi-GPU run.
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <utility>
#include "fs3d_rows.h"

#define PIPE_NW 8             // waves per workgroup of a whole sweep; the halves of short slab pieces use fewer (NW)
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

// Point-to-point hand-off between the waves of a workgroup through an LDS flag (the relay phases): the
// producer publishes its data, then the flag; the consumer polls the flag, then reads.  DS operations of a wave
// execute in order; the fences keep the compiler from moving accesses across the flag.  The poll is bounded:
// a lost hand-off never hangs the GPU; it sets bit 0 of the context's error word (g_errw, pinned host memory), which
// every synchronising entry point turns into FS3D_ERR_HIP -- wrong numbers are never returned as a success.
// The flags are accessed through explicit LDS (address space 3) pointers: a volatile access through a generic
// pointer is compiled to a FLAT instruction with system-scope cache bits and a vmcnt(0) wait behind it, several
// times the latency of the ds_read / ds_write these turn into.
typedef __attribute__((address_space(3))) volatile int lds_flag_t;
struct FlagCtl { int *errw; int bound; };      // per-kernel: where a timed-out wait reports, and after how many polls
__device__ __forceinline__ void flag_wait(volatile int *f_, const FlagCtl &fc)
{
    lds_flag_t *f = (lds_flag_t *)f_;
    int guard = 0;
    while (__builtin_amdgcn_readfirstlane(*f) == 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++guard > fc.bound) {
            if (fc.errw && (threadIdx.x & 63) == 0) __hip_atomic_fetch_or(fc.errw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void flag_set(volatile int *f_)
{
    lds_flag_t *f = (lds_flag_t *)f_;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    *f = 1;
}

// ---- fp32 division without the scaling steps --------------------------------------------------------------
// The IEEE expansion of x / y on this ISA is  ys = div_scale(y), xs = div_scale(x), r0 = rcp(ys), r = r0 + r0*(1 - ys*r0),
// q0 = xs*r, q1 = q0 + r*(xs - ys*q0), q2 = div_fmas(xs - ys*q1, r, q1), div_fixup(q2, y, x).  When neither operand
// needs scaling (no denormal operand or quotient, no overflow, x not tiny) div_scale returns its operand, div_fmas
// is a plain fma and div_fixup returns q2: the result is exactly the eight-operation core below.  The three
// scaling/fix-up instructions cost more than the core and, through VCC, keep two divisions from overlapping.
// `operand_plain(v)`: v == 0 or |v| >= 2^-100 (a zero numerator gives the correctly signed-magnitude zero in the
// core as well; the sign of a zero result may differ, as elsewhere).  Callers compute with the core, AND the
// operand tests of a whole group of cells together, and redo the group with the full division when any lane
// failed -- wave-uniform and rare (values between 0 and 8e-31).  Divisors are checked against [2^-30, 2^60) on
// the host (constants) or per cell (den); finite fields below 2^60 are assumed (the reference aborts far earlier).
template <typename R> struct DivC { R y, r; };          // divisor and its refined reciprocal
__device__ __forceinline__ float recip_refined(float y)
{
    float r = __builtin_amdgcn_rcpf(y);
    const float e = __builtin_fmaf(-y, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float div_core(float x, float y, float r)
{
    float q = x * r;
    float rem = __builtin_fmaf(-y, q, x);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-y, q, x);
    return __builtin_fmaf(rem, r, q);
}
// The operand tests are running minima (two VALU operations per operand, one register each, no lane masks):
//   numerators: (bits << 1) - 2, which wraps a zero to the top;  divisors: bits << 1.  Compared once at the end.
struct DivGuard {
    // numerators >= 2^-100 (or zero); divisors in [2^-30, 2^26): with both, every quotient the core forms is a normal number or zero and
    // the core's roundings are those of the IEEE division.  One running value each: the minimum of the numerators' exponent fields,
    // the maximum of (divisor exponent field - low bound) as unsigned -- a divisor below the low bound (or zero) wraps to a huge value
    // and fails the same test as one above the high bound.
    static constexpr unsigned DEN_LO = 2u * (97u << 23), DEN_HI = 2u * (153u << 23);      // 2^-30, 2^26
    unsigned num_min = 0xFFFFFFFFu, den_rng = 0u;
    __device__ __forceinline__ void num(float v) { const unsigned t = (__builtin_bit_cast(unsigned, v) << 1) - 2u; num_min = t < num_min ? t : num_min; asm volatile("" : "+v"(num_min)); }   // opaque: reduce now, do not keep every t alive until the end
    __device__ __forceinline__ void den(float v) { const unsigned t = (__builtin_bit_cast(unsigned, v) << 1) - DEN_LO; den_rng = t > den_rng ? t : den_rng; asm volatile("" : "+v"(den_rng)); }
    __device__ __forceinline__ void num(double) {}
    __device__ __forceinline__ void den(double) {}
    __device__ __forceinline__ bool plain() const { return num_min >= 2u * (27u << 23) - 2u && den_rng < DEN_HI - DEN_LO; }
};
// x / c.y: FASTM -> core + operand test; otherwise the full division
template <bool FASTM> __device__ __forceinline__ float divc(float x, const DivC<float> &c, DivGuard &ok)
{
    if (FASTM) { ok.num(x); return div_core(x, c.y, c.r); }
    return x / c.y;
}
template <bool FASTM> __device__ __forceinline__ double divc(double x, const DivC<double> &c, DivGuard &) { return x / c.y; }
__device__ __forceinline__ DivC<float> mkdiv(float y) { return {y, recip_refined(y)}; }
__device__ __forceinline__ DivC<double> mkdiv(double y) { return {y, 0.0}; }
// one cell of the forward chain with the division core: c' = c/den, d' = num/den share the reciprocal of den
// GUARD_DEN: U, V and W divide by the same den values, the first of them checks; c is checked once, in the P phase
template <bool GUARD_DEN>
__device__ __forceinline__ void chain_core(float c, float num, float den, float &cp, float &dp, DivGuard &ok)
{
    if (GUARD_DEN) ok.den(den);
    ok.num(num);
    const float r = recip_refined(den);
    float qc = c * r, qd = num * r;
    float rc = __builtin_fmaf(-den, qc, c), rd = __builtin_fmaf(-den, qd, num);
    qc = __builtin_fmaf(rc, r, qc); qd = __builtin_fmaf(rd, r, qd);
    rc = __builtin_fmaf(-den, qc, c); rd = __builtin_fmaf(-den, qd, num);
    cp = __builtin_fmaf(rc, r, qc); dp = __builtin_fmaf(rd, r, qd);
}
template <bool GUARD_DEN>
__device__ __forceinline__ void chain_core(double c, double num, double den, double &cp, double &dp, DivGuard &) { cp = c / den; dp = num / den; }

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// Raw buffer access: address = descriptor base (4 SGPRs) + soffset (1 SGPR, wave-uniform row) +
// voffset (1 VGPR, per-lane byte offset).  One SGPR per row instead of a 64-bit pointer, no 64-bit
// VALU address arithmetic, and a store is masked by an out-of-range voffset instead of a branch.
// Cache policy (aux operand: 2 = nt): `cur` is read once per sweep, `next` / `temp_out` are written once and read by the next launch
// only (see kernels_part.hip); the temp loads stay cached (neighbour rows, second read of the O phase).  Hints only: same bits.
#ifndef PIPE_AUX_CUR
#define PIPE_AUX_CUR 2
#endif
#ifndef PIPE_AUX_ST
#define PIPE_AUX_ST 2
#endif
template <typename R> struct Buf;
template <> struct Buf<float> {
    template <int AUX = 0> static __device__ __forceinline__ float ld(rsrc_t r, unsigned vo, unsigned so) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, AUX)); }
    template <int AUX = 0> static __device__ __forceinline__ void st(rsrc_t r, unsigned vo, unsigned so, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, vo, so, AUX); }
};
template <> struct Buf<double> {
    template <int AUX = 0> static __device__ __forceinline__ double ld(rsrc_t r, unsigned vo, unsigned so) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, AUX)); }
    template <int AUX = 0> static __device__ __forceinline__ void st(rsrc_t r, unsigned vo, unsigned so, double v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, vo, so, AUX); }
};
#define BUF_OOB 0xFFFFFFFFu      // voffset >= num_records: the hardware drops the store

// Geometry of one wave's chunk and the accessors for "one field of PC consecutive cells of the chunk".
// P and O phases walk the chunk in sub-passes of PC cells so that their transient register arrays stay small.
template <typename R, int DIR, int CH, int NW = PIPE_NW>
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

    int n, lane;
    int wpc;                   // w * PC: first cell of this wave's first piece
    int slo, shi;              // clamp range of the cell index along the sweep: [0, n-1], or one more where a ghost plane of a neighbouring slab stands
    bool lane_valid;
    unsigned row0;             // byte offset (inside a field incl. its leading halo plane) of (lane 0, cell 0)
    unsigned ssb;              // byte stride along the sweep
    unsigned vob;              // per-lane byte offset, clamped into the tile for lanes past the lane axis: loads are
                               // UNCONDITIONAL (a predicated load costs a branch and an s_waitcnt vmcnt(0) at its join)
    unsigned vob_st;           // vob for valid lanes, BUF_OOB for lanes past the lane axis: stores of cells known to be in the line
    unsigned vob_live;         // vob_st, and BUF_OOB for dead lines too (see `dead` in the kernel): stores of solved values
    bool dead, any_dead;       // this lane's whole line is neither solved nor merged / some lane of the wave is such a line
    unsigned fbytes;           // bytes of one field incl. both halo planes (descriptor range)
    int dimz, rows_valid;      // Z: row pitch (elements), number of valid tile rows
    R *tile;                   // Z: this wave's [66][PC+1] LDS tile

    // ONE descriptor per layer (4 SGPRs) instead of one per field: a field is addressed by adding its byte
    // offset fo = v * fsb to the wave-uniform offset.  21 descriptors would not fit the SGPR file.
    unsigned fsb;              // bytes between consecutive fields of a layer
    __device__ __forceinline__ rsrc_t layer(const R *first_owned, long long plane) const
    {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(first_owned - plane), 0, (int)(4u * fsb), 0x00020000);
    }
    // The CH cells of a wave are CH/PC pieces of PC cells; piece h of wave w is cells [h*NW*PC + w*PC, +PC) of the
    // line: consecutive pieces of the line belong to consecutive waves (wave 2m and 2m+1 share the 128-byte lines
    // of a Z sweep and work on them in the same sub-pass), and the relays hand over every PC cells.
    // Local cell index lt in [0, CH) -> piece lt / PC, position lt % PC.
    __device__ __forceinline__ int base(int c0) const { return (c0 / PC) * (NW * PC) + wpc; }   // c0: first local cell of a piece
    __device__ __forceinline__ int cell(int lt) const { return base(lt - lt % PC) + lt % PC; }
    // byte offset of local cell lt of lane 0, cell index clamped into the line
    __device__ __forceinline__ unsigned soff(int lt) const
    {
        int s = cell(lt);
        s = s < 0 ? 0 : (s > n - 1 ? n - 1 : s);
        return row0 + (unsigned)s * ssb;
    }
    // cell base(c0) + dt of a layer field, dt in [-1, PC]: where a neighbouring slab's ghost plane stands before /
    // behind the owned planes the index may leave [0, n-1] by one (layer fields only: they carry halo planes)
    __device__ __forceinline__ unsigned soff_halo(int c0, int dt) const
    {
        int s = base(c0) + dt;
        s = s < slo ? slo : (s > shi ? shi : s);
        return row0 + (unsigned)s * ssb;
    }
    __device__ __forceinline__ bool cell_ok(int lt) const { return cell(lt) < n; }

    // One field of a sub-pass in flight: X/Y hold the PC values themselves, Z the raw 16-byte row pieces that
    // still have to go through the transposition tile.  `issue` only starts the loads, `land` delivers out[0..PC):
    // the P phase issues the next field before it lands and consumes the current one (software pipeline).
    struct Raw { R v[DIR == 2 ? 1 : PC]; u32x4 q[DIR == 2 ? PR : 1]; u32x4 qe; R lo, hi; };
    // values of a field (+ uniform byte offset dub) at cells [c0, c0+PC) of the chunk.
    // Cells past the end of the line and lanes past the lane axis receive clamped (valid, meaningless) data.
    // EDGES (Z only): also fetch the lines just below/above the tile (-> tile rows 64 and 65 in land()).
    // HALO: also the two cells just outside [c0, c0+PC) on the own line (clamped into the line).
    // AUX: cache policy of the loads (PIPE_AUX_CUR for the read-once `cur` fields)
    template <bool EDGES, bool HALO, int AUX = 0>
    __device__ __forceinline__ void issue(rsrc_t f, int dub, int c0, Raw &r) const
    {
        if (DIR == 2) {
            const int piece = lane % PR, rsub = lane / PR;
            int pos = base(c0) + piece * VW;                // first cell of this lane's 16-byte piece
            pos = pos > n - VW ? n - VW : pos;
#pragma unroll
            for (int i = 0; i < PR; i++) {
                int row = i * RPI + rsub;
                row = row > rows_valid - 1 ? rows_valid - 1 : row;
                r.q[i] = __builtin_amdgcn_raw_buffer_load_b128(f, (unsigned)(row * dimz + pos) * (unsigned)sizeof(R), row0 + dub, AUX);
            }
            if (EDGES) {
                // lanes [0,PR): the line below the tile (row -1); lanes [PR,2PR): the line above (row rows_valid)
                const int erow = rsub == 0 ? 0 : rows_valid + 1;
                r.qe = __builtin_amdgcn_raw_buffer_load_b128(f, (unsigned)(erow * dimz + pos) * (unsigned)sizeof(R),
                                                             row0 + dub - (unsigned)dimz * (unsigned)sizeof(R), 0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < PC; t++) r.v[t] = Buf<R>::template ld<AUX>(f, vob, soff_halo(c0, t) + dub);   // cell n of a slab may be a neighbour's ghost plane
        }
        if (HALO) {
            r.lo = Buf<R>::ld(f, vob, soff_halo(c0, -1) + dub);
            r.hi = Buf<R>::ld(f, vob, soff_halo(c0, PC) + dub);
        }
    }
    template <bool EDGES>
    __device__ __forceinline__ void land(const Raw &r, R (&out)[PC]) const
    {
        if (DIR == 2) {
            const int piece = lane % PR, rsub = lane / PR;
            R *const trow = tile + (rsub * TSTRIDE + piece * VW);      // + i*RPI*TSTRIDE + k: immediate offsets
            R *const tcol = tile + lane * TSTRIDE;
#pragma unroll
            for (int i = 0; i < PR; i++) {
                R e[VW];
                __builtin_memcpy(e, &r.q[i], 16);
#pragma unroll
                for (int k = 0; k < VW; k++) trow[i * RPI * TSTRIDE + k] = e[k];
            }
            if (EDGES && rsub < 2) {
                R e[VW];
                __builtin_memcpy(e, &r.qe, 16);
#pragma unroll
                for (int k = 0; k < VW; k++) trow[64 * TSTRIDE + k] = e[k];
            }
#pragma unroll
            for (int t = 0; t < PC; t++) out[t] = tcol[t];
        } else {
#pragma unroll
            for (int t = 0; t < PC; t++) out[t] = r.v[t];
        }
    }
    // issue + land in one go (O phase)
    template <bool EDGES = false>
    __device__ __forceinline__ void load(rsrc_t f, int dub, int c0, R (&out)[PC]) const
    {
        Raw r;
        __builtin_amdgcn_sched_barrier(0);   // the previous field's consumers stay above this field's loads (register budget)
        issue<EDGES, false>(f, dub, c0, r);
        __builtin_amdgcn_sched_barrier(0);   // every load of the field is in flight before the first consumer
        land<EDGES>(r, out);
    }
    // one cell of this lane (any field-relative uniform byte offset)
    __device__ __forceinline__ R at(rsrc_t f, unsigned so) const { return Buf<R>::ld(f, vob, so); }

    // scatter in[0..PC) to cells [c0, c0+PC) of a field where wmask bit t is set (all: every valid cell is written).
    // Z sweep, all cells written: transpose through the LDS tile and store element-wide, each wave-instruction
    // covering 64/PC whole tile rows of PC contiguous cells (full 64-byte segments).
    __device__ __forceinline__ void store(rsrc_t f, unsigned fo, int c0, const R (&in)[PC], unsigned wmask, bool all) const
    {
        if (DIR == 2 && FS3D_Z_TILE_STORE) {
            constexpr int RPS = 64 / PC;                        // tile rows per store instruction
            const int col = lane % PC, rsub = lane / PC;
            R *const tcol = tile + lane * TSTRIDE;
            R *const trow = tile + (rsub * TSTRIDE + col);
            // Not every cell written (lines off the segments, e.g. the wall lines of a box): the write masks go
            // through the tile as well -- lane (row, col) needs bit col of row's mask.  Storing such sub-passes per
            // lane instead (4 bytes per 1 KB row) made the `next` stores of a Z sweep cost 5x those of the other sweeps.
            unsigned tmask = 0xFFFFFFFFu;                       // bit r: this lane's element of store instruction r is written
            if (!all) {
                typedef typename std::conditional<sizeof(R) == 4, unsigned, unsigned long long>::type bits_t;
                tcol[0] = __builtin_bit_cast(R, (bits_t)wmask);
                tmask = 0;
#pragma unroll
                for (int r = 0; r < PC; r++) {
                    const unsigned m = (unsigned)__builtin_bit_cast(bits_t, trow[r * RPS * TSTRIDE - col]);
                    tmask |= ((m >> col) & 1u) << r;
                }
            }
#pragma unroll
            for (int t = 0; t < PC; t++) tcol[t] = in[t];
            const bool col_ok = base(c0) + col < n;
#pragma unroll
            for (int r = 0; r < PC; r++) {
                const int row = r * RPS + rsub;
                const R val = trow[r * RPS * TSTRIDE];
                const bool ok = col_ok && row < rows_valid && ((tmask >> r) & 1u);
                Buf<R>::template st<PIPE_AUX_ST>(f, ok ? (unsigned)(row * dimz + base(c0) + col) * (unsigned)sizeof(R) : BUF_OOB, row0 + fo, val);
            }
        } else {
#pragma unroll
            for (int t = 0; t < PC; t++) {
                const bool ok = lane_valid && cell_ok(c0 + t) && ((wmask >> t) & 1u);
                Buf<R>::template st<PIPE_AUX_ST>(f, ok ? vob : BUF_OOB, soff(c0 + t) + fo, in[t]);
            }
        }
    }
    // store of a sub-pass whose cells are all inside the line and all written (wave-uniform fast path of the O phase)
    // live_only: solved values -- nothing is written on dead lines
    __device__ __forceinline__ void store_plain(rsrc_t f, unsigned fo, int c0, const R (&in)[PC], bool live_only = false) const
    {
        if (DIR == 2 && FS3D_Z_TILE_STORE) store(f, fo, c0, in, dead ? 0u : 0xFFFFFFFFu, !(live_only && any_dead));
        else {
            const unsigned vo = live_only ? vob_live : vob_st;
#pragma unroll
            for (int t = 0; t < PC; t++) Buf<R>::template st<PIPE_AUX_ST>(f, vo, soff(c0 + t) + fo, in[t]);
        }
    }
    // central difference along the sweep, in place: a[t] <- (a[t+1] - a[t-1]) / two_ds   (TimeLayer3D.h:338-340)
    template <bool FASTM>
    __device__ __forceinline__ static void deriv_inplace(R (&a)[PC], R a_lo, R a_hi, const DivC<R> &two_ds, DivGuard &ok)
    {
        R prev = a_lo;
#pragma unroll
        for (int t = 0; t < PC; t++) {
            const R cur = a[t];
            const R nxt = t == PC - 1 ? a_hi : a[t == PC - 1 ? t : t + 1];
            a[t] = divc<FASTM>(nxt - prev, two_ds, ok);
            prev = cur;
            if (FASTM && (t & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // the core has no VCC to serialise it: bound the overlap
        }
    }
};

// one workgroup = one bundle
// FM (fp32 only): divide with the scaling-free core; a workgroup that met an operand outside the core's range
// raises redo[bundle] and the FM = false instance, launched right behind with the same arguments, computes that
// bundle again with full divisions (every other workgroup of it returns at once).
// MODE (X sweep of an x-slab, SS6 of DESIGN.md): 0 whole sweep; 1 forward half -- the recurrences start from the
// carries of the slab below (p.carry_in) and the eliminated rows go to the HBM scratch, carries of the last cell to
// p.carry_out; 2 backward half -- rows from the scratch, x of the slab above (p.xcarry_in), then the O phase.
// Same operations on the same values as the whole sweep: a line cut into slabs gives the uncut line's numbers.
template <typename R, int DIR, int CH, bool FM, int MODE = 0, int NW = PIPE_NW>
__global__ void __launch_bounds__(NW * 64, 2) k_sweep_pipe(SweepParams<R> p, int n_o, int n_tiles, int *redo)
{
    if (!FM && redo && redo[blockIdx.x] == 0) return;      // redo pass: this bundle was fine
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    DivGuard ok;                                            // FM: were all division operands so far in the core's range?
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
    // tile-major: consecutive ids = consecutive planes.  Slab halves: row-major from p.bundle0 on, so that a block of
    // bundles is a contiguous range of lines in the carry arrays (line = j*dimz + k, 64 per bundle).
    if (MODE != 0) lb += p.bundle0;
    const int tile_id = MODE != 0 ? lb % n_tiles : lb / n_o, o = MODE != 0 ? lb / n_tiles : lb - tile_id * n_o;

    const int n_full = DIR == 0 ? p.dimx : (DIR == 1 ? p.dimy : p.dimz);
    // slab halves may work on a segment [seg_begin, seg_begin + seg_len) of the line (lines longer than the kernel holds)
    const int n = (MODE != 0 && p.seg_len) ? p.seg_len : n_full;
    const int la_len = DIR == 2 ? p.dimy : p.dimz;          // length of the lane axis
    const int l = tile_id * 64 + lane;
    const long long so = DIR == 0 ? (long long)p.dimz : p.plane;
    const int vsl = DIR == 2 ? p.dimz : 1;                  // per-lane offset step of a lane-axis neighbour
    const bool hi_edge = lane == 63 || l + 1 >= la_len;     // right lane neighbour not in this wave
    const bool lo_edge = lane == 0;

    // LDS: [NW*CH][64] d_T / d'_T | [NW*CH][64] c'_T (P and O phases: per-wave transposition tiles) | relay
    constexpr size_t LDS_D = (size_t)NW * CH * 64;
    // Per-wave transposition tiles (Z sweep, P and O phases).  A tile must not overlap the c'_T rows of
    // ANOTHER wave: wave w starts writing its own c'_T rows in its forward turn while later waves may
    // still be building rows.  If a tile fits into the wave's own c'_T row range it lives there,
    // otherwise the tiles get a region of their own behind the c'_T rows.
    constexpr size_t TILE = Chunk<R, DIR, CH, NW>::TILE_ELEMS;
    constexpr bool TILE_IN_ROWS = TILE <= (size_t)CH * 64;
    constexpr size_t LDS_C = (size_t)NW * CH * 64 + (TILE_IN_ROWS ? 0 : (size_t)NW * TILE);
    R *ldsD = (R *)smem_raw;
    R *ldsC = ldsD + LDS_D;
    R *relay = ldsC + LDS_C;                                // 8 x 64 forward (c', d' per pass), reused 4 x 64 backward
    constexpr int NPIECE = Chunk<R, DIR, CH, NW>::NPASS;        // pieces per wave; relay step (h, w) = piece h of wave w
    volatile int *fflag = (volatile int *)(relay + 8 * 64); // [4 passes][NPIECE][NW]: forward pass k of step (h, w) is done
    volatile int *bflag = fflag + 4 * NPIECE * NW;     // [NPIECE][NW]: backward step (h, w) is done
    volatile int *live_lds = bflag + NPIECE * NW;      // [NW][2]: lanes whose line has a solved or merged cell among the wave's cells
    if (threadIdx.x < 5 * NPIECE * NW) ((lds_flag_t *)fflag)[threadIdx.x] = 0;
    const FlagCtl fctl = {p.errw, p.test_drop ? (1 << 10) : (1 << 22)};
    __syncthreads();                                        // the only workgroup-wide barrier of the kernel
    // The relay visits the waves in order, so the low waves are needed first: give them the issue slots first.
    // It also takes the waves out of lockstep (they would otherwise all wait for memory at the same time).
    if (w < 2) __builtin_amdgcn_s_setprio(3);
    else if (w < 4) __builtin_amdgcn_s_setprio(2);
    else if (w < 6) __builtin_amdgcn_s_setprio(1);

    Chunk<R, DIR, CH, NW> ck;
    ck.n = n; ck.wpc = w * Chunk<R, DIR, CH, NW>::PC; ck.lane = lane; ck.lane_valid = l < la_len;
    // one cell beyond the owned range is readable where the line goes on: a neighbouring slab's ghost plane (X) or the
    // neighbouring segment of the same line
    ck.slo = ((DIR == 0 && p.ghost_lo) || (MODE != 0 && p.seg_begin > 0)) ? -1 : 0;
    ck.shi = ((DIR == 0 && p.ghost_hi) || (MODE != 0 && p.seg_len && p.seg_begin + p.seg_len < n_full)) ? n : n - 1;
    {
        // wave-uniform element offset of (lane 0, cell 0) from the first owned cell, then + the halo plane
        const long long ub = DIR == 0 ? (long long)o * p.dimz + tile_id * 64
                           : (DIR == 1 ? (long long)o * p.plane + tile_id * 64 : (long long)o * p.plane + (long long)tile_id * 64 * p.dimz);
        const long long ss = DIR == 0 ? p.plane : (DIR == 1 ? (long long)p.dimz : 1LL);
        ck.row0 = (unsigned)((ub + p.plane + (MODE != 0 ? (long long)p.seg_begin * ss : 0LL)) * (long long)sizeof(R));
        ck.ssb = (unsigned)(ss * (long long)sizeof(R));
        const int lc = l < la_len ? lane : la_len - 1 - tile_id * 64;   // clamp lanes past the lane axis
        ck.vob = (unsigned)(DIR == 2 ? lc * p.dimz : lc) * (unsigned)sizeof(R);
        ck.fbytes = (unsigned)(p.fstride * (long long)sizeof(R));                  // a whole field: halo plane, cells, halo plane (+ padding)
        ck.fsb = (unsigned)(p.fstride * (long long)sizeof(R));
        ck.vob_st = l < la_len ? ck.vob : BUF_OOB;
    }
    ck.dimz = p.dimz;
    ck.rows_valid = la_len - tile_id * 64 < 64 ? la_len - tile_id * 64 : 64;
    ck.tile = TILE_IN_ROWS ? ldsC + (size_t)w * CH * 64 : ldsC + (size_t)NW * CH * 64 + (size_t)w * TILE;
    const int s0 = w * CH;                                      // first row of this wave in the LDS arrays (storage order, not line order)
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
    typedef Chunk<R, DIR, CH, NW> CK;
    constexpr int PC = CK::PC;
    {
        // cell codes of the whole chunk: every load in flight before the first is decoded (the decode chain would otherwise
        // be scheduled load by load, CH memory round trips in a row)
        int cwv[CH];
        // Z sweep: the codes of a lane's cells are contiguous (2 bytes each): 8 per 16-byte load where the piece lies
        // inside the line, instead of one 64-row gather per cell
        constexpr int CPV = 8;                                   // codes per 16-byte load
        bool vec_codes = DIR == 2 && PC % CPV == 0;
#pragma unroll
        for (int h = 0; h < CK::NPASS; h++) vec_codes = vec_codes && ck.base(h * PC) + PC <= n;
        if (vec_codes) {
            u32x4 cq[CH / CPV > 0 ? CH / CPV : 1];
#pragma unroll
            for (int g = 0; g < CH / CPV; g++)
                cq[g] = __builtin_amdgcn_raw_buffer_load_b128(rCode, ck.vob / (sizeof(R) / 2), ck.soff(g * CPV) / (sizeof(R) / 2), 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < CH; t++) {
                unsigned d[4];
                __builtin_memcpy(d, &cq[t / CPV], 16);
                cwv[t] = (int)((d[(t % CPV) / 2] >> (16 * (t & 1))) & 0xFFFFu);
            }
        } else {
#pragma unroll
            for (int t = 0; t < CH; t++)   // unconditional load, then masked arithmetically (a select would be turned back into a branch)
                cwv[t] = __builtin_amdgcn_raw_buffer_load_b16(rCode, ck.vob / (sizeof(R) / 2), ck.soff(t) / (sizeof(R) / 2), 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < CH; t++) {
            int cw = cwv[t];
            cw &= -(int)(lane_valid && ck.cell_ok(t));
            const int code = (cw >> (4 * DIR)) & 0xF;
            cpack[t >> 3] |= (unsigned)code << (4 * (t & 7));
            if (((cw >> CODE_TYPE_SHIFT) & 3) == FS3D_NODE_IN && lane_valid && ck.cell_ok(t)) inmask |= 1u << t;
            if ((code & 3) != ROW_SKIP) segmask |= 1u << t;
            if ((code & 3) == ROW_INTERIOR) intmask |= 1u << t;
        }
        // Dead lines: no cell of the line is on a segment or NODE_IN (the wall lines of a box) -- nothing the sweep
        // computes for them is ever stored, and temp_out only receives their old temp values.  Such lanes must not
        // keep the rest of the bundle off the fast paths: they run along with INTERIOR rows, finite garbage that
        // stays in its lane.  Whole lines only (a dead stretch may hand its recurrence to a live segment): the
        // waves combine what they see of each line through the LDS.  Halves see a part of the line: no dead lanes.
        ck.dead = false; ck.any_dead = false; ck.vob_live = ck.vob_st;
        if (MODE == 0) {
            const unsigned long long live_w = __ballot(lane_valid && ((segmask | inmask) != 0));
            if (lane == 0) { ((lds_flag_t *)live_lds)[2 * w] = (int)(unsigned)live_w; ((lds_flag_t *)live_lds)[2 * w + 1] = (int)(unsigned)(live_w >> 32); }
            __syncthreads();
            unsigned lo = 0, hi = 0;
#pragma unroll
            for (int i = 0; i < NW; i++) { lo |= (unsigned)((lds_flag_t *)live_lds)[2 * i]; hi |= (unsigned)((lds_flag_t *)live_lds)[2 * i + 1]; }
            const unsigned long long live_all = ((unsigned long long)hi << 32) | lo;
            ck.dead = lane_valid && !((live_all >> lane) & 1ull);
            ck.any_dead = __any(ck.dead);
            if (ck.dead) ck.vob_live = BUF_OOB;
        }
        {
            // cells that are INTERIOR rows on every line of the bundle: AND over the lanes (lanes past the lane axis
            // and dead lines do not care).  Wave-uniform, so the fast paths below are plain scalar branches.
            unsigned m = (lane_valid && !ck.dead) ? intmask : 0xFFFFFFFFu;
#pragma unroll
            for (int k = 1; k < 64; k <<= 1) m &= (unsigned)__shfl_xor((int)m, k, 64);
            umask = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
        }
        const R two_ds = p.two_ds[DIR];
        constexpr int M1 = DIR == 0 ? 1 : 0;          // axis of the `o` neighbours
        constexpr int M2 = DIR == 2 ? 1 : 2;          // axis of the lane neighbours
        const rsrc_t tS = Ltmp;
        const unsigned foS = (unsigned)DIR * fsb;
        const DivC<R> dS = mkdiv(two_ds), dM1 = mkdiv(p.two_ds[M1]), dM2 = mkdiv(p.two_ds[M2]), dDt = mkdiv(p.dt);
        // One sub-pass of PC cells: the INTERIOR row of every cell (BuildMatrix, AdiSolver3D.cpp:732-802), no
        // row-kind tests and no node-value loads; p_fix below replaces the rows of the other kinds.
        // Software pipeline over the nine field groups: group k+1 is issued before group k is landed and consumed,
        // so a wave always has one group of loads in flight while it computes (SB pins that order).
#define SB __builtin_amdgcn_sched_barrier(0)
        constexpr int VA = DIR == 0 ? 1 : 0;          // the two components other than the advecting one, ascending
        constexpr int VB = DIR == 2 ? 1 : 2;
        auto p_pass = [&](const int c0) __attribute__((always_inline)) {
            typedef typename CK::Raw Raw;
            R gS[PC], x1[PC], x2[PC];                  // d(Vs)/ds, d(Vs)/d(o axis), d(Vs)/d(lane axis)
            R q[PC];
            Raw rS, rP, rM;
            R l_lo[DIR == 2 ? 1 : PC], l_hi[DIR == 2 ? 1 : PC];
            // I1: advecting component Vs = temp[DIR] (centre, sweep halo, Z: edge lines), X/Y: its lane neighbours
            ck.template issue<true, true>(tS, (int)foS, c0, rS);
            if (DIR != 2) {
#pragma unroll
                for (int t = 0; t < PC; t++) {
                    // the same rows one element left/right (same cache lines as the centre load)
                    l_lo[t] = ck.at(tS, ck.soff(c0 + t) + foS - vslb);
                    l_hi[t] = ck.at(tS, ck.soff(c0 + t) + foS + vslb);
                }
            }
            SB;
            // I2: Vs on the two neighbouring `o` planes/rows
            ck.template issue<false, false>(tS, (int)foS + sob, c0, rP);
            ck.template issue<false, false>(tS, (int)foS - sob, c0, rM);
            SB;
            // C1: q, d(Vs)/d(lane axis), d(Vs)/ds
            ck.template land<true>(rS, gS);
#pragma unroll
            for (int t = 0; t < PC; t++) {
                R lo_, hi_;
                if (DIR == 2) {
                    // Z: neighbouring lanes; the wave's two edge lanes read the edge rows of the tile
                    lo_ = __shfl_up(gS[t], 1, 64); hi_ = __shfl_down(gS[t], 1, 64);
                    const R e_lo = ck.tile[64 * CK::TSTRIDE + t], e_hi = ck.tile[65 * CK::TSTRIDE + t];
                    lo_ = lo_edge ? e_lo : lo_;
                    hi_ = hi_edge ? e_hi : hi_;
                } else { lo_ = l_lo[t]; hi_ = l_hi[t]; }
                q[t] = divc<FM>(gS[t], dS, ok);                 // temp->Vs / (2*ds)
                if (FM) { ok.num(q[t] - p.vis_v); ok.num(q[t] - p.vis_t); }   // the numerators c = q - vis of the forward passes
                x2[t] = divc<FM>(hi_ - lo_, dM2, ok);
                if (FM && (t & 1) == 1) SB;
            }
            CK::template deriv_inplace<FM>(gS, rS.lo, rS.hi, dS, ok);
            SB;
            // I3: first other velocity component
            Raw rA;
            ck.template issue<false, true>(Ltmp, (int)(VA * fsb), c0, rA);
            SB;
            // C2: d(Vs)/d(o axis)
            {
                R c[PC];
                ck.template land<false>(rP, x1);
                ck.template land<false>(rM, c);
#pragma unroll
                for (int t = 0; t < PC; t++) { x1[t] = divc<FM>(x1[t] - c[t], dM1, ok); if (FM && (t & 3) == 3) SB; }
            }
            SB;
            // I4: second other velocity component
            Raw rB;
            ck.template issue<false, true>(Ltmp, (int)(VB * fsb), c0, rB);
            SB;
            // C3: d(V_a)/ds
            R gA[PC];
            ck.template land<false>(rA, gA);
            CK::template deriv_inplace<FM>(gA, rA.lo, rA.hi, dS, ok);
            SB;
            // I5: temperature (temp layer)
            Raw rT;
            ck.template issue<false, true>(Ltmp, (int)(3 * fsb), c0, rT);
            SB;
            // C4: d(V_b)/ds and DissFunc{X,Y,Z} (TimeLayer3D.h:554-588):
            //     (((tU + tV) + tW) + g_M1*x1) + g_M2*x2, summed in that order
            R acc[PC];
            {
                R gB[PC];
                ck.template land<false>(rB, gB);
                CK::template deriv_inplace<FM>(gB, rB.lo, rB.hi, dS, ok);
#pragma unroll
                for (int t = 0; t < PC; t++) {
                    const R g0 = DIR == 0 ? gS[t] : gA[t];
                    const R g1 = DIR == 1 ? gS[t] : (DIR == 0 ? gA[t] : gB[t]);
                    const R g2 = DIR == 2 ? gS[t] : gB[t];
                    const R t0 = DIR == 0 ? (R(2) * g0) * g0 : g0 * g0;
                    const R t1 = DIR == 1 ? (R(2) * g1) * g1 : g1 * g1;
                    const R t2 = DIR == 2 ? (R(2) * g2) * g2 : g2 * g2;
                    const R gm1 = M1 == 0 ? g0 : g1, gm2 = M2 == 1 ? g1 : g2;
                    const R y1 = gm1 * x1[t], y2 = gm2 * x2[t];
                    acc[t] = p.t_phi * ((((t0 + t1) + t2) + y1) + y2);   // t_phi * DissFunc
                }
            }
            SB;
            // I6: temperature (cur layer)
            Raw rCT;
            ck.template issue<false, false, PIPE_AUX_CUR>(Lcur, (int)(3 * fsb), c0, rCT);
            SB;
            // C5: temperature gradient along s (momentum RHS, AdiSolver3D.cpp:766/781/796)
            R gT[PC];
            ck.template land<false>(rT, gT);
            CK::template deriv_inplace<FM>(gT, rT.lo, rT.hi, dS, ok);
#pragma unroll
            for (int t = 0; t < PC; t++) gT[t] = p.v_T * gT[t];
            SB;
            // I7..I9 / C6..C9: the four `cur` fields -> right-hand sides
            Raw rC0, rC1, rC2;
            ck.template issue<false, false, PIPE_AUX_CUR>(Lcur, 0, c0, rC0);
            SB;
            {
                R cT[PC];
                ck.template land<false>(rCT, cT);
#pragma unroll
                for (int t = 0; t < PC; t++) { myD[(c0 + t) * 64] = divc<FM>(cT[t] * R(3), dDt, ok) + acc[t]; if (FM && (t & 3) == 3) SB; }   // T right-hand side -> LDS
            }
            SB;
            ck.template issue<false, false, PIPE_AUX_CUR>(Lcur, (int)fsb, c0, rC1);
            SB;
            {
                R cV[PC];
                ck.template land<false>(rC0, cV);
#pragma unroll
                for (int t = 0; t < PC; t++) { R d = divc<FM>(cV[t] * R(3), dDt, ok); if (DIR == 0) d = d - gT[t]; st1[c0 + t] = d; if (FM && (t & 3) == 3) SB; }
            }
            SB;
            ck.template issue<false, false, PIPE_AUX_CUR>(Lcur, (int)(2 * fsb), c0, rC2);
            SB;
            {
                R cV[PC];
                ck.template land<false>(rC1, cV);
#pragma unroll
                for (int t = 0; t < PC; t++) { R d = divc<FM>(cV[t] * R(3), dDt, ok); if (DIR == 1) d = d - gT[t]; st2[c0 + t] = d; if (FM && (t & 3) == 3) SB; }
            }
            SB;
            {
                R cV[PC];
                ck.template land<false>(rC2, cV);
#pragma unroll
                for (int t = 0; t < PC; t++) { R d = divc<FM>(cV[t] * R(3), dDt, ok); if (DIR == 2) d = d - gT[t]; st3[c0 + t] = d; if (FM && (t & 3) == 3) SB; }
            }
#pragma unroll
            for (int t = 0; t < PC; t++) st0[c0 + t] = q[t];
            SB;   // pass boundary
        };
#undef SB
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
        if (MODE != 2) {
            static_for<CK::NPASS>([&](auto pass_c) __attribute__((always_inline)) {
                constexpr int c0 = decltype(pass_c)::value * PC;
                constexpr unsigned PCM = PC >= 32 ? 0xFFFFFFFFu : ((1u << PC) - 1u);
                p_pass(c0);                                              // INTERIOR rows for every cell
                if (((umask >> c0) & PCM) != PCM) p_fix(c0);            // wave-uniform: some line has another row kind here
            });
        }
    }
    // halves: the eliminated rows travel through the HBM scratch in the bundle's own layout -- [segment][bundle][array][row of
    // this workgroup's LDS order][64 lanes] -- so that every access is a whole 64-lane row whatever the sweep direction
    constexpr long long SCR_ARR = (long long)NW * CH * 64;
    R *const scr = p.scr_ + (((long long)p.seg_index * p.scr_bundles + lb) * 6) * SCR_ARR + (long long)s0 * 64 + lane;
    const long long cline = (long long)o * la_len + l;      // this lane's line in the carry arrays [value][line]
    const long long cpitch = p.carry_pitch;                 // lines per value row

    // ------------------------------------------------------------------ F: forward relays, staggered
    // The four right-hand sides (T, U, V, W) are four independent first-order recurrences once each pass
    // carries its own c' (U, V, W repeat the shared c'_uvw recurrence: same inputs, same operations, same
    // rounding -> identical values).  Each is a relay over the waves; wave w runs pass k in turn w + k, so
    // up to four waves (on different SIMDs) advance at the same time and the forward phase takes
    // NW + 3 turns of one short pass instead of NW turns of one long one.
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
    __builtin_amdgcn_s_setprio(3);     // the serial chains are latency-critical: ahead of other waves' P/O work
    if (MODE != 2) {
    if (w > 0) flag_wait(&fflag[3 * NPIECE * NW + w - 1], fctl);
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
#define FWD_PASS(VAR, H, DREAD, DWRITE, CWRITE)                                                           \
    {                                                                                                     \
        /* step (H, w) follows (H, w-1), or (H-1, NW-1) for w = 0: the pieces in the order of the line */ \
        R cp = R(0), dp = R(0);                                                                           \
        if (w > 0 || H > 0) {                                                                             \
            flag_wait(&fflag[(VAR * NPIECE + (w > 0 ? H : H - 1)) * NW + (w > 0 ? w - 1 : NW - 1)], fctl); \
            cp = relay[(2 * VAR) * 64 + lane]; dp = relay[(2 * VAR + 1) * 64 + lane];                     \
        } else if (MODE == 1 && p.carry_in && lane_valid) {                                               \
            /* the recurrence continues the slab below (k_xsweep_fwd's carry layout) */                  \
            cp = p.carry_in[(VAR == 3 ? 1 : 0) * cpitch + cline];                                        \
            dp = p.carry_in[(2 + VAR) * cpitch + cline];                                                 \
        }                                                                                                 \
        R vis = VAR == 3 ? p.vis_t : p.vis_v, bb = VAR == 3 ? p.b_t : p.b_v;                              \
        /* opaque per pass: otherwise U computes every a, c once and parks them in scratch for V and W */ \
        asm volatile("" : "+s"(vis), "+s"(bb));                                                           \
        _Pragma("unroll") for (int g = H * PC; g < (H + 1) * PC; g += 4) {                                \
            R a4[4], b4[4], c4[4];                                                                        \
            /* the group's q values become available only with the chain state of the previous group:   \
               otherwise the coefficients of the whole chunk are computed up front (96 live registers) */ \
            asm volatile("" : "+v"(cp), "+v"(dp), "+v"(st0[g]), "+v"(st0[g + 1]), "+v"(st0[g + 2]), "+v"(st0[g + 3])); \
            /* only the coefficients differ between the two arms; the chain below is common code */      \
            _Pragma("unroll") for (int i = 0; i < 4; i++) FWD_COEF(VAR, true)                             \
            if (((umask >> g) & 0xFu) != 0xFu) {          /* rare: the interior values are overwritten in place */ \
                _Pragma("unroll") for (int i = 0; i < 4; i++) FWD_COEF(VAR, false)                        \
            }                                                                                             \
            _Pragma("unroll") for (int t = g; t < g + 4; t++) {                                           \
                const R a = a4[t - g], b = b4[t - g], c = c4[t - g];                                      \
                const R d = DREAD;                                                                        \
                const R den = b - a * cp;                                                                 \
                const R num = d - dp * a;                                                                 \
                if (FM) chain_core<VAR == 3 || VAR == 0>(c, num, den, cp, dp, ok);                        \
                else { cp = c / den; dp = num / den; }                                                    \
                DWRITE;                                                                                   \
                CWRITE;                                                                                   \
            }                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                            \
        }                                                                                                 \
        relay[(2 * VAR) * 64 + lane] = cp; relay[(2 * VAR + 1) * 64 + lane] = dp;                         \
        if (!(p.test_drop && blockIdx.x == 0 && w == 2 && VAR == 3 && H == 0))   /* test hook: a lost hand-over */ \
            flag_set(&fflag[(VAR * NPIECE + H) * NW + w]);                                           \
    }
    static_for<NPIECE>([&](auto h_c) __attribute__((always_inline)) {
        constexpr int H = decltype(h_c)::value;
        R dTv[PC];                                             // T right-hand sides of the piece: out of the LDS before the wait for the turn
#pragma unroll
        for (int t = 0; t < PC; t++) dTv[t] = myD[(H * PC + t) * 64];
        FWD_PASS(3, H, dTv[t - H * PC], myD[t * 64] = dp, myC[t * 64] = cp)
        FWD_PASS(0, H, st1[t], st1[t] = dp, (void)0)
        FWD_PASS(1, H, st2[t], st2[t] = dp, (void)0)
        FWD_PASS(2, H, st3[t], st3[t] = dp, st0[t] = cp)       // last pass over the cell: c'_uvw replaces q
    });
#undef FWD_PASS
    if (FM) { if (!__all(ok.plain() || ck.dead) && lane == 0) atomicOr(&redo[blockIdx.x], 1); }
#undef FWD_COEF
    STAMP(3);
    }   // MODE != 2
    if (MODE == 1) {
        // forward half of a slab: rows to the scratch, the carries of the slab's last cell to the next rank
#pragma unroll
        for (int t = 0; t < CH; t++) {
            scr[0 * SCR_ARR + t * 64] = st0[t]; scr[1 * SCR_ARR + t * 64] = myC[t * 64];
            scr[2 * SCR_ARR + t * 64] = st1[t]; scr[3 * SCR_ARR + t * 64] = st2[t];
            scr[4 * SCR_ARR + t * 64] = st3[t]; scr[5 * SCR_ARR + t * 64] = myD[t * 64];
        }
        // the slab's last cell n-1: piece hl of wave wl, local cell tl
        const int hl = (n - 1) / (NW * PC), rl = (n - 1) - hl * (NW * PC), wl = rl / PC, tl = hl * PC + rl % PC;
        if (w == wl && lane_valid) {
            R cv = R(0), d0 = R(0), d1 = R(0), d2 = R(0);
#pragma unroll
            for (int t = 0; t < CH; t++) if (t == tl) { cv = st0[t]; d0 = st1[t]; d1 = st2[t]; d2 = st3[t]; }
            p.carry_out[0 * cpitch + cline] = cv;  p.carry_out[1 * cpitch + cline] = myC[tl * 64];
            p.carry_out[2 * cpitch + cline] = d0;  p.carry_out[3 * cpitch + cline] = d1;
            p.carry_out[4 * cpitch + cline] = d2;  p.carry_out[5 * cpitch + cline] = myD[tl * 64];
        }
        // handled (the forward half ends here): without this the flag of a bundle that once asked for the full
        // divisions stayed up and every later forward half was computed twice
        if (!FM && redo && threadIdx.x == 0) redo[blockIdx.x] = 0;
        return;
    }
    if (MODE == 2) {
        // backward half: rows back from the scratch.  Cells past the slab get c' = -1, d' = 0: x passes through them.
#pragma unroll
        for (int t = 0; t < CH; t++) {
            const bool in = ck.cell_ok(t);
            const R a0 = scr[0 * SCR_ARR + t * 64], a1 = scr[1 * SCR_ARR + t * 64], a2 = scr[2 * SCR_ARR + t * 64];
            const R a3 = scr[3 * SCR_ARR + t * 64], a4 = scr[4 * SCR_ARR + t * 64], a5 = scr[5 * SCR_ARR + t * 64];
            st0[t] = in ? a0 : R(-1); myC[t * 64] = in ? a1 : R(-1);
            st1[t] = in ? a2 : R(0); st2[t] = in ? a3 : R(0); st3[t] = in ? a4 : R(0); myD[t * 64] = in ? a5 : R(0);
            if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ------------------------------------------------------------------ B: backward relay (registers/LDS only)
    // The last wave finishes its last forward pass after every other wave has finished all of theirs (each pass
    // of wave w waits for the same pass of wave w-1), so the relay slots are free for the way back.
    STAMP(4);
    static_for<NPIECE>([&](auto hr_c) __attribute__((always_inline)) {
        constexpr int H = NPIECE - 1 - decltype(hr_c)::value;   // pieces from the end of the line to its start
        // step (H, w) follows (H, w+1), or (H+1, 0) for the last wave
        R x[4] = {R(0), R(0), R(0), R(0)};
        // the T rows of the piece come from the LDS: requested before the wait for the turn, not inside the chain
        R ctv[PC], e3v[PC];
#pragma unroll
        for (int t = 0; t < PC; t++) { ctv[t] = myC[(H * PC + t) * 64]; e3v[t] = myD[(H * PC + t) * 64]; }
        if (w < NW - 1 || H < NPIECE - 1) {
            flag_wait(&bflag[(w < NW - 1 ? H : H + 1) * NW + (w < NW - 1 ? w + 1 : 0)], fctl);
            x[0] = relay[0 * 64 + lane]; x[1] = relay[1 * 64 + lane];
            x[2] = relay[2 * 64 + lane]; x[3] = relay[3 * 64 + lane];
        } else if (MODE == 2 && p.xcarry_in && lane_valid) {
            // x of the first cell of the slab above (k_xsweep_bwd's carry layout)
            x[0] = p.xcarry_in[0 * cpitch + cline]; x[1] = p.xcarry_in[1 * cpitch + cline];
            x[2] = p.xcarry_in[2 * cpitch + cline]; x[3] = p.xcarry_in[3 * cpitch + cline];
        }
#pragma unroll
        for (int t = (H + 1) * PC - 1; t >= H * PC; t--) {
            const R c_v = st0[t], c_t = ctv[t - H * PC];
            const R e0 = st1[t], e1 = st2[t], e2 = st3[t], e3 = e3v[t - H * PC];
            // x[num-1] = d[num-1] (Algorithms.h:34): END and SKIP rows carry c' = 0 and so do not look at x[i+1]
            x[0] = e0 - c_v * x[0]; x[1] = e1 - c_v * x[1];   // Algorithms.h:36-37
            x[2] = e2 - c_v * x[2]; x[3] = e3 - c_t * x[3];
            st0[t] = x[3]; st1[t] = x[0]; st2[t] = x[1]; st3[t] = x[2];   // x replaces c',d'
            if ((t & 3) == 0) __builtin_amdgcn_sched_barrier(0);
        }
        relay[0 * 64 + lane] = x[0]; relay[1 * 64 + lane] = x[1];
        relay[2 * 64 + lane] = x[2]; relay[3 * 64 + lane] = x[3];
        flag_set(&bflag[H * NW + w]);
        if (MODE == 2 && H == 0 && w == 0 && p.xcarry_out && lane_valid) {
            p.xcarry_out[0 * cpitch + cline] = x[0]; p.xcarry_out[1 * cpitch + cline] = x[1];
            p.xcarry_out[2 * cpitch + cline] = x[2]; p.xcarry_out[3 * cpitch + cline] = x[3];
        }
    });
    __builtin_amdgcn_s_setprio(0);
    STAMP(5);
    // no barrier: the transposition tiles of the O phase live in the wave's own c'_T rows (or its own region)
    STAMP(6);

    // ------------------------------------------------------------------ O: scatter + merge, field by field, PC cells per pass
    {
        // does every valid cell of the chunk sit on a segment?  (then whole tiles can be stored)
        unsigned chunk_mask = 0;                                // the wave's cells that lie inside the line
#pragma unroll
        for (int t = 0; t < CH; t++) if (ck.cell_ok(t)) chunk_mask |= 1u << t;
        const bool all_seg = __all((!lane_valid) || ((segmask & chunk_mask) == chunk_mask));
        static_for<CK::NPASS>([&](auto pass_c) __attribute__((always_inline)) {
            constexpr int c0 = decltype(pass_c)::value * PC;
            const unsigned seg_p = segmask >> c0, in_p = inmask >> c0;
            constexpr unsigned PCM = PC >= 32 ? 0xFFFFFFFFu : ((1u << PC) - 1u);
            if (((umask >> c0) & PCM) == PCM) {
                // every cell of the sub-pass is an INTERIOR row (hence NODE_IN and on a segment) on every line:
                // unmasked stores, unconditional merge, and the next field's temp values requested one field ahead
                typename CK::Raw rt[2];
                if (p.merge) ck.template issue<false, false>(Ltmp, 0, c0, rt[0]);
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    R xv[PC];
#pragma unroll
                    for (int t = 0; t < PC; t++) xv[t] = v == 0 ? st1[c0 + t] : (v == 1 ? st2[c0 + t] : (v == 2 ? st3[c0 + t] : st0[c0 + t]));
                    if (p.store_next) ck.store_plain(Lnext, v * fsb, c0, xv, true);
                    if (p.merge) {
                        if (v < 3) ck.template issue<false, false>(Ltmp, (int)((v + 1) * fsb), c0, rt[(v + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        R tv[PC];
                        ck.template land<false>(rt[v & 1], tv);
#pragma unroll
                        for (int t = 0; t < PC; t++) {
                            R mv = (tv[t] + xv[t]) / R(2);                           // MergeFieldTo (TimeLayer3D.h:415-436)
                            if (p.merge == 2) mv = (mv + xv[t]) / R(2);
                            tv[t] = ck.dead ? tv[t] : mv;                            // dead lines: the old temp value moves on
                        }
                        ck.store_plain(Ltout, v * fsb, c0, tv);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
#pragma unroll
            for (int v = 0; v < 4; v++) {
                R xv[PC];
#pragma unroll
                for (int t = 0; t < PC; t++) xv[t] = v == 0 ? st1[c0 + t] : (v == 1 ? st2[c0 + t] : (v == 2 ? st3[c0 + t] : st0[c0 + t]));
                if (p.store_next) ck.store(Lnext, v * fsb, c0, xv, seg_p, all_seg);
                if (p.merge) {
                    R tv[PC];
                    ck.load(Ltmp, (int)(v * fsb), c0, tv);
                    if (((in_p & ~seg_p) & PCM) != 0) {
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
                    ck.store(Ltout, v * fsb, c0, tv, 0xFFFFFFFFu, true);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    }
    STAMP(7);
#undef STAMP
    if (!FM && redo && threadIdx.x == 0) redo[blockIdx.x] = 0;   // handled: the flags are all zero again for the next sweep
}

template <typename R, int DIR, int CH>
static bool launch_one(fs3d_ctx *c, const SweepParams<R> &p)
{
    const int la_len = DIR == 2 ? p.dimy : p.dimz;
    const int n_o = DIR == 0 ? p.dimy : p.dimx;
    const int n_tiles = (la_len + 63) / 64;
    if (DIR == 2 && p.dimz % Chunk<R, DIR, CH>::VW != 0) return false;   // Z moves whole 16-byte row pieces
    const size_t tile = Chunk<R, DIR, CH>::TILE_ELEMS;
    const size_t lds_c = (size_t)PIPE_NW * CH * 64 + (tile <= (size_t)CH * 64 ? 0 : (size_t)PIPE_NW * tile);
    const size_t lds = ((size_t)PIPE_NW * CH * 64 + lds_c + 8 * 64) * sizeof(R) + (5 * Chunk<R, DIR, CH>::NPASS + 2) * PIPE_NW * sizeof(int);
    const int grid = n_o * n_tiles;
    constexpr bool HAS_FM = std::is_same<R, float>::value;
    // the attribute is per device: one bit per device id (several devices can be driven from one process)
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long dev_bit = 1ull << (c->device & 63);
    if (!(attr_set.load() & dev_bit)) {
        if (hipFuncSetAttribute((const void *)k_sweep_pipe<R, DIR, CH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return false;
        if (HAS_FM && hipFuncSetAttribute((const void *)k_sweep_pipe<R, DIR, CH, HAS_FM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return false;
        attr_set.fetch_or(dev_bit);
    }
    if (HAS_FM && p.fast_div) {
        // division core first; the full-division instance right behind it redoes the bundles that asked for it
        if (c->redo_cap < grid) {
            if (c->redo) { hipStreamSynchronize(c->stream); hipFree(c->redo); c->redo = nullptr; c->redo_cap = 0; }
            if (hipMalloc(&c->redo, (size_t)grid * sizeof(int)) != hipSuccess) return false;
            if (hipMemsetAsync(c->redo, 0, (size_t)grid * sizeof(int), c->stream) != hipSuccess) return false;
            c->redo_cap = grid;
        }
        hipLaunchKernelGGL((k_sweep_pipe<R, DIR, CH, HAS_FM>), dim3((unsigned)grid), dim3(PIPE_NW * 64), lds, c->stream, p, n_o, n_tiles, c->redo);
        hipLaunchKernelGGL((k_sweep_pipe<R, DIR, CH, false>), dim3((unsigned)grid), dim3(PIPE_NW * 64), lds, c->stream, p, n_o, n_tiles, c->redo);
    } else {
        hipLaunchKernelGGL((k_sweep_pipe<R, DIR, CH, false>), dim3((unsigned)grid), dim3(PIPE_NW * 64), lds, c->stream, p, n_o, n_tiles, (int *)nullptr);
    }
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

// ---- sweep halves: X sweep of an x-slab (cross-slab pipeline, fs3d_hip.hip: xsweep_multi) and the segments of
// ---- lines longer than one launch holds on chip (launch_sweep_pipe_segmented) ------------------------------
// the context's scratch, at least `elems` elements of R (shared with the thread-per-line kernel, which needs 6 per cell)
template <typename R>
static bool pipe_scratch(fs3d_ctx *c, size_t elems)
{
    const size_t bytes = std::max(elems, (size_t)6 * (size_t)c->nstride) * sizeof(R);
    if (c->scr && c->scr_bytes >= bytes) return true;
    if (c->scr) { hipStreamSynchronize(c->stream); hipFree(c->scr); c->scr = nullptr; c->scr_bytes = 0; }
    if (hipMalloc(&c->scr, bytes) != hipSuccess) return false;
    c->scr_bytes = bytes;
    return true;
}

template <typename R, int DIR, int CH, int NW = PIPE_NW>
static bool launch_half(fs3d_ctx *c, SweepParams<R> p, int half, int b0, int b1)
{
    constexpr bool HAS_FM = std::is_same<R, float>::value;
    const int la_len = DIR == 2 ? p.dimy : p.dimz;
    const int n_o = DIR == 0 ? p.dimy : p.dimx, n_tiles = (la_len + 63) / 64, grid = b1 - b0;
    if (DIR == 2 && p.dimz % Chunk<R, DIR, CH, NW>::VW != 0) { c->err = "pipe halves: dimz is not a multiple of the 16-byte vector"; return false; }
    const size_t tile = Chunk<R, DIR, CH, NW>::TILE_ELEMS;
    const size_t lds_c = (size_t)NW * CH * 64 + (tile <= (size_t)CH * 64 ? 0 : (size_t)NW * tile);
    const size_t lds = ((size_t)NW * CH * 64 + lds_c + 8 * 64) * sizeof(R) + (5 * Chunk<R, DIR, CH, NW>::NPASS + 2) * NW * sizeof(int);
    // the attribute is per device: one bit per device id (several devices can be driven from one process)
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long dev_bit = 1ull << (c->device & 63);
    if (!(attr_set.load() & dev_bit)) {
        if (hipFuncSetAttribute((const void *)k_sweep_pipe<R, DIR, CH, false, 1, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { c->err = std::string("pipe halves: hipFuncSetAttribute: ") + hipGetErrorString(hipGetLastError()); return false; }
        if (hipFuncSetAttribute((const void *)k_sweep_pipe<R, DIR, CH, false, 2, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        if (HAS_FM && hipFuncSetAttribute((const void *)k_sweep_pipe<R, DIR, CH, HAS_FM, 1, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        attr_set.fetch_or(dev_bit);
    }
    if (grid <= 0) return true;
    p.bundle0 = b0;
    if (!p.scr_ || p.scr_bundles < b1) { c->err = "pipe halves: scratch not prepared"; return false; }
    if (half == 1) {
        if (HAS_FM && p.fast_div) {
            if (c->redo_cap < grid) {
                if (c->redo) { hipStreamSynchronize(c->stream); hipFree(c->redo); c->redo = nullptr; c->redo_cap = 0; }
                const int cap = std::max(grid, n_o * n_tiles);
                if (hipMalloc(&c->redo, (size_t)cap * sizeof(int)) != hipSuccess) return false;
                if (hipMemsetAsync(c->redo, 0, (size_t)cap * sizeof(int), c->stream) != hipSuccess) return false;
                c->redo_cap = cap;
            }
            hipLaunchKernelGGL((k_sweep_pipe<R, DIR, CH, HAS_FM, 1, NW>), dim3((unsigned)grid), dim3(NW * 64), lds, c->stream, p, n_o, n_tiles, c->redo);
            hipLaunchKernelGGL((k_sweep_pipe<R, DIR, CH, false, 1, NW>), dim3((unsigned)grid), dim3(NW * 64), lds, c->stream, p, n_o, n_tiles, c->redo);
        } else {
            hipLaunchKernelGGL((k_sweep_pipe<R, DIR, CH, false, 1, NW>), dim3((unsigned)grid), dim3(NW * 64), lds, c->stream, p, n_o, n_tiles, (int *)nullptr);
        }
    } else {
        hipLaunchKernelGGL((k_sweep_pipe<R, DIR, CH, false, 2, NW>), dim3((unsigned)grid), dim3(NW * 64), lds, c->stream, p, n_o, n_tiles, (int *)nullptr);
    }
    return true;
}

template <> bool xslab_pipe_supported<float>(const SweepParams<float> &p)
{
    return p.dimz % 64 == 0 && p.dimx <= PIPE_NW * 32 && (unsigned long long)p.fstride * 4ull * sizeof(float) < (1ull << 32);
}
template <> bool xslab_pipe_supported<double>(const SweepParams<double> &p)
{
    return p.dimz % 64 == 0 && p.dimx <= PIPE_NW * 16 && (unsigned long long)p.fstride * 4ull * sizeof(double) < (1ull << 32);
}
// a slab piece of n planes keeps ceil(n / CH) waves busy: workgroups of 1, 2, 4 or 8 waves (small workgroups leave
// room for several bundles per CU, which is what a thin slab needs)
template <typename R, int CH>
static bool launch_xslab_nw(fs3d_ctx *c, SweepParams<R> p, int half, int b0, int b1)
{
    const int nbt = p.dimy * (p.dimz / 64);
    const int nw = p.dimx <= CH ? 1 : (p.dimx <= 2 * CH ? 2 : (p.dimx <= 4 * CH ? 4 : 8));
    if (!pipe_scratch<R>(c, (size_t)nbt * 6 * nw * CH * 64)) return false;
    p.scr_ = (R *)c->scr; p.scr_bundles = nbt; p.seg_index = 0;
    switch (nw) {
    case 1: return launch_half<R, 0, CH, 1>(c, p, half, b0, b1);
    case 2: return launch_half<R, 0, CH, 2>(c, p, half, b0, b1);
    case 4: return launch_half<R, 0, CH, 4>(c, p, half, b0, b1);
    default: return launch_half<R, 0, CH, 8>(c, p, half, b0, b1);
    }
}
template <> bool launch_xslab_pipe<float>(fs3d_ctx *c, SweepParams<float> p, int half, int b0, int b1)
{
    if (p.dimx <= PIPE_NW * 16) return launch_xslab_nw<float, 16>(c, p, half, b0, b1);
    const int nbt = p.dimy * (p.dimz / 64);
    if (!pipe_scratch<float>(c, (size_t)nbt * 6 * PIPE_NW * 32 * 64)) return false;
    p.scr_ = (float *)c->scr; p.scr_bundles = nbt; p.seg_index = 0;
    return launch_half<float, 0, 32>(c, p, half, b0, b1);
}
template <> bool launch_xslab_pipe<double>(fs3d_ctx *c, SweepParams<double> p, int half, int b0, int b1)
{
    return launch_xslab_nw<double, 16>(c, p, half, b0, b1);
}

// A line of n cells, n above what one launch holds on chip (8 waves x CH cells): forward halves of the segments in
// line order, backward halves in reverse, the carries of a segment's last / first cell in two per-line arrays that
// each bundle reads before it overwrites them.  Cell for cell the arithmetic of the unsegmented sweep.
template <typename R, int DIR, int CH>
static bool run_segments(fs3d_ctx *c, SweepParams<R> p)
{
    const int n = DIR == 0 ? p.dimx : (DIR == 1 ? p.dimy : p.dimz);
    const int la_len = DIR == 2 ? p.dimy : p.dimz, n_o = DIR == 0 ? p.dimy : p.dimx;
    const int seg = PIPE_NW * CH, nseg = (n + seg - 1) / seg, nb = n_o * ((la_len + 63) / 64);
    const long long lines = (long long)n_o * la_len;
    if (c->seg_carry_lines < lines) {
        for (int i = 0; i < 2; i++) if (c->seg_carry[i]) { hipStreamSynchronize(c->stream); hipFree(c->seg_carry[i]); c->seg_carry[i] = nullptr; }
        // two forward carry arrays, used alternately: a bundle that asks for the full-division instance reads its carry_in
        // a second time, after the first instance has stored the segment's own carries
        if (hipMalloc(&c->seg_carry[0], 12 * (size_t)lines * sizeof(R)) != hipSuccess) { c->err = "pipe segments: hipMalloc of the carries failed"; return false; }
        if (hipMalloc(&c->seg_carry[1], 4 * (size_t)lines * sizeof(R)) != hipSuccess) return false;
        c->seg_carry_lines = lines;
    }
    p.carry_pitch = lines;
    if (!pipe_scratch<R>(c, (size_t)nseg * nb * 6 * PIPE_NW * CH * 64)) { c->err = "pipe segments: hipMalloc of the scratch failed"; return false; }
    p.scr_ = (R *)c->scr; p.scr_bundles = nb;
    for (int s = 0; s < nseg; s++) {
        p.seg_index = s;
        p.seg_begin = s * seg; p.seg_len = std::min(seg, n - s * seg);
        R *const fc[2] = {(R *)c->seg_carry[0], (R *)c->seg_carry[0] + 6 * lines};
        p.carry_in = s > 0 ? (const R *)fc[(s - 1) & 1] : nullptr; p.carry_out = fc[s & 1];
        if (!launch_half<R, DIR, CH>(c, p, 1, 0, nb)) return false;
    }
    for (int s = nseg - 1; s >= 0; s--) {
        p.seg_index = s;
        p.seg_begin = s * seg; p.seg_len = std::min(seg, n - s * seg);
        p.xcarry_in = s < nseg - 1 ? (const R *)c->seg_carry[1] : nullptr; p.xcarry_out = (R *)c->seg_carry[1];
        if (!launch_half<R, DIR, CH>(c, p, 2, 0, nb)) return false;
    }
    return true;
}

template <> bool launch_sweep_pipe_segmented<float>(fs3d_ctx *c, int dir, SweepParams<float> p)
{
    if ((unsigned long long)p.fstride * 4ull * sizeof(float) >= (1ull << 32)) return false;
    if (dir == 0 && (p.ghost_lo || p.ghost_hi)) return false;          // a slab's X sweep has its own path
    return dir == 0 ? run_segments<float, 0, 32>(c, p) : (dir == 1 ? run_segments<float, 1, 32>(c, p) : run_segments<float, 2, 32>(c, p));
}
template <> bool launch_sweep_pipe_segmented<double>(fs3d_ctx *c, int dir, SweepParams<double> p)
{
    if ((unsigned long long)p.fstride * 4ull * sizeof(double) >= (1ull << 32)) return false;
    if (dir == 0 && (p.ghost_lo || p.ghost_hi)) return false;
    return dir == 0 ? run_segments<double, 0, 16>(c, p) : (dir == 1 ? run_segments<double, 1, 16>(c, p) : run_segments<double, 2, 16>(c, p));
}
