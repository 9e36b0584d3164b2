// Partition (reduced-interface) sweep kernels (FS3D_SWEEP_PART) for CDNA4 -- the production path for fp32.
//
// What the reference does: one sequential Thomas recurrence per grid line (Common/Algorithms.h:21-38), rows from
// BuildMatrix / ApplyBC0/1 (FluidSolver3D/AdiSolver3D.cpp:732-852), scatter + merge (UpdateSegment :707-730,
// MergeFieldTo TimeLayer3D.h:415-436).  The exact kernel (kernels_pipe.hip) keeps that recurrence and therefore
// a 256-deep serial chain between the loads and the stores of a bundle; it is the bit-exact checker.  Here the
// line is cut into chunks that are eliminated AT THE SAME TIME, coupled by one small interface system per line
// (algebra and notation: cmc_fluid_solver_amd/partition.py, tests/test_partition_algebra.py).  Same equations,
// algebraically exact, different rounding: results agree with the CPU oracle to the tolerance stated in DESIGN.md
// section 5 (fp32: ~1e-7 rel-L2 per sweep), not bit for bit.  Compiled with FMA contraction; reciprocals are
// v_rcp_f32 + one Newton step; x/(2h) is a multiplication by the rounded reciprocal.
//
// X / Y sweeps  (k_sweep_part): lanes run along k (unit stride).  Workgroup = 32 neighbouring lines x the whole
//   line; thread (kk, ch) owns the M cells [ch*M, ch*M+M) of line kk, so a wave-wide access is two 128-byte rows.
//     P   rows of the thread's cells: q = Vs/(2h), dU, dV, dW in registers, dT in LDS (cell-by-cell software
//         pipeline: the next cells' 12 loads are in flight while a cell is computed)
//     E   down- and up-sweep over the chunk (independent chains, interleaved) -> interface coefficients to LDS
//     R   the NCH x NCH interface systems: one thread per (line, right-hand side), 128 threads
//     S   forward elimination with the left interface value known, back-substitution from the own one
//     O   scatter x to `next`, merged temp to `temp_out`
//   6 words per cell stay on chip between P and O: 4 in registers, dT/d'_T and c'_T in registers after P (LDS
//   during P), nothing goes through HBM.  Two workgroups per CU (<= 128 VGPRs, 68 KiB LDS): one streams while the
//   other solves.
// Z sweep  (k_sweep_part_z): lanes run along the line itself: lane l owns cells [4l, 4l+4) of ONE line (a 16-byte
//   piece; a wave-wide access is the whole contiguous line), neighbouring lines are further registers of the same
//   lane.  Chunks of a line = lanes of a wave: the interface system is solved by parallel cyclic reduction across
//   the lanes (shuffles), no LDS, no barrier, nothing resident but the lines in flight.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "fs3d_common.h"

template <typename F, int... I>
__device__ __forceinline__ void pstatic_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void pstatic_for(F &&f) { pstatic_for_impl(f, std::make_integer_sequence<int, N>{}); }

typedef __amdgpu_buffer_rsrc_t prsrc_t;
typedef unsigned pu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
#define PART_OOB 0xFFFFFFFFu      // voffset >= num_records: the hardware drops the store / returns 0 for the load

// Cache policy of the streams (the `aux` operand of the raw buffer intrinsics on gfx94x/gfx950: 1 = sc0, 2 = nt, 16 = sc1).
// `cur` is read exactly once per sweep, `next` / `temp_out` are written once and read again only by the NEXT launch, a gigabyte
// later: nontemporal.  X/Y kernels: the main `temp` loads stay cached -- the o+-1 rows of the neighbouring workgroups and the
// second read of the store phase are served from them (nt there: X +6 %, Y +3 % slower).  Z kernel: nt on every stream.
// Measured (profiles/r3_ab_nt.txt, one box, interleaved, per launch with all stores): X 0.300 -> 0.279, Y 0.272 -> 0.248,
// Z 0.242 -> 0.219 ms; tools/ubench/stream2.hip: nt loads read at 7.0 instead of 6.0-6.2 TB/s.
#ifndef FS3D_PART_AUX_CUR
#define FS3D_PART_AUX_CUR 2
#endif
#ifndef FS3D_PART_AUX_TMP
#define FS3D_PART_AUX_TMP 0
#endif
#ifndef FS3D_PART_AUX_ST
#define FS3D_PART_AUX_ST 2
#endif
#ifndef FS3D_PARTZ_AUX_TMP
#define FS3D_PARTZ_AUX_TMP 2
#endif
#ifndef FS3D_PARTZ_AUX_W
#define FS3D_PARTZ_AUX_W 0        // W of the line itself: the rows j+-1 and planes i+-1 read it again as their neighbour
#endif
template <typename R> struct PBuf;
template <> struct PBuf<float> {
    template <int AUX = 0> static __device__ __forceinline__ float ld(prsrc_t r, unsigned vo, unsigned so) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, AUX)); }
    template <int AUX = 0> static __device__ __forceinline__ void st(prsrc_t r, unsigned vo, unsigned so, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, vo, so, AUX); }
};
template <> struct PBuf<double> {
    template <int AUX = 0> static __device__ __forceinline__ double ld(prsrc_t r, unsigned vo, unsigned so) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, AUX)); }
    template <int AUX = 0> static __device__ __forceinline__ void st(prsrc_t r, unsigned vo, unsigned so, double v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(pu32x2, v), r, vo, so, AUX); }
};

// reciprocal: v_rcp_f32 (1 ulp) + one Newton step
__device__ __forceinline__ float prcp(float y)
{
    const float r = __builtin_amdgcn_rcpf(y);
    return __builtin_fmaf(__builtin_fmaf(-y, r, 1.0f), r, r);
}
__device__ __forceinline__ double prcp(double y) { return 1.0 / y; }
__device__ __forceinline__ float pfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double pfma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// x / y for a divisor y that is constant over the kernel (2h, dt), r = 1/y rounded: one correction step on x*r.
// The plain product x*r would carry the rounding error of r into EVERY cell with the same sign -- a systematic
// perturbation of the advection coefficients and right-hand sides (like a slightly different h or dt) that does not
// average out over the steps; with the correction the quotient is the correctly rounded one in all but rare cases.
__device__ __forceinline__ float pdivc(float x, float y, float r) { const float q = x * r; return __builtin_fmaf(__builtin_fmaf(-y, q, x), r, q); }
__device__ __forceinline__ double pdivc(double x, double y, double) { return x / y; }

// Opaque copies: the compiler must not merge the (cheap) address / coefficient computations of different phases
// into one computation whose results stay live -- in scratch memory -- from the first phase to the last.
template <typename T> __device__ __forceinline__ T opq_v(T x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ unsigned opq_s(unsigned x) { asm volatile("" : "+s"(x)); return x; }

template <typename R> struct PMat { R a, b, c; };

// (a, b, c) of the velocity and the temperature matrix for one cell: INTERIOR rows from q (BuildMatrix,
// AdiSolver3D.cpp:760-762), the other kinds from the 4-bit row code (ApplyBC0/1, :804-852; SKIP = identity row).
template <typename R, bool GEN>
__device__ __forceinline__ void part_coefs(R q, int code4, R vis_v, R b_v, R vis_t, R b_t, PMat<R> &mv, PMat<R> &mt)
{
    mv.a = -q - vis_v; mv.b = b_v; mv.c = q - vis_v;
    mt.a = -q - vis_t; mt.b = b_t; mt.c = q - vis_t;
    if (GEN) {
        const int kind = code4 & 3;
        const bool is_int = kind == ROW_INTERIOR;
        const bool fv = (code4 & ROW_VELFREE) != 0, ft = (code4 & ROW_TEMPFREE) != 0;
        mv.a = is_int ? mv.a : ((kind == ROW_END && fv) ? R(-1) : R(0));
        mv.c = is_int ? mv.c : ((kind == ROW_START && fv) ? R(-1) : R(0));
        mv.b = is_int ? mv.b : (fv ? R(2) : R(1));
        mt.a = is_int ? mt.a : ((kind == ROW_END && ft) ? R(-1) : R(0));
        mt.c = is_int ? mt.c : ((kind == ROW_START && ft) ? R(-1) : R(0));
        mt.b = is_int ? mt.b : (ft ? R(2) : R(1));
    }
}

// one elimination step of a sweep over the chunk: (lead, diag, trail) = (a, b, c) going down, (c, b, a) going up.
//   den = diag - lead*cp;  rhs' = (rhs - lead*rhs'_prev)/den;  spike' = -lead*spike_prev/den;  cp' = trail/den
// Measured on the shipped 64^3 example against the fp64 oracle: with the correction the partition kernels deviate
// from the fp64 solution as much as the sequential fp32 recurrence does (8.5e-7 vs 7.9e-7 after 10 steps, 1.2e-6 vs
// 1.0e-6 after 30); without it twice as much.  No measurable cost: the phases that divide are not the ones that bind.
#ifndef FS3D_PART_QCORR
#define FS3D_PART_QCORR 1         // 1: every quotient of the elimination gets a correction step (correctly rounded x/den)
#endif
template <typename R>
__device__ __forceinline__ R pquot(R num, R den, R r)
{
    const R q = num * r;
    return FS3D_PART_QCORR ? pfma(pfma(-den, q, num), r, q) : q;
}
template <typename R, int NR>
__device__ __forceinline__ void part_step(R lead, R diag, R trail, R &cp, R &sp, R (&dp)[NR], const R (&d)[NR])
{
    const R den = pfma(-lead, cp, diag);
    const R r = prcp(den);
#pragma unroll
    for (int k = 0; k < NR; k++) dp[k] = pquot(pfma(-lead, dp[k], d[k]), den, r);
    sp = pquot(-lead * sp, den, r);
    cp = pquot(trail, den, r);
    // pin: the step is evaluated HERE (otherwise the chain is sunk below the per-cell row-kind branches that follow
    // and every cell's coefficients wait for it in scratch memory)
    asm volatile("" : "+v"(cp), "+v"(sp));
#pragma unroll
    for (int k = 0; k < NR; k++) asm volatile("" : "+v"(dp[k]));
}

// (r3) The velocity and the temperature matrix take the same elimination step on different coefficients, and the right-hand sides
// come in pairs: U/V against the velocity matrix, W/T against (velocity, temperature).  In fp32 the step runs on 2-vectors --
// v_pk_fma_f32 / v_pk_mul_f32 do two lanes' worth per issue slot: 18 packed + 2 v_rcp_f32 instead of 37 scalar instructions per step;
// the kernels spend ~40 % of their time issuing VALU instructions.  Component for component the same
// operations as part_step: the same bits.
// Measured (profiles/r3_ab_packed.txt, interleaved, bit-identical fields): Z kernel (226 of 256 VGPRs, no spills) 0.2111 -> 0.2022 ms;
// X/Y kernels 0.254 -> 0.275 / 0.245 -> 0.271 -- they sit at their 128-VGPR budget and the even-aligned register pairs cost 16-21
// spilled registers.  Packed in the Z kernel only.
#ifndef FS3D_PART_UCOL
#define FS3D_PART_UCOL 1              // shared code columns (SweepParams::ucol)
#endif
#ifndef FS3D_PART_PACKED_XY
#define FS3D_PART_PACKED_XY 0
#endif
#ifndef FS3D_PART_PACKED_Z
#define FS3D_PART_PACKED_Z 1
#endif
typedef float pf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pf2 pk_fma(pf2 a, pf2 b, pf2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ pf2 pk_quot(pf2 num, pf2 den, pf2 r)
{
    const pf2 q = num * r;
    return FS3D_PART_QCORR ? pk_fma(pk_fma(-den, q, num), r, q) : q;
}
// lead/diag/trail: (a, b, c) going down, (c, b, a) going up, of the velocity (_v) and temperature (_t) matrix
template <typename R, bool PK>
__device__ __forceinline__ void part_step_vt(R lead_v, R diag_v, R trail_v, R lead_t, R diag_t, R trail_t, R &cpv, R &cpt, R &spv, R &spt,
                                             R (&dp3)[3], R (&dp1)[1], const R (&d3)[3], const R (&d1)[1])
{
    if constexpr (std::is_same<R, float>::value && PK) {
        const pf2 lead = {lead_v, lead_t}, diag = {diag_v, diag_t}, trail = {trail_v, trail_t};
        pf2 cp = {cpv, cpt}, sp = {spv, spt};
        const pf2 den = pk_fma(-lead, cp, diag);
        pf2 r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
        r = pk_fma(pk_fma(-den, r, pf2{1.0f, 1.0f}), r, r);
        const pf2 lead_vv = lead.xx, den_vv = den.xx, r_vv = r.xx;
        pf2 uv = {dp3[0], dp3[1]}, wt = {dp3[2], dp1[0]};
        uv = pk_quot(pk_fma(-lead_vv, uv, pf2{d3[0], d3[1]}), den_vv, r_vv);
        wt = pk_quot(pk_fma(-lead, wt, pf2{d3[2], d1[0]}), den, r);
        sp = pk_quot(-lead * sp, den, r);
        cp = pk_quot(trail, den, r);
        asm volatile("" : "+v"(cp), "+v"(sp), "+v"(uv), "+v"(wt));       // pin: see part_step
        cpv = cp.x; cpt = cp.y; spv = sp.x; spt = sp.y;
        dp3[0] = uv.x; dp3[1] = uv.y; dp3[2] = wt.x; dp1[0] = wt.y;
    } else {
        part_step<R, 3>(lead_v, diag_v, trail_v, cpv, spv, dp3, d3);
        part_step<R, 1>(lead_t, diag_t, trail_t, cpt, spt, dp1, d1);
    }
}

// ------------------------------------------------------------------------------------------------------------
// X / Y sweeps
// ------------------------------------------------------------------------------------------------------------
#ifndef FS3D_PART_PF
#define FS3D_PART_PF 2            // cells whose loads are in flight ahead of the cell being computed (P phase)
#endif
#ifndef FS3D_PART_OPF
#define FS3D_PART_OPF 2           // cells whose temp values are in flight ahead of the cell being stored (O phase)
#endif
#define PART_EXW 18               // interface words per (line, chunk): 5 per matrix, 2 per right-hand side

template <typename R, int DIR, int M, int NCH, int WPS, int LT, int PF = FS3D_PART_PF, int XB = 0, int OPF = FS3D_PART_OPF, bool KT = false>
__global__ void __launch_bounds__(LT * NCH, WPS) k_sweep_part(SweepParams<R> p, int n_o, int n_tiles, int order)
{
    static_assert(DIR == 0 || DIR == 1, "lanes along k: X and Y sweeps");
    extern __shared__ __attribute__((aligned(16))) unsigned char part_smem[];
    static_assert(LT == 16 || LT == 32 || LT == 64, "lines per workgroup");
    R *const ldsD = (R *)part_smem;                      // [NCH*M][LT]  dT of every cell (P -> E)
    R *const ex = ldsD + NCH * M * LT;                  // [PART_EXW][NCH][LT]
    const int t = threadIdx.x, kk = t % LT, ch = t / LT;
    // measurement only (fs3d_profile_sweep): 8 s_memtime stamps per wave; wave 7 stamps the constant-rate, chip-wide
    // s_memrealtime (100 MHz) instead -- s_memtime counters are not aligned between CUs
    unsigned long long *const stamp = p.stamps ? p.stamps + ((size_t)blockIdx.x * 8 + (t >> 6) % 8) * 8 : nullptr;
#define PSTAMP(k) do { if (stamp && (t & 63) == 0 && (t >> 6) < 8) stamp[k] = (t >> 6) == 7 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
    // Start stagger: the first workgroup of every CU starts with the launch, so all CUs load, then solve, then store at the
    // same time and HBM idles while they solve (tools/part_phases.py timeline).  Part of the first generation starts late.
    if ((order & 0x30) && blockIdx.x < 256) {
        const int grp = (order & 0x20) ? (int)((blockIdx.x >> 3) % 3) : (int)((blockIdx.x >> 3) & 1);
        for (int w = grp * (order >> 8); w > 0; w--) __builtin_amdgcn_s_sleep(127);
    }
    if (order & 0x40) {                                   // several workgroups per CU: the k-th one of every CU starts k delays late
        constexpr int WPC = 1024 / (LT * NCH) < 2 ? 2 : 1024 / (LT * NCH);
        const int slot = (int)(blockIdx.x >> 8);
        if (slot > 0 && slot < WPC)
            for (int w = slot * (order >> 8); w > 0; w--) __builtin_amdgcn_s_sleep(127);
    }
    PSTAMP(0);

    // XCD-aware block order (speed only): blocks b, b+8, .. share an XCD under round-robin dispatch; give each XCD a
    // contiguous range of logical ids so that the o+-1 rows a workgroup reads are streamed by CUs of the same L2
    int lb = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = lb & 7, slot = lb >> 3;
        lb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + slot;
    }
    // order 0: consecutive ids = consecutive rows/planes `o` of one lane tile; 1: = the lane tiles of one row/plane
    // (concurrently running workgroups then stream whole rows: DRAM page locality)
#ifdef FS3D_EXPERIMENTS              // timing experiments only (WRONG NUMBERS): never compiled into libfs3d_hip.so (build.build_variant only)
    const bool x_nonb = order & 2, x_nore = order & 4;
#else
    constexpr bool x_nonb = false, x_nore = false;
#endif
    if (order & 8) {                                      // experiment: the later waves of the workgroup first during P
        const int wq = (t >> 6) * 4 / (LT * NCH / 64);
        if (wq == 1) __builtin_amdgcn_s_setprio(1); else if (wq == 2) __builtin_amdgcn_s_setprio(2); else if (wq == 3) __builtin_amdgcn_s_setprio(3);
    }      // timing experiments only (wrong numbers): FS3D_PART_ORDER bits 1, 2
    const int tile_id = (order & 1) ? lb % n_tiles : lb / n_o;
    const int o = ((order & 1) ? lb / n_tiles : lb - tile_id * n_o) + (DIR == 1 ? p.o_begin : 0);

    const int n = DIR == 0 ? p.dimx : p.dimy;
    const int la_len = p.dimz;
    const int k = tile_id * LT + kk;
    const bool lane_valid = k < la_len;
    const int kc = lane_valid ? k : la_len - 1;
    const long long ss = DIR == 0 ? p.plane : (long long)p.dimz;     // element stride along the sweep
    const long long os = DIR == 0 ? (long long)p.dimz : p.plane;     // element stride of the `o` axis
    const int s0 = ch * M;
    const unsigned ssb_p = (unsigned)(ss * (long long)sizeof(R)), osb = (unsigned)(os * (long long)sizeof(R));
    const unsigned fsb_p = (unsigned)(p.fstride * (long long)sizeof(R)), nsb = (unsigned)(p.nstride * (long long)sizeof(R));
    const unsigned ssb = ssb_p, fsb = fsb_p;
    // per-lane element offset of the chunk's first cell; the wave-uniform rest goes through the scalar offset
    const unsigned vel = (unsigned)((long long)s0 * ss + kc);
    const unsigned vo = vel * (unsigned)sizeof(R);
    const unsigned so0_p = (unsigned)((p.plane + (long long)o * os) * (long long)sizeof(R));  // layer fields: one halo plane first
    const unsigned so0 = so0_p;
    const unsigned son = (unsigned)(((long long)o * os) * (long long)sizeof(R));              // node values / codes: no halo plane
    const unsigned lbytes = 4u * fsb;
    const prsrc_t Lcur = __builtin_amdgcn_make_buffer_rsrc((void *)(p.cur_ - p.plane), 0, (int)lbytes, 0x00020000);
    const prsrc_t Ltmp = __builtin_amdgcn_make_buffer_rsrc((void *)(p.temp_ - p.plane), 0, (int)lbytes, 0x00020000);
    const prsrc_t Lnext = __builtin_amdgcn_make_buffer_rsrc((void *)(p.next_ - p.plane), 0, (int)lbytes, 0x00020000);
    const prsrc_t Ltout = __builtin_amdgcn_make_buffer_rsrc((void *)(p.temp_out_ - p.plane), 0, (int)lbytes, 0x00020000);
    const prsrc_t rNode = __builtin_amdgcn_make_buffer_rsrc((void *)p.node_, 0, (int)(4u * nsb), 0x00020000);
    const prsrc_t rCode = __builtin_amdgcn_make_buffer_rsrc((void *)p.code, 0, (int)(nsb / (sizeof(R) / 2)), 0x00020000);

    constexpr int M1 = DIR == 0 ? 1 : 0;                 // axis of the `o` neighbours
    constexpr int M2 = 2;                                // lane axis
    const R h2s = p.two_ds[DIR], h2o = p.two_ds[M1], h2l = p.two_ds[M2], dtv = p.dt;
    const R ir2s = R(1) / h2s, ir2o = R(1) / h2o, ir2l = R(1) / h2l, irdt = R(1) / dtv;
    const R vis_v = p.vis_v, vis_t = p.vis_t, b_v = p.b_v, b_t = p.b_t;

    // ---- row codes of the chunk: 4 bits per cell, NODE_IN mask, INTERIOR mask --------------------------------
    constexpr int NCP = (M + 7) / 8;
    unsigned cpack[NCP];
    unsigned inmask = 0, intmask = 0, segmask = 0;
    int cwv[M];
    {
        // (r3) a group of lines that all carry the same codes reads ONE shared column (M codes = M/8 16-byte loads, the same
        // addresses on all lanes of a chunk: cache hits) instead of M 2-byte loads per lane, 2 bytes per cell from HBM
        bool tile_uni = false;
        unsigned ucol_id = 0;
        if (p.uflag && XB == 0 && FS3D_PART_UCOL != 0) {
            const unsigned f = p.uflag[(long long)o * p.ung + tile_id * LT / 32];
            tile_uni = LT == 64 ? (f & 2u) != 0 : (f & 1u) != 0;
            ucol_id = f >> 2;
        }
        if (tile_uni) {
            const pu32x4 *const col = (const pu32x4 *)(p.ucol + (long long)ucol_id * UCOL_PITCH + s0);
#pragma unroll
            for (int q4 = 0; q4 < M / 8; q4++) {
                const pu32x4 w = col[q4];
                cwv[8 * q4 + 0] = (int)(w.x & 0xFFFFu); cwv[8 * q4 + 1] = (int)(w.x >> 16); cwv[8 * q4 + 2] = (int)(w.y & 0xFFFFu); cwv[8 * q4 + 3] = (int)(w.y >> 16);
                cwv[8 * q4 + 4] = (int)(w.z & 0xFFFFu); cwv[8 * q4 + 5] = (int)(w.z >> 16); cwv[8 * q4 + 6] = (int)(w.w & 0xFFFFu); cwv[8 * q4 + 7] = (int)(w.w >> 16);
            }
        } else {
        unsigned s_c = (son / (unsigned)sizeof(R)) * 2u;
        const unsigned ssc = (ssb / (unsigned)sizeof(R)) * 2u;
#pragma unroll
        for (int i = 0; i < M; i++) { cwv[i] = __builtin_amdgcn_raw_buffer_load_b16(rCode, vel * 2u, s_c, 0); s_c = opq_s(s_c + ssc); }
        }
    }
    // the first loads of the P phase do not depend on the codes: in flight before the codes are waited for (one memory
    // round trip less on the critical path of the workgroup)
    struct CellLd { R tp[4], c[4], om, op, le; };        // le: the lane-axis neighbour of the tile's two edge lanes
    // running scalar offsets, opaque from cell to cell: otherwise the offsets of all M cells are computed up front and
    // held in (spilled) SGPRs
    unsigned s_is = so0, s_nd = son;
    const unsigned vo_edge = kk == 0 ? vo - (unsigned)sizeof(R) : (kk == LT - 1 ? vo + (unsigned)sizeof(R) : PART_OOB);
    auto issue = [&](CellLd &L) __attribute__((always_inline)) {
        const unsigned sc = s_is;
        s_is = opq_s(s_is + ssb);
#pragma unroll
        for (int f = 0; f < 4; f++) L.tp[f] = PBuf<R>::template ld<FS3D_PART_AUX_TMP>(Ltmp, vo, sc + ssb + (unsigned)f * fsb);
#pragma unroll
        for (int f = 0; f < 4; f++) L.c[f] = PBuf<R>::template ld<FS3D_PART_AUX_CUR>(Lcur, vo, sc + (unsigned)f * fsb);
        const unsigned sv = sc + (unsigned)DIR * fsb;
        if (x_nonb) { L.om = L.op = L.le = L.c[0]; }
        else {
            L.om = PBuf<R>::ld(Ltmp, vo, sv - osb); L.op = PBuf<R>::ld(Ltmp, vo, sv + osb);
            // lane-axis neighbours: the lanes next door hold them (DPP shifts below); only the tile's first and last lane
            // fetch theirs from outside the tile -- every other lane of this load is out of range (no memory access)
            L.le = PBuf<R>::ld(Ltmp, vo_edge, sv);
        }
    };
    R Tm[4], Tc[4];
#pragma unroll
    for (int f = 0; f < 4; f++) { Tm[f] = PBuf<R>::ld(Ltmp, vo, so0 - ssb + (unsigned)f * fsb); Tc[f] = PBuf<R>::ld(Ltmp, vo, so0 + (unsigned)f * fsb); }
    CellLd L[PF + 1];
#pragma unroll
    for (int i = 0; i < PF && i < M; i++) issue(L[i]);
    __builtin_amdgcn_sched_barrier(0);
    {
#pragma unroll
        for (int i = 0; i < NCP; i++) cpack[i] = 0;
#pragma unroll
        for (int i = 0; i < M; i++) {
            int cw = cwv[i] & 0xFFFF;
            cw = (s0 + i < n) ? cw : 0;                  // cells past the end of the line: SKIP rows, never stored
            const int code4 = (cw >> (4 * DIR)) & 0xF;
            cpack[i >> 3] |= (unsigned)code4 << (4 * (i & 7));
            if (((cw >> CODE_TYPE_SHIFT) & 3) == FS3D_NODE_IN && s0 + i < n) inmask |= 1u << i;
            if ((code4 & 3) == ROW_INTERIOR) intmask |= 1u << i;
            if ((code4 & 3) != ROW_SKIP) segmask |= 1u << i;
        }
        // pin: the packed forms are computed here (otherwise their computation is sunk to the O phase and the M raw
        // code words wait for it in scratch memory)
        asm volatile("" : "+v"(inmask), "+v"(segmask), "+v"(intmask));
#pragma unroll
        for (int i = 0; i < NCP; i++) asm volatile("" : "+v"(cpack[i]));
    }
    // Dead lines (the wall lines of a box): nothing computed for them is stored, temp_out receives their old temp.
    // They must not keep the other lines of the wave off the select-free paths: they run along as INTERIOR rows --
    // finite or not, whatever they produce stays in its lane (and in its own interface system).
    const bool dead = p.dead[(long long)o * p.dimz + kc] != 0;
    // X sweep of an x-slab (reduced-interface form, fs3d_hip.hip: xsweep_reduced): the values just below / above the slab on
    // this line are given -- x[-1] in p.carry_in rows 2..5, x[n] in p.xcarry_in rows 0..3 ([value][line], line = j*dimz + k)
    constexpr bool xb = XB == 1 && DIR == 0;             // a template parameter: the single-GPU instance must not pay registers for it
    // XB == 2: first pass of that sweep -- rows and chunk elimination as always, then the slab's 18 interface words per line
    // (the layout of k_xiface, kernels_line.hip) to p.carry_out, and nothing else: no boundary values are known yet
    constexpr bool xa = XB == 2 && DIR == 0;
    // this lane's line in the carry arrays, recomputed where it is needed (a register kept across the kernel costs spills)
    auto cline_of = [&]() __attribute__((always_inline)) {
        const unsigned kc_ = opq_v(vo) / (unsigned)sizeof(R) - (unsigned)((long long)(opq_v(t) / LT * M) * ss);
        return (long long)o * p.dimz + kc_;
    };
    // cells that are INTERIOR rows on all 64 lanes of the wave: plain scalar branches pick the select-free code
    unsigned umask;
    {
        unsigned m = dead ? 0xFFFFFFFFu : intmask;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) m &= (unsigned)__shfl_xor((int)m, d, 64);
        umask = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
    }
    // opaque per use: the lane masks derived from a code (SGPR pairs) must not be shared between the phases
    auto code_of = [&](int i) __attribute__((always_inline)) { return (int)((opq_v(cpack[i >> 3]) >> (4 * (i & 7))) & 0xFu); };

    PSTAMP(1);
    // ---- P: rows -------------------------------------------------------------------------------------------
    R q[M], dU[M], dV[M], dW[M];
    R tkeep[KT ? 4 : 1][KT ? M : 1];                     // KT: the temp values of the own cells stay in registers for the merge (no second read)
    {
        pstatic_for<M>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if (i + PF < M) issue(L[(i + PF) % (PF + 1)]);
            __builtin_amdgcn_sched_barrier(0);
            const CellLd &c = L[i % (PF + 1)];
            R qq = pdivc(Tc[DIR], h2s, ir2s);
            R g[4];
#pragma unroll
            for (int f = 0; f < 4; f++) g[f] = pdivc(c.tp[f] - Tm[f], h2s, ir2s);          // d/ds of U, V, W, T (TimeLayer3D.h:338-340)
            R lm, lp;                                             // Vs of the lanes next door (wave_shr:1 / wave_shl:1)
            if (sizeof(R) == 4) {
                const int cv = __builtin_bit_cast(int, (float)Tc[DIR]);
                lm = (R)__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(cv, cv, 0x138, 0xF, 0xF, false));
                lp = (R)__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(cv, cv, 0x130, 0xF, 0xF, false));
            } else { lm = __shfl_up(Tc[DIR], 1, 64); lp = __shfl_down(Tc[DIR], 1, 64); }
            lm = kk == 0 ? c.le : lm; lp = kk == LT - 1 ? c.le : lp;
            const R x1 = pdivc(c.op - c.om, h2o, ir2o), x2 = pdivc(lp - lm, h2l, ir2l);   // d(Vs)/d(o axis), d(Vs)/d(lane axis)
            const R t0 = (DIR == 0 ? R(2) : R(1)) * g[0] * g[0], t1 = (DIR == 1 ? R(2) : R(1)) * g[1] * g[1], t2 = g[2] * g[2];
            const R diss = (((t0 + t1) + t2) + g[M1] * x1) + g[M2] * x2;          // DissFuncX/Y (TimeLayer3D.h:554-577)
            R dd[4];
#pragma unroll
            for (int f = 0; f < 3; f++) dd[f] = pdivc(c.c[f] * R(3), dtv, irdt);    // cur * 3 / dt (AdiSolver3D.cpp:764-799)
            dd[DIR] = pfma(-p.v_T, g[3], dd[DIR]);
            dd[3] = pfma(p.t_phi, diss, pdivc(c.c[3] * R(3), dtv, irdt));
            if (!((opq_s(umask) >> i) & 1u)) {
                // some line has another row kind here: START/END  d = node value (NOSLIP) or 0 (FREE); SKIP  d = 0
                const int code4 = code_of(i), kind = code4 & 3;
                const bool is_int = kind == ROW_INTERIOR;
                const bool ns_v = kind != ROW_SKIP && !(code4 & ROW_VELFREE), ns_t = kind != ROW_SKIP && !(code4 & ROW_TEMPFREE);
                R nv[4];
#pragma unroll
                for (int f = 0; f < 4; f++) nv[f] = PBuf<R>::ld(rNode, vo, s_nd + (unsigned)f * nsb);
                qq = is_int ? qq : R(0);
#pragma unroll
                for (int f = 0; f < 3; f++) dd[f] = is_int ? dd[f] : (ns_v ? nv[f] : R(0));
                dd[3] = is_int ? dd[3] : (ns_t ? nv[3] : R(0));
                if (xb && __any(s0 + i == n)) {
                    // the first cell past the slab: an identity row that holds the given x[n] (the last plane's rows couple to it)
                    const long long cline = cline_of();
#pragma unroll
                    for (int f = 0; f < 4; f++) { const R xr = p.xcarry_in[(long long)f * p.carry_pitch + cline]; dd[f] = s0 + i == n ? xr : dd[f]; }
                }
            }
            q[i] = qq; dU[i] = dd[0]; dV[i] = dd[1]; dW[i] = dd[2];
            if constexpr (KT) {
#pragma unroll
                for (int f = 0; f < 4; f++) tkeep[f][i] = Tc[f];
            }
            ldsD[(s0 + i) * LT + kk] = dd[3];
#pragma unroll
            for (int f = 0; f < 4; f++) { Tm[f] = Tc[f]; Tc[f] = c.tp[f]; }
            s_nd = opq_s(s_nd + ssb);
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    if (order & 8) __builtin_amdgcn_s_setprio(0);
    PSTAMP(2);
    // ---- E: chunk elimination -> interface coefficients ------------------------------------------------------
    // every use recomputes the coefficients from an opaque copy of q (4 operations) instead of keeping 6 x M values alive
    auto coefs = [&](int i, PMat<R> &mv, PMat<R> &mt) __attribute__((always_inline)) {
        const R qi = opq_v(q[i]);
        if ((opq_s(umask) >> i) & 1u) part_coefs<R, false>(qi, 0, vis_v, b_v, vis_t, b_t, mv, mt);
        else part_coefs<R, true>(qi, code_of(i), vis_v, b_v, vis_t, b_t, mv, mt);
    };
    const R *const myD = ldsD + (s0 * LT + kk);         // dT of local cell i at myD[i * LT] (this thread wrote it: no barrier)
    {
        R cpv = R(0), lpv = R(-1), cpt = R(0), lpt = R(-1), dp3[3] = {R(0), R(0), R(0)}, dp1[1] = {R(0)};   // down: x[M-2] = dp - lp X_{p-1} - cp X_p
        R apv = R(0), upv = R(-1), apt = R(0), upt = R(-1), ep3[3] = {R(0), R(0), R(0)}, ep1[1] = {R(0)};   // up:   x[0]   = ep - ap X_{p-1} - up X_p
        // the T right-hand sides stay in the LDS during this phase (register budget); read one step ahead of their use
        R tdn = myD[0], tup = myD[(M - 2) * LT];
        pstatic_for<M - 1>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value, j = M - 2 - i;
            const R tdi = tdn, tdj = tup;
            if (i + 1 < M - 1) { tdn = myD[(i + 1) * LT]; tup = myD[(j - 1 < 0 ? 0 : j - 1) * LT]; }
            PMat<R> mv, mt, nv, nt;
            coefs(i, mv, mt);
            coefs(j, nv, nt);
            { const R d3[3] = {dU[i], dV[i], dW[i]}, d1[1] = {tdi}; part_step_vt<R, FS3D_PART_PACKED_XY != 0>(mv.a, mv.b, mv.c, mt.a, mt.b, mt.c, cpv, cpt, lpv, lpt, dp3, dp1, d3, d1); }
            { const R d3[3] = {dU[j], dV[j], dW[j]}, d1[1] = {tdj}; part_step_vt<R, FS3D_PART_PACKED_XY != 0>(nv.c, nv.b, nv.a, nt.c, nt.b, nt.a, apv, apt, upv, upt, ep3, ep1, d3, d1); }
        });
        // the interface cell's own row with x[M-2] eliminated:  A X_{p-1} + Bp X_p + cl x_first(p+1) = Dp
        PMat<R> lv, lt;
        coefs(M - 1, lv, lt);
        R *const e = ex + (ch * LT + kk);
        constexpr int ES = NCH * LT;
        e[0 * ES] = -lv.a * lpv; e[1 * ES] = pfma(-lv.a, cpv, lv.b); e[2 * ES] = lv.c; e[3 * ES] = apv; e[4 * ES] = upv;
        e[5 * ES] = -lt.a * lpt; e[6 * ES] = pfma(-lt.a, cpt, lt.b); e[7 * ES] = lt.c; e[8 * ES] = apt; e[9 * ES] = upt;
        e[10 * ES] = pfma(-lv.a, dp3[0], dU[M - 1]); e[11 * ES] = pfma(-lv.a, dp3[1], dV[M - 1]);
        e[12 * ES] = pfma(-lv.a, dp3[2], dW[M - 1]); e[13 * ES] = pfma(-lt.a, dp1[0], myD[(M - 1) * LT]);
        e[14 * ES] = ep3[0]; e[15 * ES] = ep3[1]; e[16 * ES] = ep3[2]; e[17 * ES] = ep1[0];
    }
    PSTAMP(3);
    __syncthreads();
    PSTAMP(4);

    if constexpr (xa) {
        // ---- R (first pass of a slab sweep): the chunks' interface rows -> the SLAB's interface row and first cell.
        // Chunks 0..cA hold the slab (n is a multiple of M: the slab's last plane is the interface cell of chunk cA).
        //   forward over the chunks:  X_c = dq - lq X_below - cq X_{c+1};  the row of chunk cA then reads
        //                             (-lo lq) X_below + (Bp - lo cq) X_cA + cl x_first(slab above) = Dp - lo dq
        //   backward over the chunks: X_c = eq - aq X_{c-1} - uq X_cA;  x_first = Gf0 - Vf0 X_below - Wf0 X_0
        if (t < 4 * LT) {
            constexpr int ES = NCH * LT;
            const int sys = t / LT, cA = n / M - 1;
            const R *const em = ex + (sys == 3 ? 5 * ES : 0) + kk;
            const R *const er = ex + (10 + sys) * ES + kk;
            auto row = [&](int c, R &lo, R &di, R &up, R &rhs) __attribute__((always_inline)) {
                lo = em[0 * ES + c * LT];
                const R bp = em[1 * ES + c * LT], cl = em[2 * ES + c * LT];
                const R vf = em[3 * ES + (c + 1) * LT], wf = em[4 * ES + (c + 1) * LT], gf = er[4 * ES + (c + 1) * LT];
                di = pfma(-cl, vf, bp); up = -cl * wf; rhs = pfma(-cl, gf, er[c * LT]);
            };
            R cq = R(0), dq = R(0), lq = R(-1);
#pragma nounroll
            for (int c = 0; c < cA; c++) {
                R lo, di, up, rhs;
                row(c, lo, di, up, rhs);
                const R den = pfma(-lo, cq, di), r = prcp(den);
                cq = pquot(up, den, r); dq = pquot(pfma(-lo, dq, rhs), den, r); lq = pquot(-lo * lq, den, r);
            }
            R eq = R(0), aq = R(0), uq = R(-1);
#pragma nounroll
            for (int c = cA - 1; c >= 0; c--) {
                R lo, di, up, rhs;
                row(c, lo, di, up, rhs);
                const R den = pfma(-up, aq, di), r = prcp(den);
                eq = pquot(pfma(-up, eq, rhs), den, r); uq = pquot(-up * uq, den, r); aq = pquot(lo, den, r);
            }
            const R lo = em[0 * ES + cA * LT], bp = em[1 * ES + cA * LT], cl = em[2 * ES + cA * LT];
            const R vf0 = em[3 * ES], wf0 = em[4 * ES], gf0 = er[4 * ES];
            // dead lines (no cell of theirs is ever stored) hand over identity words: whatever ran through their lanes stays there
            R *const o = p.carry_out + cline_of();
            const long long nl = p.carry_pitch;
            if (sys == 0 || sys == 3) {
                const int m = sys == 3 ? 5 : 0;
                o[(m + 0) * nl] = dead ? R(0) : -lo * lq; o[(m + 1) * nl] = dead ? R(1) : pfma(-lo, cq, bp); o[(m + 2) * nl] = dead ? R(0) : cl;
                o[(m + 3) * nl] = dead ? R(0) : pfma(-wf0, aq, vf0); o[(m + 4) * nl] = dead ? R(0) : -wf0 * uq;
            }
            o[(10 + sys) * nl] = dead ? R(0) : pfma(-lo, dq, er[cA * LT]);
            o[(14 + sys) * nl] = dead ? R(0) : pfma(-wf0, eq, gf0);
        }
        return;
    }

    // ---- R: interface systems, one thread per (line, right-hand side) ----------------------------------------
    if (t < 4 * LT) {
        constexpr int ES = NCH * LT;
        const int sys = t / LT;
        const R *const em = ex + (sys == 3 ? 5 * ES : 0) + kk;      // this system's matrix words
        R *const er = ex + (10 + sys) * ES + kk;                    // Dp (-> X), Gf at + 4*ES
        // c', d' of the interface system: registers up to 16 chunks, LDS beyond (c' in an array of its own, d' over Dp)
        constexpr bool RL = NCH > 16;
        R cpa[RL ? 1 : NCH], dpa[RL ? 1 : NCH];
        R *const cx = ex + PART_EXW * ES + sys * ES + kk;
        // slab mode: the row of chunk 0 couples to the given x[-1] (lo * x[-1] moves to the right-hand side: start the
        // recurrence from d' = x[-1], c' = 0); a line that fills all chunks couples its last row to the given x[n]
        R cp = R(0), dp = R(0), xr_last = R(0);
        if (xb) {
            const long long cline = cline_of();
            dp = p.carry_in[(long long)(2 + sys) * p.carry_pitch + cline];
            if (n == NCH * M) xr_last = p.xcarry_in[(long long)sys * p.carry_pitch + cline];
        }
        auto fwd = [&](int c) __attribute__((always_inline)) {
            const R lo = em[0 * ES + c * LT], bp = em[1 * ES + c * LT], cl = em[2 * ES + c * LT];
            R vf = R(0), wf = R(0), gf = R(0);
            if (c + 1 < NCH) { vf = em[3 * ES + (c + 1) * LT]; wf = em[4 * ES + (c + 1) * LT]; gf = er[4 * ES + (c + 1) * LT]; }
            else gf = xr_last;
            const R di = pfma(-cl, vf, bp), up = -cl * wf, rhs = pfma(-cl, gf, er[c * LT]);
            const R den = pfma(-lo, cp, di), r = prcp(den);
            cp = pquot(up, den, r); dp = pquot(pfma(-lo, dp, rhs), den, r);
        };
        if (RL) {
            // many chunks: a rolled loop, c' and d' through the LDS (unrolled, the reads of all chunks are hoisted and spill)
#pragma nounroll
            for (int c = 0; c < NCH; c++) { fwd(c); cx[c * LT] = cp; er[c * LT] = dp; }
            R x = dp;
#pragma nounroll
            for (int c = NCH - 2; c >= 0; c--) { x = pfma(-cx[c * LT], x, er[c * LT]); er[c * LT] = x; }
        } else {
#pragma unroll
            for (int c = 0; c < NCH; c++) { fwd(c); cpa[c] = cp; dpa[c] = dp; }
            R x = dp;
            er[(NCH - 1) * LT] = x;
#pragma unroll
            for (int c = NCH - 2; c >= 0; c--) { x = pfma(-cpa[c], x, dpa[c]); er[c * LT] = x; }
        }
    }
    __syncthreads();
    PSTAMP(5);

    // ---- S: the chunk with both interface values known ---------------------------------------------------------
    R cT[M], dT[M];                                     // c'_T and d'_T, then x_T in dT
    {
        constexpr int ES = NCH * LT;
        R xo[4], dp3[3], dp1[1];
#pragma unroll
        for (int s = 0; s < 4; s++) xo[s] = ex[(10 + s) * ES + ch * LT + kk];
#pragma unroll
        for (int s = 0; s < 3; s++) dp3[s] = ch > 0 ? ex[(10 + s) * ES + (ch - 1) * LT + kk] : R(0);
        dp1[0] = ch > 0 ? ex[13 * ES + (ch - 1) * LT + kk] : R(0);
        if (xb && __any(ch == 0)) {                      // chunk 0 of a slab: the given x[-1]
            const long long cline = cline_of();
#pragma unroll
            for (int s = 0; s < 3; s++) { const R xl = p.carry_in[(long long)(2 + s) * p.carry_pitch + cline]; dp3[s] = ch == 0 ? xl : dp3[s]; }
            const R xl = p.carry_in[5ll * p.carry_pitch + cline];
            dp1[0] = ch == 0 ? xl : dp1[0];
        }
        R cpv = R(0), cpt = R(0), sp = R(0);
        R tdn = myD[0];
        pstatic_for<M - 1>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            const R tdi = tdn;
            if (i + 1 < M - 1) tdn = myD[(i + 1) * LT];
            PMat<R> mv, mt;
            coefs(i, mv, mt);
            { const R d3[3] = {dU[i], dV[i], dW[i]}, d1[1] = {tdi}; R spt = sp; part_step_vt<R, FS3D_PART_PACKED_XY != 0>(mv.a, mv.b, mv.c, mt.a, mt.b, mt.c, cpv, cpt, sp, spt, dp3, dp1, d3, d1); }
            q[i] = cpv; cT[i] = cpt; dU[i] = dp3[0]; dV[i] = dp3[1]; dW[i] = dp3[2]; dT[i] = dp1[0];
        });
        dU[M - 1] = xo[0]; dV[M - 1] = xo[1]; dW[M - 1] = xo[2]; dT[M - 1] = xo[3];
#pragma unroll
        for (int i = M - 2; i >= 0; i--) {
            xo[0] = pfma(-q[i], xo[0], dU[i]); xo[1] = pfma(-q[i], xo[1], dV[i]);
            xo[2] = pfma(-q[i], xo[2], dW[i]); xo[3] = pfma(-cT[i], xo[3], dT[i]);
            dU[i] = xo[0]; dV[i] = xo[1]; dW[i] = xo[2]; dT[i] = xo[3];
        }
    }

    PSTAMP(6);
    // ---- O: scatter + merge -------------------------------------------------------------------------------------
    {
        const unsigned vo_st = lane_valid ? vo : PART_OOB;
        const unsigned so0 = opq_s(so0_p), ssb = opq_s(ssb_p), fsb = opq_s(fsb_p);   // not the P phase's address arithmetic kept alive
        R tv[OPF + 1][4];
        unsigned s_is = so0, s_o = so0;                 // running scalar offsets (issue side / store side), opaque per cell
        const unsigned segm = opq_v(segmask), inm = opq_v(inmask);
        const int nloc = opq_v(n - s0);                 // cells of this chunk inside the line
        auto issue = [&](R (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int f = 0; f < 4; f++) v[f] = x_nore ? R(f) : PBuf<R>::ld(Ltmp, vo, s_is + (unsigned)f * fsb);
            s_is = opq_s(s_is + ssb);
        };
        if (p.merge && !KT) {
#pragma unroll
            for (int i = 0; i < OPF && i < M; i++) issue(tv[i]);
        }
        pstatic_for<M>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            const unsigned sc = s_o;
            s_o = opq_s(s_o + ssb);
            if (p.merge && !KT && i + OPF < M) issue(tv[(i + OPF) % (OPF + 1)]);
            __builtin_amdgcn_sched_barrier(0);
            R xv[4] = {dU[i], dV[i], dW[i], dT[i]};
            const bool uni = (opq_s(umask) >> i) & 1u;
            const bool seg = !dead && (uni || ((segm >> i) & 1u)), isin = !dead && (uni || ((inm >> i) & 1u));
            const bool in_line = i < nloc;
            if (p.store_next) {
                const unsigned v_ = seg ? vo_st : PART_OOB;            // UpdateSegment: every cell of a segment, nothing else
#pragma unroll
                for (int f = 0; f < 4; f++) PBuf<R>::template st<FS3D_PART_AUX_ST>(Lnext, v_, sc + (unsigned)f * fsb, xv[f]);
            }
            if (p.merge) {
                R tq[4];
#pragma unroll
                for (int f = 0; f < 4; f++) tq[f] = KT ? tkeep[f][KT ? i : 0] : tv[i % (OPF + 1)][f];
                if (!uni && __any(isin && !seg)) {
                    // NODE_IN cell outside every segment (run without a closing cell, Grid3D.cpp:87-117): the reference
                    // merges the stale `next` value
#pragma unroll
                    for (int f = 0; f < 4; f++) { const R sv = PBuf<R>::ld(Lnext, vo, sc + (unsigned)f * fsb); xv[f] = (isin && !seg) ? sv : xv[f]; }
                }
                const unsigned v_ = in_line ? vo_st : PART_OOB;
#pragma unroll
                for (int f = 0; f < 4; f++) {
                    R mv = (tq[f] + xv[f]) * R(0.5);                                   // MergeFieldTo (TimeLayer3D.h:415-436)
                    if (p.merge == 2) mv = (mv + xv[f]) * R(0.5);
                    PBuf<R>::template st<FS3D_PART_AUX_ST>(Ltout, v_, sc + (unsigned)f * fsb, isin ? mv : tq[f]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    PSTAMP(7);
#undef PSTAMP
}

// Kernel-experiment knobs (tile order, start delays, LDS padding, ...; FS3D_PART_ORDER bits 1/2 skip loads and give WRONG numbers):
// read from the environment only in -DFS3D_EXPERIMENTS builds (build.build_variant); libfs3d_hip.so compiles the defaults in.
#ifdef FS3D_EXPERIMENTS
static int part_exp_env(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
#else
static constexpr int part_exp_env(const char *, int dflt) { return dflt; }
#endif

template <typename R, int DIR, int M, int NCH, int WPS, int LT, int PF = FS3D_PART_PF, int XB = 0, int OPF = FS3D_PART_OPF, bool KT = false>
static bool part_launch_xy(fs3d_ctx *c, const SweepParams<R> &p)
{
    const int n_o = DIR == 0 ? p.dimy : (p.o_count ? p.o_count : p.dimx);
    const int n_tiles = (p.dimz + LT - 1) / LT;
    // FS3D_PART_LDSPAD (kernel experiments): more dynamic LDS than needed, to hold the workgroups per CU down
    static const size_t lds_pad = (size_t)part_exp_env("FS3D_PART_LDSPAD", 0);
    const size_t lds = ((size_t)NCH * M * LT + (size_t)(PART_EXW + (NCH > 16 ? 4 : 0)) * NCH * LT) * sizeof(R) + lds_pad;
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long dev_bit = 1ull << (c->device & 63);
    if (!(attr_set.load() & dev_bit)) {
        if (hipFuncSetAttribute((const void *)k_sweep_part<R, DIR, M, NCH, WPS, LT, PF, XB, OPF, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            c->err = std::string("partition kernel: hipFuncSetAttribute: ") + hipGetErrorString(hipGetLastError());
            return false;
        }
        attr_set.fetch_or(dev_bit);
    }
    // workgroup order: 32-line tiles run the lane tiles of one row/plane on consecutive workgroups (the 128-byte pieces of a
    // row are then fetched together: 512^3 X 2.93 -> 2.71 ms, Y 2.18 -> 2.10), 64-line tiles the rows/planes of one lane tile
    static const int order_env = part_exp_env("FS3D_PART_ORDER", -1);
    // 32-line tiles, two workgroups per CU: the second workgroup of every CU starts ~16 us late, so that the two do not load,
    // solve and store at the same time (tools/ab_tiles.py, interleaved on one box: 64 lines 6.12 ms per step, 32 lines 5.81,
    // 32 lines with the late start 5.63-5.66).  Few workgroups (thin slabs) and 512-cell lines: lane tiles fastest.
    // (the late start only where it was measured: 512-thread workgroups, two per CU, at least two generations of them)
    const bool late = LT <= 32 && NCH == 16 && M == 16 && (long long)n_o * n_tiles >= 1024;
    static const int late_slab = part_exp_env("FS3D_PART_LATE_SLAB", 0);   // experiment: slab kernels (XB != 0), delay units
    int order = order_env >= 0 ? order_env : (LT <= 32 ? (((long long)n_o * n_tiles < 1024 || NCH == 32) ? 1 : 0) | (late ? 0x40 | (4 << 8) : 0) : 0);
    if (XB != 0 && late_slab > 0 && LT * NCH <= 512 && (long long)n_o * n_tiles >= 512) order |= 0x40 | (late_slab << 8);
    hipLaunchKernelGGL((k_sweep_part<R, DIR, M, NCH, WPS, LT, PF, XB, OPF, KT>), dim3((unsigned)(n_o * n_tiles)), dim3(LT * NCH), lds, c->stream, p, n_o, n_tiles, order);
    return true;
}

template <typename R, int DIR>
static bool part_dispatch_xy(fs3d_ctx *c, const SweepParams<R> &p)
{
    const int n = DIR == 0 ? p.dimx : p.dimy;
    if (n < 4) return false;
    if constexpr (std::is_same<R, float>::value) {       // fp64 contexts run the exact kernels
        static const int variant = getenv("FS3D_PART_VARIANT") ? atoi(getenv("FS3D_PART_VARIANT")) : 0;   // 16 / 32 / 64 lines per workgroup: same chunks, same arithmetic, same bits (tested)
        if (DIR == 0 && p.xiface_pass) {
            // first pass of the cross-slab sweep: the slab's interface words (whole chunks only)
            constexpr int D0 = 0;
            if (!p.carry_out) return false;
            if (n <= 32 && n % 8 == 0) return part_launch_xy<R, D0, 8, 4, 4, 64, FS3D_PART_PF, 2>(c, p);   // thin slabs: 8-cell chunks, every thread has cells
            if (n % 16 != 0) return false;
            const bool wide = (long long)p.dimy * ((p.dimz + 63) / 64) >= 512;      // enough 64-line tiles to fill the chip: 256-byte row pieces
            if (n <= 64) return wide ? part_launch_xy<R, D0, 16, 4, 4, 64, FS3D_PART_PF, 2>(c, p) : part_launch_xy<R, D0, 16, 4, 4, 32, FS3D_PART_PF, 2>(c, p);
            if (n <= 128) return wide ? part_launch_xy<R, D0, 16, 8, 4, 64, FS3D_PART_PF, 2>(c, p) : part_launch_xy<R, D0, 16, 8, 4, 32, FS3D_PART_PF, 2>(c, p);
            if (n <= 256) return part_launch_xy<R, D0, 16, 16, 4, 64, FS3D_PART_PF, 2>(c, p);
            return false;
        }
        if (DIR == 0 && p.carry_in && p.xcarry_in) {
            // x-slab with the values below / above it given (reduced-interface form of the cross-slab sweep)
            constexpr int D0 = 0;
            if (n <= 32) return part_launch_xy<R, D0, 8, 4, 4, 64, FS3D_PART_PF, 1>(c, p);      // thin slabs (a 32-plane slab of the 256^3 box)
            const bool wide = (long long)p.dimy * ((p.dimz + 63) / 64) >= 512;      // enough 64-line tiles to fill the chip: 256-byte row pieces
            if (n <= 64) return wide ? part_launch_xy<R, D0, 16, 4, 4, 64, FS3D_PART_PF, 1>(c, p) : part_launch_xy<R, D0, 16, 4, 4, 32, FS3D_PART_PF, 1>(c, p);
            if (n <= 128) return wide ? part_launch_xy<R, D0, 16, 8, 4, 64, FS3D_PART_PF, 1>(c, p) : part_launch_xy<R, D0, 16, 8, 4, 32, FS3D_PART_PF, 1>(c, p);
            if (n <= 256) return part_launch_xy<R, D0, 16, 16, 4, 64, FS3D_PART_PF, 1>(c, p);
            return false;
        }
        if (n <= 64) return part_launch_xy<R, DIR, 16, 4, 4, 32>(c, p);
        if (n <= 128) return part_launch_xy<R, DIR, 16, 8, 4, 32>(c, p);
        if (n <= 256) {
#ifdef FS3D_EXPERIMENTS
            // measured alternatives (profiles/r2_variants.txt): other chunk sizes = another rounding of the same algebra, so they
            // exist in experiment builds only
            if (variant == 1) return part_launch_xy<R, DIR, 32, 8, 2, 32>(c, p);     // 256 threads x 32 cells, <= 256 VGPRs: 1.15x slower
            if (variant == 10) return part_launch_xy<R, DIR, 8, 32, 4, 32, 2, false, 2, true>(c, p);   // 8 cells per thread, 32 lines, the temp values
                                                                                                    // stay in registers for the merge: 1.1x slower
                                                                                                    // (without keeping them: 1.25x)
#endif
            // 64 lines x 16 chunks (one workgroup of 1024 threads per CU, 256-byte row pieces) was the default until the layer fields
            // were padded; since then two 32-line workgroups per CU are faster (their phases overlap inside the CU)
            if (variant == 64) return part_launch_xy<R, DIR, 16, 16, 4, 64>(c, p);
            if (variant == 32) return part_launch_xy<R, DIR, 16, 16, 4, 32>(c, p);
            // (r3) with the nontemporal streams the X sweep -- rows a plane apart -- is faster again with 256-byte row pieces
            // (64-line tiles: 0.2685 -> 0.2514 ms), the Y sweep stays with two 32-line workgroups per CU (0.2427 vs 0.2482):
            // profiles/r3_ab_env.txt.  Few workgroups (small planes): 32-line tiles, twice as many.
            if (DIR == 0 && variant == 0 && (long long)p.dimy * ((p.dimz + 63) / 64) >= 512) return part_launch_xy<R, DIR, 16, 16, 4, 64>(c, p);
            if (variant == 16) return part_launch_xy<R, DIR, 16, 16, 4, 16>(c, p);      // 16-line tiles: 256 threads, four workgroups per CU, 64-byte row pieces
            return part_launch_xy<R, DIR, 16, 16, 4, 32>(c, p);
        }
        if (n <= 512) return part_launch_xy<R, DIR, 16, 32, 4, 32>(c, p);
    }
    return false;
}


// ------------------------------------------------------------------------------------------------------------
// Z sweep: lanes along the line
// ------------------------------------------------------------------------------------------------------------
// Lane l of a line owns the 4 cells [4l, 4l+4): one 16-byte piece, so a wave-wide access is LI = 64/LPL whole
// contiguous lines (LPL lanes per line: 64 for 128 < dimz <= 256).  A wave works through LG such rows of lines, one
// after the other, with the next row's loads in flight; nothing is shared between waves (no LDS, no barrier).
// Per line:  rows of the 4 cells -> down-/up-sweep over cells 0..2 -> the interface row of cell 3 -> the LPL x LPL
// interface system by parallel cyclic reduction across the lanes (log2(LPL) steps of shuffles) -> back-substitution
// -> x to `next`, merged temp to `temp_out`; the temp values needed by the merge are still in registers.
typedef float pf4 __attribute__((ext_vector_type(4)));
struct PV4 { float v[4]; };
template <int AUX = 0>
__device__ __forceinline__ PV4 pld4(prsrc_t r, unsigned vo, unsigned so)
{
    const pu32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, AUX);
    PV4 o;
    __builtin_memcpy(o.v, &q, 16);
    return o;
}
template <int AUX = 0>
__device__ __forceinline__ void pst4(prsrc_t r, unsigned vo, unsigned so, const float (&v)[4])
{
    pu32x4 q;
    __builtin_memcpy(&q, v, 16);
    __builtin_amdgcn_raw_buffer_store_b128(q, r, vo, so, AUX);
    // A 16-byte store reads its data registers a few cycles after it issues.  hipcc pads a following VALU write of
    // those registers only for stores without a scalar offset; here (SGPR soffset) it reused them two instructions
    // later and the first dword of the NEXT field's data reached memory (seen on gfx950: sporadic, last lanes of a
    // line).  The four values stay live until this statement, which supplies the wait states.
    asm volatile("s_nop 1" : : "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w));
}

struct ZLine { PV4 tc[4], cu[4], wim, wip, wjm, wjp; pu32x2 code; float nve[4], nb[4]; };   // wjm/wjp: only where a wave-wide access holds several lines; nb: NW == 2

// NW == 2 (lines of 260..512 cells): a line is held by a PAIR of waves (waves 2q, 2q+1 of the workgroup: cells [0,256) and
// [256,512)).  Each wave reduces its 64 interface unknowns by itself; the single coupling between the halves (the row of the
// lower wave's last lane <-> the upper wave's first unknown) is carried through the cyclic reduction as one more right-hand
// side per matrix, and a 2x2 system per right-hand side joins the halves (two small LDS exchanges + barriers per line).
template <int LPL, int WPS, int NW = 1>
__global__ void __launch_bounds__(256, WPS) k_sweep_part_z(SweepParams<float> p, int n_grp, int LG)
{
    typedef float R;
    static_assert(NW == 1 || (NW == 2 && LPL == 64), "a pair of waves per line: 64 lanes each");
    constexpr int LI = 64 / LPL;                        // lines per wave-wide access
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = lane % LPL, sub = lane / LPL;
    const int hi = NW == 2 ? (w & 1) : 0;               // upper half of the line
    const int gl = l + 64 * hi;                         // position of this lane's piece along the line
    __shared__ float zx1[2][8], zx2[2][2][8];           // NW == 2: [pair][..] exchange buffers
    int lb = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = lb & 7, slot = lb >> 3;
        lb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + slot;
    }
    // task = (group of LG*LI lines, plane): consecutive tasks = consecutive planes of one line group
    const int task = NW == 2 ? lb * 2 + (w >> 1) : lb * 4 + w;
    const int npl = p.o_count ? p.o_count : p.dimx;    // planes of this launch
    int grp = task / npl;
    const int i = task - grp * npl + p.o_begin;
    bool task_ok = true;
    if (NW == 1) { if (grp >= n_grp) return; }          // whole wave (wave-uniform)
    else { task_ok = grp < n_grp; grp = task_ok ? grp : n_grp - 1; }   // the pairs of a workgroup meet at barriers: a pair past the end runs along, stores nothing
    const int n = p.dimz;
    const int j0 = grp * LG * LI;
    const bool l_ok = 4 * gl < n;
    const int lc = l_ok ? gl : (n / 4 - 1);
    const unsigned fsb = (unsigned)(p.fstride * 4ll), nsb = (unsigned)(p.nstride * 4ll);
    const unsigned rowb = (unsigned)p.dimz * 4u, planeb = (unsigned)(p.plane * 4ll);
    const unsigned lbytes = 4u * fsb;
    const prsrc_t Lcur = __builtin_amdgcn_make_buffer_rsrc((void *)(p.cur_ - p.plane), 0, (int)lbytes, 0x00020000);
    const prsrc_t Ltmp = __builtin_amdgcn_make_buffer_rsrc((void *)(p.temp_ - p.plane), 0, (int)lbytes, 0x00020000);
    const prsrc_t Lnext = __builtin_amdgcn_make_buffer_rsrc((void *)(p.next_ - p.plane), 0, (int)lbytes, 0x00020000);
    const prsrc_t Ltout = __builtin_amdgcn_make_buffer_rsrc((void *)(p.temp_out_ - p.plane), 0, (int)lbytes, 0x00020000);
    const prsrc_t rNode = __builtin_amdgcn_make_buffer_rsrc((void *)p.node_, 0, (int)(4u * nsb), 0x00020000);
    const prsrc_t rCode = __builtin_amdgcn_make_buffer_rsrc((void *)p.code, 0, (int)(nsb / 2u), 0x00020000);

    const R h2s = p.two_ds[2], h2o = p.two_ds[0], h2l = p.two_ds[1], dtv = p.dt;
    const R ir2s = R(1) / h2s, ir2o = R(1) / h2o, ir2l = R(1) / h2l, irdt = R(1) / dtv;
    const R vis_v = p.vis_v, vis_t = p.vis_t, b_v = p.b_v, b_t = p.b_t;

    // byte offset of (plane i, line j, cell 0) inside a layer field (one halo plane first) / inside the node arrays
    auto line_so = [&](int j) __attribute__((always_inline)) { return (unsigned)(((long long)(i + 1) * p.plane + (long long)j * p.dimz) * 4ll); };
    auto line_son = [&](int j) __attribute__((always_inline)) { return (unsigned)(((long long)i * p.plane + (long long)j * p.dimz) * 4ll); };
    const unsigned vo_l = (unsigned)(sub * p.dimz + 4 * lc) * 4u;          // per-lane bytes: own line of the row, own piece
    // the two lanes that hold the ends of the line (cell 0: START or SKIP; cell n-1: END or SKIP) fetch that cell's node
    // values with the line's other loads: every line has them, they must not cost a memory round trip of their own
    const bool is_end = l_ok && (gl == 0 || gl == n / 4 - 1);
    const int ec = gl == 0 ? 0 : 3;
    const unsigned vo_e = is_end ? vo_l + 4u * (unsigned)ec : PART_OOB;
    // NW == 2: the cell across the cut between the two waves (cell 255 for the upper wave's first lane, 256 for the lower
    // wave's last lane) comes with the line's other loads
    const bool at_cut = NW == 2 && (hi ? l == 0 : l == 63);
    const unsigned vo_nb = at_cut ? (hi ? 255u * 4u : 256u * 4u) : PART_OOB;

    auto issue = [&](int jrow, ZLine &L) __attribute__((always_inline)) {
        // jrow: first line of the row (wave-uniform); this lane's line is jrow + sub, clamped into the plane for the loads
        const int jr = jrow < p.dimy ? jrow : p.dimy - 1;   // a row past the plane (tail of the last group): valid addresses, nothing stored
        const unsigned so = opq_s(line_so(jr));
        L.tc[2] = pld4<FS3D_PARTZ_AUX_W>(Ltmp, vo_l, so + 2u * fsb);       // W first: the row before needs it as its j+1 neighbour... and the stencils
        L.tc[0] = pld4<FS3D_PARTZ_AUX_TMP>(Ltmp, vo_l, so); L.tc[1] = pld4<FS3D_PARTZ_AUX_TMP>(Ltmp, vo_l, so + fsb); L.tc[3] = pld4<FS3D_PARTZ_AUX_TMP>(Ltmp, vo_l, so + 3u * fsb);
#pragma unroll
        for (int f = 0; f < 4; f++) L.cu[f] = pld4<FS3D_PART_AUX_CUR>(Lcur, vo_l, so + (unsigned)f * fsb);
        L.wim = pld4(Ltmp, vo_l, so + 2u * fsb - planeb); L.wip = pld4(Ltmp, vo_l, so + 2u * fsb + planeb);
        if (LI > 1) { L.wjm = pld4(Ltmp, vo_l, so + 2u * fsb - rowb); L.wjp = pld4(Ltmp, vo_l, so + 2u * fsb + rowb); }
        const unsigned son = opq_s(line_son(jr));
        L.code = __builtin_amdgcn_raw_buffer_load_b64(rCode, vo_l / 2u, son / 2u, 0);
#pragma unroll
        for (int f = 0; f < 4; f++) L.nve[f] = PBuf<R>::ld(rNode, vo_e, son + (unsigned)f * nsb);     // other lanes: out of range, no memory access
        if (NW == 2) {
#pragma unroll
            for (int f = 0; f < 4; f++) L.nb[f] = PBuf<R>::ld(Ltmp, vo_nb, so + (unsigned)f * fsb);
        }
    };

    // wjm / wjp: W of the lines j-1 / j+1 (LI == 1: the neighbouring rows' registers; else loaded with the line)
    auto process = [&](int jrow, const ZLine &L, const PV4 &wjm, const PV4 &wjp) __attribute__((always_inline)) {
        const int j = jrow + sub;
        const bool st_ok = l_ok && j < p.dimy && task_ok; // lines past the plane compute on whatever was loaded, nothing is stored
        const unsigned so = opq_s(line_so(jrow)), son = opq_s(line_son(jrow));
        // ---- codes
        int code4[4]; bool isin[4], seg[4], inter[4];
        bool all_int = true;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            int cw = (int)((c < 2 ? L.code.x : L.code.y) >> (16 * (c & 1))) & 0xFFFF;
            cw = l_ok ? cw : 0;
            code4[c] = (cw >> 8) & 0xF;
            isin[c] = ((cw >> CODE_TYPE_SHIFT) & 3) == FS3D_NODE_IN && l_ok;
            inter[c] = (code4[c] & 3) == ROW_INTERIOR;
            seg[c] = (code4[c] & 3) != ROW_SKIP;
            all_int = all_int && inter[c];
        }
        // ---- rows: neighbours along the line from the lanes next door
        R tm1[4], tp4[4];                                 // cell 4l-1 and cell 4l+4 of U, V, W, T
#pragma unroll
        for (int f = 0; f < 4; f++) {
            tm1[f] = __shfl_up(L.tc[f].v[3], 1, LPL); tp4[f] = __shfl_down(L.tc[f].v[0], 1, LPL);
            if (NW == 2) { tm1[f] = (at_cut && hi) ? L.nb[f] : tm1[f]; tp4[f] = (at_cut && !hi) ? L.nb[f] : tp4[f]; }
        }
        R q[4], d[4][4];                                  // d[f][c]
#pragma unroll
        for (int c = 0; c < 4; c++) {
            R g[4];
#pragma unroll
            for (int f = 0; f < 4; f++) {
                const R lo = c == 0 ? tm1[f] : L.tc[f].v[c == 0 ? 0 : c - 1], hi = c == 3 ? tp4[f] : L.tc[f].v[c == 3 ? 3 : c + 1];
                g[f] = pdivc(hi - lo, h2s, ir2s);         // d/dz of U, V, W, T
            }
            const R x1 = pdivc(L.wip.v[c] - L.wim.v[c], h2o, ir2o), x2 = pdivc(wjp.v[c] - wjm.v[c], h2l, ir2l);   // dW/dx, dW/dy
            const R diss = (((g[0] * g[0] + g[1] * g[1]) + R(2) * g[2] * g[2]) + g[0] * x1) + g[1] * x2;   // DissFuncZ (TimeLayer3D.h:578-588)
            q[c] = pdivc(L.tc[2].v[c], h2s, ir2s);
            d[0][c] = pdivc(L.cu[0].v[c] * R(3), dtv, irdt); d[1][c] = pdivc(L.cu[1].v[c] * R(3), dtv, irdt);
            d[2][c] = pfma(-p.v_T, g[3], pdivc(L.cu[2].v[c] * R(3), dtv, irdt));
            d[3][c] = pfma(p.t_phi, diss, pdivc(L.cu[3].v[c] * R(3), dtv, irdt));
        }
        PMat<R> mv[4], mt[4];
#pragma unroll
        for (int c = 0; c < 4; c++) part_coefs<R, false>(q[c], 0, vis_v, b_v, vis_t, b_t, mv[c], mt[c]);
        {
            // Rows that are not INTERIOR: the ends of every line (node values came with the line) and, rarely, obstacles /
            // lanes past the line (their node values are fetched here).  Per cell slot c a wave-uniform test: in a box only
            // c = 0 (first lane) and c = 3 (last lane) take the selects.
            bool slow = false;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int kind = code4[c] & 3;
                slow = slow || ((kind == ROW_START || kind == ROW_END) && !(is_end && c == ec));
            }
            PV4 nv[4];
            if (__any(slow)) {
#pragma unroll
                for (int f = 0; f < 4; f++) nv[f] = pld4(rNode, vo_l, son + (unsigned)f * nsb);
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (__any(!inter[c])) {
                    const int kind = code4[c] & 3;
                    const bool ns_v = kind != ROW_SKIP && !(code4[c] & ROW_VELFREE), ns_t = kind != ROW_SKIP && !(code4[c] & ROW_TEMPFREE);
                    const bool pre = is_end && c == ec;
                    part_coefs<R, true>(q[c], code4[c], vis_v, b_v, vis_t, b_t, mv[c], mt[c]);
#pragma unroll
                    for (int f = 0; f < 3; f++) d[f][c] = inter[c] ? d[f][c] : (ns_v ? (pre ? L.nve[f] : nv[f].v[c]) : R(0));
                    d[3][c] = inter[c] ? d[3][c] : (ns_t ? (pre ? L.nve[3] : nv[3].v[c]) : R(0));
                }
            }
        }
        // ---- chunk elimination: down-sweep over cells 0..2 (kept for the back-substitution), up-sweep 2..0
        R cpv[3], lpv[3], cpt[3], lpt[3], dpd[3][4];
        {
            R cv = R(0), lv = R(-1), ct = R(0), lt = R(-1), d3[3] = {R(0), R(0), R(0)}, d1[1] = {R(0)};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                { const R dd3[3] = {d[0][c], d[1][c], d[2][c]}, dd1[1] = {d[3][c]}; part_step_vt<R, FS3D_PART_PACKED_Z != 0>(mv[c].a, mv[c].b, mv[c].c, mt[c].a, mt[c].b, mt[c].c, cv, ct, lv, lt, d3, d1, dd3, dd1); }
                cpv[c] = cv; lpv[c] = lv; cpt[c] = ct; lpt[c] = lt;
                dpd[c][0] = d3[0]; dpd[c][1] = d3[1]; dpd[c][2] = d3[2]; dpd[c][3] = d1[0];
            }
        }
        R apv = R(0), upv = R(-1), apt = R(0), upt = R(-1), ep3[3] = {R(0), R(0), R(0)}, ep1[1] = {R(0)};
#pragma unroll
        for (int c = 2; c >= 0; c--) {
            { const R dd3[3] = {d[0][c], d[1][c], d[2][c]}, dd1[1] = {d[3][c]}; part_step_vt<R, FS3D_PART_PACKED_Z != 0>(mv[c].c, mv[c].b, mv[c].a, mt[c].c, mt[c].b, mt[c].a, apv, apt, upv, upt, ep3, ep1, dd3, dd1); }
        }
        // ---- interface row (cell 3) with x[2] eliminated and x_first of the next lane substituted; normalised
        R av, cv_, at, ct_, dd[4];
        {
            R nvf = __shfl_down(apv, 1, LPL), nwf = __shfl_down(upv, 1, LPL), ntf = __shfl_down(apt, 1, LPL), nuf = __shfl_down(upt, 1, LPL);
            R ng0 = __shfl_down(ep3[0], 1, LPL), ng1 = __shfl_down(ep3[1], 1, LPL), ng2 = __shfl_down(ep3[2], 1, LPL), ng3 = __shfl_down(ep1[0], 1, LPL);
            if (NW == 2) {
                // the lower wave's last lane takes the up-sweep of the upper wave's first lane
                float *const b = zx1[w >> 1];
                if (hi && l == 0) { b[0] = apv; b[1] = upv; b[2] = apt; b[3] = upt; b[4] = ep3[0]; b[5] = ep3[1]; b[6] = ep3[2]; b[7] = ep1[0]; }
                __syncthreads();
                if (at_cut && !hi) { nvf = b[0]; nwf = b[1]; ntf = b[2]; nuf = b[3]; ng0 = b[4]; ng1 = b[5]; ng2 = b[6]; ng3 = b[7]; }
            }
            const bool last = NW == 2 ? (hi && l == 63) : l == LPL - 1;   // no lane behind: its first cell does not exist (the row has c = 0 anyway)
            const R clv = last ? R(0) : mv[3].c, clt = last ? R(0) : mt[3].c;
            const R lov = -mv[3].a * lpv[2], div = pfma(-clv, nvf, pfma(-mv[3].a, cpv[2], mv[3].b)), upv_ = -clv * nwf;
            const R lot = -mt[3].a * lpt[2], dit = pfma(-clt, ntf, pfma(-mt[3].a, cpt[2], mt[3].b)), upt_ = -clt * nuf;
            const R rv = prcp(div), rt = prcp(dit);
            av = pquot(lov, div, rv); cv_ = pquot(upv_, div, rv); at = pquot(lot, dit, rt); ct_ = pquot(upt_, dit, rt);
            dd[0] = pquot(pfma(-clv, ng0, pfma(-mv[3].a, dpd[2][0], d[0][3])), div, rv);
            dd[1] = pquot(pfma(-clv, ng1, pfma(-mv[3].a, dpd[2][1], d[1][3])), div, rv);
            dd[2] = pquot(pfma(-clv, ng2, pfma(-mv[3].a, dpd[2][2], d[2][3])), div, rv);
            dd[3] = pquot(pfma(-clt, ng3, pfma(-mt[3].a, dpd[2][3], d[3][3])), dit, rt);
        }
        // NW == 2: the coupling across the cut leaves the wave's system and becomes a right-hand side of its own
        // (x_l = X_l - Z * E_l with Z the unknown on the other side of the cut)
        R ev = R(0), et = R(0);
        if (NW == 2) {
            if (at_cut && !hi) { ev = cv_; et = ct_; cv_ = R(0); ct_ = R(0); }
            if (at_cut && hi) { ev = av; et = at; av = R(0); at = R(0); }
        }
        // ---- parallel cyclic reduction over the LPL lanes of the line.  (r3) on 2-vectors: (velocity, temperature) matrix words,
        // right-hand sides U/V against the velocity matrix and W/T against (velocity, temperature): the same operations, component for
        // component, as the scalar form (FS3D_PART_PACKED_Z 0), half the VALU issue slots
        if constexpr (FS3D_PART_PACKED_Z != 0) {
            pf2 A = {av, at}, C = {cv_, ct_}, D01 = {dd[0], dd[1]}, D23 = {dd[2], dd[3]}, E = {ev, et};
            auto sh_up = [&](pf2 v, int s) __attribute__((always_inline)) { return pf2{__shfl_up(v.x, s, LPL), __shfl_up(v.y, s, LPL)}; };
            auto sh_dn = [&](pf2 v, int s) __attribute__((always_inline)) { return pf2{__shfl_down(v.x, s, LPL), __shfl_down(v.y, s, LPL)}; };
#pragma unroll
            for (int s = 1; s < LPL; s <<= 1) {
                const pf2 Am = sh_up(A, s), Cm = sh_up(C, s), Ap = sh_dn(A, s), Cp = sh_dn(C, s);
                const pf2 Dm01 = sh_up(D01, s), Dm23 = sh_up(D23, s), Dq01 = sh_dn(D01, s), Dq23 = sh_dn(D23, s);
                const bool has_m = l >= s, has_p = l + s < LPL;
                const pf2 zero = {0.0f, 0.0f}, one = {1.0f, 1.0f};
                const pf2 a = has_m ? A : zero, c = has_p ? C : zero;
                const pf2 dn = pk_fma(-a, Cm, pk_fma(-c, Ap, one));
                pf2 r = {__builtin_amdgcn_rcpf(dn.x), __builtin_amdgcn_rcpf(dn.y)};
                r = pk_fma(pk_fma(-dn, r, one), r, r);
                const pf2 avv = a.xx, cvv = c.xx, dvv = dn.xx, rvv = r.xx;
                D01 = pk_quot(pk_fma(-avv, Dm01, pk_fma(-cvv, Dq01, D01)), dvv, rvv);
                D23 = pk_quot(pk_fma(-a, Dm23, pk_fma(-c, Dq23, D23)), dn, r);
                if (NW == 2) {
                    const pf2 Em = sh_up(E, s), Eq = sh_dn(E, s);
                    E = pk_quot(pk_fma(-a, Em, pk_fma(-c, Eq, E)), dn, r);
                }
                A = pk_quot(-a * Am, dn, r); C = pk_quot(-c * Cp, dn, r);
            }
            av = A.x; at = A.y; cv_ = C.x; ct_ = C.y; dd[0] = D01.x; dd[1] = D01.y; dd[2] = D23.x; dd[3] = D23.y; ev = E.x; et = E.y;
        } else {
#pragma unroll
        for (int s = 1; s < LPL; s <<= 1) {
            const R amv = __shfl_up(av, s, LPL), cmv = __shfl_up(cv_, s, LPL), apv_ = __shfl_down(av, s, LPL), cpv_ = __shfl_down(cv_, s, LPL);
            const R amt = __shfl_up(at, s, LPL), cmt = __shfl_up(ct_, s, LPL), apt_ = __shfl_down(at, s, LPL), cpt_ = __shfl_down(ct_, s, LPL);
            R dm[4], dq[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { dm[k] = __shfl_up(dd[k], s, LPL); dq[k] = __shfl_down(dd[k], s, LPL); }
            // lanes without a partner at this distance: their a (c) is zero by now, the partner values must only be finite
            const bool has_m = l >= s, has_p = l + s < LPL;
            const R a_v = has_m ? av : R(0), c_v = has_p ? cv_ : R(0), a_t = has_m ? at : R(0), c_t = has_p ? ct_ : R(0);
            const R dnv = pfma(-a_v, cmv, pfma(-c_v, apv_, R(1))), dnt = pfma(-a_t, cmt, pfma(-c_t, apt_, R(1)));
            const R rv = prcp(dnv), rt = prcp(dnt);
#pragma unroll
            for (int k = 0; k < 3; k++) dd[k] = pquot(pfma(-a_v, dm[k], pfma(-c_v, dq[k], dd[k])), dnv, rv);
            dd[3] = pquot(pfma(-a_t, dm[3], pfma(-c_t, dq[3], dd[3])), dnt, rt);
            if (NW == 2) {
                const R emv = __shfl_up(ev, s, LPL), eqv = __shfl_down(ev, s, LPL), emt = __shfl_up(et, s, LPL), eqt = __shfl_down(et, s, LPL);
                ev = pquot(pfma(-a_v, emv, pfma(-c_v, eqv, ev)), dnv, rv);
                et = pquot(pfma(-a_t, emt, pfma(-c_t, eqt, et)), dnt, rt);
            }
            av = pquot(-a_v * amv, dnv, rv); cv_ = pquot(-c_v * cpv_, dnv, rv);
            at = pquot(-a_t * amt, dnt, rt); ct_ = pquot(-c_t * cpt_, dnt, rt);
        }
        }
        // ---- back-substitution: x[3] = X, x[c] = d'[c] - l[c] X_left - c'[c] x[c+1]
        R x[4][4];                                        // x[f][c]
        {
            R xcut[4] = {R(0), R(0), R(0), R(0)};         // NW == 2, upper wave: the lower wave's last unknown
            if (NW == 2) {
                // join the halves: X = Xd - Y0 E_lo, Y = Yd - X63 E_hi  ->  per right-hand side a 2x2 system in (X63, Y0)
                float *const bl = zx2[w >> 1][0], *const bh = zx2[w >> 1][1];
                if (at_cut) { float *const b = hi ? bh : bl; b[0] = dd[0]; b[1] = dd[1]; b[2] = dd[2]; b[3] = dd[3]; b[4] = ev; b[5] = et; }
                __syncthreads();
                const R elv = bl[4], elt = bl[5], ehv = bh[4], eht = bh[5];
                const R rdv = prcp(pfma(-elv, ehv, R(1))), rdt = prcp(pfma(-elt, eht, R(1)));
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const R el = k < 3 ? elv : elt, eh = k < 3 ? ehv : eht, den = pfma(-el, eh, R(1));
                    const R x63 = pquot(pfma(-el, bh[k], bl[k]), den, k < 3 ? rdv : rdt);
                    const R y0 = pfma(-x63, eh, bh[k]);
                    dd[k] = pfma(-(hi ? x63 : y0), k < 3 ? ev : et, dd[k]);
                    xcut[k] = x63;
                }
            }
            R xl[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { xl[k] = __shfl_up(dd[k], 1, LPL); xl[k] = l == 0 ? (NW == 2 && hi ? xcut[k] : R(0)) : xl[k]; x[k][3] = dd[k]; }
#pragma unroll
            for (int c = 2; c >= 0; c--) {
#pragma unroll
                for (int k = 0; k < 3; k++) x[k][c] = pfma(-cpv[c], x[k][c + 1], pfma(-lpv[c], xl[k], dpd[c][k]));
                x[3][c] = pfma(-cpt[c], x[3][c + 1], pfma(-lpt[c], xl[3], dpd[c][3]));
            }
        }
        // ---- scatter + merge
        const unsigned vo_st = st_ok ? vo_l : PART_OOB;
        const bool all_seg = seg[0] && seg[1] && seg[2] && seg[3];
        if (p.store_next) {
            if (__all(all_seg || !st_ok)) {
#pragma unroll
                for (int f = 0; f < 4; f++) pst4<FS3D_PART_AUX_ST>(Lnext, vo_st, so + (unsigned)f * fsb, x[f]);
            } else {
#pragma unroll
                for (int f = 0; f < 4; f++)
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        PBuf<R>::st(Lnext, seg[c] ? vo_st : PART_OOB, so + (unsigned)f * fsb + 4u * (unsigned)c, x[f][c]);
            }
        }
        if (p.merge) {
            bool stale = false;
#pragma unroll
            for (int c = 0; c < 4; c++) stale = stale || (isin[c] && !seg[c]);
            if (__any(stale)) {
                // NODE_IN cell outside every segment: the reference merges the stale `next` value (Grid3D.cpp:87-117)
#pragma unroll
                for (int f = 0; f < 4; f++) {
                    const PV4 sv = pld4(Lnext, vo_l, so + (unsigned)f * fsb);
#pragma unroll
                    for (int c = 0; c < 4; c++) x[f][c] = (isin[c] && !seg[c]) ? sv.v[c] : x[f][c];
                }
            }
#pragma unroll
            for (int f = 0; f < 4; f++) {
                R o4[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    R mvv = (L.tc[f].v[c] + x[f][c]) * R(0.5);       // MergeFieldTo (TimeLayer3D.h:415-436)
                    if (p.merge == 2) mvv = (mvv + x[f][c]) * R(0.5);
                    o4[c] = isin[c] ? mvv : L.tc[f].v[c];
                }
                pst4<FS3D_PART_AUX_ST>(Ltout, vo_st, so + (unsigned)f * fsb, o4);
            }
        }
    };

    // ---- the rows of this wave's group, two per trip, the next row's loads in flight
    // LI == 1: W of the lines j-1 / j+1 are the previous / next row's registers; the lines just outside the group come
    // from two extra loads.  (The next row's W is its first load: it has arrived long before the rest.)
    ZLine La, Lb;
    PV4 wprev, wedge;
    if (LI == 1) {
        wprev = pld4(Ltmp, vo_l, opq_s(line_so(j0)) + 2u * fsb - rowb);
        const int jl = j0 + LG < p.dimy ? j0 + LG : p.dimy - 1;
        wedge = pld4(Ltmp, vo_l, opq_s(line_so(jl)) + 2u * fsb);
    }
    issue(j0, La);
    for (int r = 0; r < LG; r += 2) {
        const int ja = j0 + r * LI, jb = ja + LI, jc = jb + LI;
        if (r + 1 < LG) issue(jb, Lb);
        __builtin_amdgcn_sched_barrier(0);
        if (LI == 1) {
            PV4 wn;
#pragma unroll
            for (int c = 0; c < 4; c++) wn.v[c] = wedge.v[c];
            if (r + 1 < LG) {
#pragma unroll
                for (int c = 0; c < 4; c++) wn.v[c] = Lb.tc[2].v[c];
            }
            process(ja, La, wprev, wn);
#pragma unroll
            for (int c = 0; c < 4; c++) wprev.v[c] = La.tc[2].v[c];
        }
        else process(ja, La, La.wjm, La.wjp);
        __builtin_amdgcn_sched_barrier(0);
        if (r + 1 < LG) {
            if (r + 2 < LG) issue(jc, La);
            __builtin_amdgcn_sched_barrier(0);
            if (LI == 1) {
                PV4 wn;
#pragma unroll
                for (int c = 0; c < 4; c++) wn.v[c] = wedge.v[c];
                if (r + 2 < LG) {
#pragma unroll
                    for (int c = 0; c < 4; c++) wn.v[c] = La.tc[2].v[c];
                }
                process(jb, Lb, wprev, wn);
#pragma unroll
                for (int c = 0; c < 4; c++) wprev.v[c] = Lb.tc[2].v[c];
            }
            else process(jb, Lb, Lb.wjm, Lb.wjp);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int LPL, int NW = 1>
static bool part_launch_z(fs3d_ctx *c, const SweepParams<float> &p)
{
    constexpr int LI = 64 / LPL;
    static const int lg_env = part_exp_env("FS3D_PART_ZLG", 0);
    const int rows = (p.dimy + LI - 1) / LI;              // rows of LI lines per plane
    const int npl = p.o_count ? p.o_count : p.dimx;
    // 8 rows per wave where the grid is large enough to give every CU several workgroups that way; fewer on small grids /
    // thin slabs (a 64^3 grid would otherwise launch 32 workgroups for 256 CUs)
    int LG = lg_env > 0 ? lg_env : 16;                   // (r3) 16 rows per wave: 0.219 -> 0.207 ms at 256^3 (profiles/r3_ab_env.txt)
    while (LG > 1 && (long long)((rows + LG - 1) / LG) * npl < 4096) LG >>= 1;
    if (LG > rows) LG = rows;
    const int n_grp = (rows + LG - 1) / LG;
    const long long tasks = (long long)n_grp * (p.o_count ? p.o_count : p.dimx);
    // 2 waves per SIMD: the kernel needs ~200 VGPRs (at 168 it spills 70 of them and runs 1.5x slower)
    if (NW == 2) hipLaunchKernelGGL((k_sweep_part_z<LPL, 2, NW>), dim3((unsigned)((tasks + 1) / 2)), dim3(256), 0, c->stream, p, n_grp, LG);
    else hipLaunchKernelGGL((k_sweep_part_z<LPL, 2, NW>), dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, c->stream, p, n_grp, LG);
    return true;
}

static bool part_dispatch_z(fs3d_ctx *c, const SweepParams<float> &p)
{
    const int n = p.dimz;
    if (n % 4 != 0 || n < 8) return false;                // whole 16-byte pieces
    if (p.dimy < 4) return false;
    if (n <= 64) return part_launch_z<16>(c, p);
    if (n <= 128) return part_launch_z<32>(c, p);
    if (n <= 256) return part_launch_z<64>(c, p);
    if (n <= 512) return part_launch_z<64, 2>(c, p);       // a pair of waves per line
    return false;
}
static bool part_dispatch_z(fs3d_ctx *, const SweepParams<double> &) { return false; }

// false: dims / precision / slab configuration not covered -> the caller falls back to the exact kernels
template <typename R>
bool launch_sweep_part(fs3d_ctx *c, int dir, const SweepParams<R> &p)
{
    if ((unsigned long long)p.fstride * 4ull * sizeof(R) >= (1ull << 32)) return false;   // 32-bit buffer offsets span a layer
    if (dir == 0 && (p.ghost_lo || p.ghost_hi) && !(p.carry_in && p.xcarry_in) && !p.xiface_pass) return false;   // X sweep of an x-slab: only with the values below / above the slab given
    if (dir == 0) return part_dispatch_xy<R, 0>(c, p);
    if (dir == 1) return part_dispatch_xy<R, 1>(c, p);
    return part_dispatch_z(c, p);
}
template bool launch_sweep_part<float>(fs3d_ctx *, int, const SweepParams<float> &);
template bool launch_sweep_part<double>(fs3d_ctx *, int, const SweepParams<double> &);
