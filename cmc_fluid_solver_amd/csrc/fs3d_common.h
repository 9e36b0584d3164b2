// Shared declarations of the HIP implementation behind include/fs3d.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/fs3d.h"

// ---- per-cell code word (uint16) ------------------------------------------------
// bits 0..3   row code of the X sweep
// bits 4..7   row code of the Y sweep
// bits 8..11  row code of the Z sweep
// bits 12..13 NodeType (Geometry.h:31-36)
// row code: bits 0..1 kind, bit 2 velocity BC is FREE, bit 3 temperature BC is FREE
// (BC bits are only meaningful for START/END rows).
// The kinds restate Grid3D::GenerateListSegments (Grid3D.cpp:47-127) per cell: a
// segment is START, INTERIOR.., END along its line; everything else is SKIP.
enum { ROW_SKIP = 0, ROW_INTERIOR = 1, ROW_START = 2, ROW_END = 3 };
#define ROW_VELFREE 4
#define ROW_TEMPFREE 8
#define CODE_TYPE_SHIFT 12

template <typename R>
struct SweepParams {
    // geometry
    int dimx, dimy, dimz;       // owned planes
    long long plane;            // dimy*dimz
    // layers: one allocation per layer, field v at + v*fstride; the pointers address the first
    // OWNED cell of field 0 (one halo plane precedes every field).  Few pointers = few SGPRs.
    const R *cur_;
    const R *temp_;             // temp read by the stencils (old temp)
    R *next_;
    R *temp_out_;               // merged temp (== temp_ when not double-buffered)
    long long fstride;          // elements between consecutive fields of a layer
    const uint16_t *code;
    const uint8_t *dead;        // per line of the sweep direction: 1 = no cell of the line is on a segment or NODE_IN (partition kernels)
    // (r3) shared code columns of the X / Y partition kernels: where every live line of a group of 32 neighbouring lines carries the
    // same row codes (a box: all of them), the group reads ONE column of codes instead of 2 bytes per cell from HBM
    const uint16_t *ucol;       // [column id][UCOL_PITCH] the distinct columns (bits as in `code`), or nullptr -- a box has three of them: hot in the caches
    const unsigned *uflag;      // [o][group]: bit 0 the group is uniform, bit 1 so is the pair (group, group + 1) with equal columns (64-line tiles),
                                // bits 2.. the id of the group's (pair's) column
    int ung;                    // groups per row / plane: ceil(dimz / 32)
    const R *node_;             // node boundary values (v.x, v.y, v.z, T), field v at + v*nstride
    R *scr_;                    // LINE kernel scratch: c'_uvw, c'_T, d'_U, d'_V, d'_W, d'_T at + v*nstride
    long long nstride;          // elements between consecutive fields of node_ / scr_ (>= number of owned cells)
    __host__ __device__ const R *cur(int v) const { return cur_ + v * fstride; }
    __host__ __device__ const R *temp(int v) const { return temp_ + v * fstride; }
    __host__ __device__ R *next(int v) const { return next_ + v * fstride; }
    __host__ __device__ R *temp_out(int v) const { return temp_out_ + v * fstride; }
    __host__ __device__ const R *node(int v) const { return node_ + v * nstride; }
    __host__ __device__ R *scr(int v) const { return scr_ + v * nstride; }
    // constants, all rounded exactly as the reference's FTYPE expressions
    R two_ds[3];                // 2*dx, 2*dy, 2*dz            (TimeLayer3D.h:338-340)
    R vis_v, vis_t;             // v_vis/(ds*ds), t_vis/(ds*ds) (AdiSolver3D.cpp:744-751) for the sweep axis
    R b_v, b_t;                 // 3/dt + 2*vis                 (AdiSolver3D.cpp:761)
    R dt;
    R v_T, t_phi;
    unsigned long long *stamps; // measurement only: per-wave phase time stamps (s_memtime), or nullptr
    // x-slab halves of the X sweep (kernels_line.hip k_xsweep_*, kernels_pipe.hip MODE 1/2); null/0 otherwise
    const R *carry_in; R *carry_out; const R *xcarry_in; R *xcarry_out;
    int ghost_lo, ghost_hi;     // a neighbouring slab's plane stands before / behind the owned planes (X sweep stencils may read it)
    int bundle0;                // first bundle of this launch (line block of the cross-slab pipeline)
    int seg_index, scr_bundles; // halves: scratch slot = seg_index * scr_bundles + bundle
    int seg_begin, seg_len;     // halves: segment of the line this launch works on (seg_len 0: the whole line)
    long long carry_pitch;      // lines per value row of the carry arrays
    int fast_div;               // pipe kernel, fp32: constant divisors are in the range of the division core (kernels_pipe.hip)
    int merge;                  // 0: write next only; 1: also temp_out = merged; 2: merged twice (sweep merge + global merge)
    int o_begin, o_count;       // partition kernels: only the planes [o_begin, o_begin + o_count) of the slab (Y and Z sweeps; o_count 0: all) --
                                // interior planes run beside the halo exchange, the two edge planes after it
    int *errw;                  // device-visible error word (pinned host memory): bit 0 = a relay hand-over of the pipe kernel timed out
    int xiface_pass;            // X partition kernel on an x-slab: 1 = first pass, the slab's 18 interface words per line -> carry_out (pitch carry_pitch)
    int test_drop;              // test hook (env FS3D_TEST_DROP_HANDOFF): one wave never signals its hand-over; the poll bound is short
    int store_next;             // pipe kernel, fused time step: 0 when a later local iteration overwrites `next` unread (only the merge uses x)
};

#define UCOL_PITCH 512                 // codes per shared column (the partition kernels take lines of <= 512 cells)
#define FS3D_XREDUCE_MAX_RANKS 64      // k_xreduce (kernels_line.hip) holds the R x R slab system of a line in per-thread arrays of this size

struct fs3d_ctx {
    int device = 0;
    fs3d_precision prec = FS3D_F32;
    int dimx = 0, dimy = 0, dimz = 0, x_offset = 0, dimx_global = 0;
    double gdx = 0, gdy = 0, gdz = 0;
    double v_T = 1, v_vis = 0, t_vis = 0, t_phi = 0;
    bool have_params = false, have_nodes = false;
    size_t esize = 4;
    long long plane = 0, ncell = 0;
    long long nstride = 0;      // elements between the fields of the node-value / scratch arrays: ncell + padding
    // 5 layer buffers (cur,temp,half,next + spare temp for double-buffering)
    void *lay[5] = {};          // one allocation per layer buffer: 4 fields of fstride elements (lay_raw + a per-layer skew)
    void *lay_raw[5] = {};      // what hipMalloc returned
    long long fstride = 0;      // ncell + 2*plane + padding (the fields / layers of one cell must not share their low address bits)
    int slot[4] = {0, 1, 2, 3}; // layer id -> buffer
    int spare = 4;
    uint16_t *code = nullptr;
    uint16_t *ucol[2] = {};     // shared code columns of the X / Y partition kernels (SweepParams::ucol)
    unsigned *uflag[2] = {};
    uint8_t *dead[3] = {};      // per direction, one byte per line: the line has no segment cell and no NODE_IN cell (X: [j][k], Y: [i][k], Z: [i][j])
    void *node = nullptr;       // 4 x ncell
    void *scr = nullptr;        // >= 6 x ncell: rows of the thread-per-line kernel / of the pipe kernel's halves (allocated on demand)
    size_t scr_bytes = 0;
    // compact list of NODE_BOUND / NODE_VALVE cells (AdiSolver3D.cpp:286-311)
    int *bnd_idx = nullptr;
    void *bnd_val[4] = {};
    int n_bnd = 0;
    int nseg[3] = {0, 0, 0};
    long long stale_in_cells = 0;  // NODE_IN cells on no segment of some direction (they merge stale `next` values): 0 for closed geometries
    // div error partials
    double *red_buf = nullptr;     // device
    double *red_host = nullptr;    // pinned
    int red_blocks = 0;
    double diffError = 0.0;
    int test_drop = 0;             // fault-injection hook of tests/test_gpu_failures.py: env FS3D_TEST_DROP_HANDOFF, read ONCE at fs3d_create
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;     // multi-GPU: halo planes travel here, beside the interior planes' sweep on `stream`
    hipStream_t xstream = nullptr;         // the stream the transport works on right now (stream or comm_stream)
    hipEvent_t ev_src = nullptr, ev_halo = nullptr;
    int opt_overlap = 1;                   // FS3D_OPT_OVERLAP
    int opt_keep_temp = 0;                 // FS3D_OPT_KEEP_TEMP
    // options
    int opt_kernel = FS3D_SWEEP_AUTO;
    int ran_kernel[3] = {0, 0, 0};   // per direction: the kernel the last sweep really ran (fs3d_last_sweep_kernel)
    int ran_segmented[3] = {0, 0, 0};
    int opt_fuse = 1;
    // timing
    bool timing = false;           // events around the launches being enqueued now
    int timing_period = 0;         // fs3d_enable_timing(on): 0 off, 1 every time step, N every N-th time step (sampling)
    long timing_steps = 0;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    std::vector<int> ev_class;
    // device time per event of the reference's Profiler vocabulary (AdiSolver3D.cpp:297-367, 555-680): 0 SolveSegments_Z, 1 _Y, 2 _X,
    // 3 CopyLayer, 4 MergeLayer, 5 EvalDivError, 6 UpdateBoundaries, 7 syncHalos
    float t_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int t_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double t_create_segments_ms = 0;       // host time of fs3d_upload_nodes (CreateSegments + Init_GPU)
    unsigned long long *stamps = nullptr;   // device buffer for fs3d_profile_sweep
    int stamps_cap = 0;
    // comm
    void *comm = nullptr;          // ncclComm_t
    void *local = nullptr;         // fs3d_local_group* (in-process transport)
    void *carry[4] = {};           // cross-slab X sweep: fwd in/out (6 x plane), bwd in/out (4 x plane)
    int opt_div_core = 1;          // FS3D_OPT_DIV_CORE
    void *seg_carry[2] = {};       // segmented long-line sweeps: forward (6 x lines) and backward (4 x lines) carries
    long long seg_carry_lines = 0;
    int *redo = nullptr;           // pipe kernel, fp32: per-bundle "compute again with full divisions" flags (all zero between sweeps)
    int redo_cap = 0;
    int *errw_host = nullptr, *errw_dev = nullptr;   // pinned + mapped error word of the kernels (checked at every synchronisation)
    int rank = 0, nranks = 1;
    void *xif_send = nullptr, *xif_all = nullptr;   // reduced-interface X sweep: this slab's 18 words per line / all ranks'
    void *xa2a[4] = {};            // distributed interface solve: packed words out / in, boundary values out / in ([rank][words][lines per rank])
    int opt_xsolve = 0;            // FS3D_OPT_XSOLVE: 0 auto, 1 pipelined (bit-exact), 2 reduced interface
    int ran_xsolve = 0;            // what the last cross-slab X sweep ran: 1 pipelined, 2 reduced interface, 3 reduced interface with the interface words from the partition kernel
    int ran_xa2a = 0;              // ... and whether the interface solve was distributed over the ranks (two all-to-alls instead of one all-gather)
    int xblocks = 4;               // line blocks of the cross-slab X sweep pipeline (env FS3D_XBLOCKS)
    std::string err;
};

// kernels_*.hip
template <typename R> void launch_sweep_line(fs3d_ctx *c, int dir, const SweepParams<R> &p);
template <typename R> bool launch_sweep_pipe(fs3d_ctx *c, int dir, const SweepParams<R> &p); // false: dims unsupported
// kernels_part.hip: partition (reduced-interface) solve, results to a stated tolerance; false: dims / precision unsupported
template <typename R> bool launch_sweep_part(fs3d_ctx *c, int dir, const SweepParams<R> &p);
// X sweep halves of an x-slab for the bundles [b0, b1) (64 lines each, line = j*dimz + k); false: dims unsupported
template <typename R> bool xslab_pipe_supported(const SweepParams<R> &p);
template <typename R> bool launch_xslab_pipe(fs3d_ctx *c, SweepParams<R> p, int half, int b0, int b1);
// lines longer than the pipe kernel holds: the sweep as a sequence of segment halves on one GPU; false: unsupported
template <typename R> bool launch_sweep_pipe_segmented(fs3d_ctx *c, int dir, SweepParams<R> p);
template <typename R> void launch_xsweep_fwd(fs3d_ctx *c, const SweepParams<R> &p, const void *carry_in, void *carry_out, long long l0, long long l1);
// reduced-interface cross-slab X sweep: interface coefficients of this slab (18 words per line), the R x R interface solve
template <typename R> void launch_xiface(fs3d_ctx *c, const SweepParams<R> &p, void *out);
template <typename R> void launch_xreduce(fs3d_ctx *c, const void *all, long long nl, int nranks, int me, void *carry_in, void *xcarry_in);
template <typename R> void launch_xpack(fs3d_ctx *c, const void *in, long long nl, long long lp, void *out);
template <typename R> void launch_xreduce_a2a(fs3d_ctx *c, const void *all, long long nl, long long lp, int nranks, int me, void *out);
template <typename R> void launch_xunpack(fs3d_ctx *c, const void *in, long long nl, long long lp, void *carry_in, void *xcarry_in);
template <typename R> void launch_xsweep_bwd(fs3d_ctx *c, const SweepParams<R> &p, const void *xcarry_in, void *xcarry_out, long long l0, long long l1);
