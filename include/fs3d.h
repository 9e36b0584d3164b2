/*
 * fs3d.h -- C ABI of the MI355X-native FluidSolver3D hot path (libfs3d_hip.so).
 *
 * This is the drop-in boundary.  The reference has no FFI layer; its backend seam
 * is the set of free functions the host classes call when hw == GPU and that the
 * .cu files define.  Each entry point below names the
 * reference interface it replaces (paths relative to /root/reference/src).
 *
 * Conventions
 *  - plain C types only; all arrays are host pointers unless the name says dev.
 *  - cell index = i*dimy*dimz + j*dimz + k, k unit-stride (TimeLayer3D.h:256-259).
 *  - "real" arrays are float when the context precision is FS3D_F32 and double
 *    when FS3D_F64 (the reference's compile-time FTYPE, Geometry.h:21).
 *  - every call returns an fs3d_status; fs3d_last_error() gives the message
 *    (the reference throws std::runtime_error from gpuSafeCall, GPUplan.cpp:173-193;
 *    the C++ host wrapper in cmc_fluid_solver_amd/host re-throws from the status).
 *  - calls are synchronous with respect to their results (as every reference
 *    launcher ends in deviceSynchronize, e.g. AdiSolver3D.cu:520) unless stated.
 *  - one context drives one GPU (one x-slab).  Multi-GPU = one process per GPU,
 *    one context each, joined by fs3d_comm_init (RCCL).
 */
#ifndef FS3D_H
#define FS3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fs3d_ctx fs3d_ctx;

typedef enum { FS3D_F32 = 0, FS3D_F64 = 1 } fs3d_precision;

typedef enum {
    FS3D_OK = 0,
    FS3D_ERR_INVALID = 1,      /* bad argument / call order */
    FS3D_ERR_HIP = 2,          /* HIP runtime error (message has device id + hip error) */
    FS3D_ERR_DIVERGED = 3,     /* diffError > ERR_THRESHOLD (AdiSolver3D.cpp:371-374 throws) */
    FS3D_ERR_UNSUPPORTED = 4,  /* geometry/feature outside what the kernels implement */
    FS3D_ERR_COMM = 5          /* RCCL error */
} fs3d_status;

/* Geometry.h:29-43 */
enum { FS3D_NODE_IN = 0, FS3D_NODE_OUT = 1, FS3D_NODE_BOUND = 2, FS3D_NODE_VALVE = 3 };
enum { FS3D_BC_NOSLIP = 0, FS3D_BC_FREE = 1 };
enum { FS3D_DIR_X = 0, FS3D_DIR_Y = 1, FS3D_DIR_Z = 2 };
/* the solver's four TimeLayer3D objects (AdiSolver3D.cpp:254-258) */
enum { FS3D_LAYER_CUR = 0, FS3D_LAYER_TEMP = 1, FS3D_LAYER_HALF = 2, FS3D_LAYER_NEXT = 3 };
enum { FS3D_VAR_U = 0, FS3D_VAR_V = 1, FS3D_VAR_W = 2, FS3D_VAR_T = 3 };

/* kernel selection for the line sweeps (fs3d_set_option(FS3D_OPT_SWEEP_KERNEL)) */
enum {
    FS3D_SWEEP_AUTO = 0,      /* fastest kernel that supports the dims: PART where it applies (fp32), else as EXACT */
    FS3D_SWEEP_LINE = 1,      /* thread-per-line Thomas, c'/d' scratch in HBM (any dims); bit-equal to the CPU path */
    FS3D_SWEEP_PIPE = 2,      /* wave-pipelined Thomas, c'/d' in registers+LDS; bit-equal to the CPU path */
    FS3D_SWEEP_PART = 3,      /* partition (reduced-interface) solve: every chunk of a line eliminated at once; same
                                 equations, different rounding -- equal to the CPU path to a stated tolerance
                                 (DESIGN.md section 5), not bit for bit; errors where it does not apply */
    FS3D_SWEEP_EXACT = 4      /* fastest of the bit-exact kernels (PIPE, its segmented form, LINE) */
};  /* environment FS3D_DEFAULT_KERNEL=<id> sets the initial value of FS3D_OPT_SWEEP_KERNEL for new contexts */
enum {
    FS3D_OPT_SWEEP_KERNEL = 0,
    FS3D_OPT_FUSE_MERGE = 1,  /* 1 (default): merge fused into the sweep; 0: separate merge kernels */
    FS3D_OPT_OVERLAP = 4,     /* 1 (default): in a multi-GPU group the halo planes of a Y / Z sweep travel on a second stream beside the
                                 sweep of the interior planes, the two edge planes follow; 0: exchange first, then one launch */
    FS3D_OPT_XSOLVE = 3,      /* cross-slab X sweep of a multi-GPU group: 1 = pipelined over the ranks (the reference's form,
                                 AdiSolver3D.cu:524-640; bit-equal to one GPU), 2 = reduced interface (every rank eliminates its slab
                                 at once; equal to one GPU to rounding) with ONE all-gather per sweep, every rank solving every line's
                                 interface system, 3 = reduced interface with the interface solve distributed over the ranks (two
                                 all-to-alls of point-to-point transfers; bit-identical to 2), 0 (default) = 3 from three ranks on,
                                 2 for two ranks -- unless the sweep-kernel option asks for the bit-exact kernels (then 1) */
    FS3D_OPT_KEEP_TEMP = 5,   /* 0 (default): the merged temp of the LAST sweep of a fused time step is not computed or stored -- nothing
                                 reads it: the next step starts from temp := cur (AdiSolver3D.cpp:320), GetLayer and EvalDivError read
                                 `next`; FS3D_LAYER_TEMP then holds the iterate before that last merge.  1: store it, as the
                                 reference's private `temp` member holds it after TimeStep (the parity tests that download it) */
    FS3D_OPT_DIV_CORE = 2     /* 1 (default): fp32 pipe kernel divides with the scaling-free core of the IEEE expansion and
                                 falls back to the full division where an operand needs scaling (same results); 0: always full */
};

/* ---- lifetime ---------------------------------------------------------------
 * Replaces GPUplan::init / multiDevAlloc of layers, scratch and tables
 * (GPUplan.cpp:35-77, AdiSolver3D.cpp:166-268 AdiSolver3D::Init).
 * dimx is the number of x-planes this context owns; x_offset/dimx_global place the
 * slab in the global grid (PARAplan::getOffset1D/getLength1D, PARAplan.cpp:71-126).
 * Single GPU: x_offset = 0, dimx_global = dimx.  dx,dy,dz are Grid3D::dx.. (double,
 * cast to FTYPE as TimeLayer3D does, TimeLayer3D.h:1078-1080). */
fs3d_status fs3d_create(fs3d_ctx **out, int device, fs3d_precision prec,
                        int dimx, int dimy, int dimz, double dx, double dy, double dz,
                        int x_offset, int dimx_global);
void fs3d_destroy(fs3d_ctx *ctx);
/* message of the last failing call on ctx (ctx == NULL: last fs3d_create failure) */
const char *fs3d_last_error(const fs3d_ctx *ctx);

/* FluidParams (Geometry.h:538-562) as passed by value into every sweep launch
 * (AdiSolver3D.h:40-41).  Values are the FTYPE members widened to double. */
fs3d_status fs3d_set_params(fs3d_ctx *ctx, double v_T, double v_vis, double t_vis, double t_phi);

fs3d_status fs3d_set_option(fs3d_ctx *ctx, int option, int value);

/* ---- geometry ---------------------------------------------------------------
 * Replaces Grid3D::Init_GPU (node-type upload, Grid3D.cpp:526-565) together with
 * AdiSolver3D::CreateSegments (AdiSolver3D.cpp:393-473, 553-562: segment lists and
 * NodesBoundary3D records built from the Node array and copied to the device).
 * Input is the slab's Node array as SoA (Grid3D.h:73-88): type, bc_vel, bc_temp
 * (uint8) and v.x, v.y, v.z, T (real).  Segment semantics are those of
 * Grid3D::GenerateListSegments (Grid3D.cpp:47-127); the device keeps them as per-cell
 * row codes instead of 40-byte Segment3D records.  n_seg_out[3] (optional) receives
 * the segment counts per direction X,Y,Z for cross-checking with the caller's lists.
 * All seven arrays cover the GLOBAL grid (dimx_global*dimy*dimz cells), as every rank of
 * the reference holds the whole Grid3D and clips global segments to its slab
 * (AdiSolver3D.cpp:475-524); the context keeps only its own planes. */
fs3d_status fs3d_upload_nodes(fs3d_ctx *ctx, const uint8_t *type, const uint8_t *bc_vel,
                              const uint8_t *bc_temp, const void *vx, const void *vy,
                              const void *vz, const void *T, int n_seg_out[3]);

/* ---- layers -----------------------------------------------------------------
 * TimeLayer3D(backend, grid) constructor: cur <- every node's v and T
 * (TimeLayer3D.h:736-780 CopyFromGrid, :1076-1090); other layers zero. */
fs3d_status fs3d_init_layers_from_nodes(fs3d_ctx *ctx);
/* multiDevMemcpy H2D / D2H of one layer's four fields (GPUplan.h:110-179,
 * TimeLayer3D::CopyLayerTo, TimeLayer3D.h:685-724).  NULL pointers are skipped. */
fs3d_status fs3d_upload_layer(fs3d_ctx *ctx, int layer, const void *u, const void *v,
                              const void *w, const void *T);
fs3d_status fs3d_download_layer(fs3d_ctx *ctx, int layer, void *u, void *v, void *w, void *T);
/* device pointer of one field (owned planes only), for zero-copy interop
 * (ScalarField3D::getMultiArray, TimeLayer3D.h:267-270). */
fs3d_status fs3d_field_dev_ptr(fs3d_ctx *ctx, int layer, int var, void **dev_ptr);

/* ---- the hot path -------------------------------------------------------------
 * AdiSolver3D::UpdateBoundaries (AdiSolver3D.cpp:286-304): re-impose NODE_BOUND and
 * NODE_VALVE node values into cur (CPU semantics: CopyFromGrid of both types). */
fs3d_status fs3d_update_boundaries(fs3d_ctx *ctx);

/* AdiSolver3D::TimeStep (AdiSolver3D.cpp:306-391) with CPU-backend ordering (all
 * four variables of a sweep read the same temp; merge after the sweep).
 * compute_error != 0 evaluates EvalDivError on next; *err_out (optional) receives the
 * solver's diffError member (last evaluated value).  Returns FS3D_ERR_DIVERGED, and
 * does NOT swap cur/next, when diffError > 0.01 -- where the reference throws. */
fs3d_status fs3d_time_step(fs3d_ctx *ctx, double dt, int num_global, int num_local,
                           int compute_error, double *err_out);

/* Asynchronous variant used by the benchmark loop: enqueues the step on the
 * context's stream and returns; no divergence check, no host sync.  Call
 * fs3d_synchronize() before reading results. */
fs3d_status fs3d_time_step_async(fs3d_ctx *ctx, double dt, int num_global, int num_local);
fs3d_status fs3d_synchronize(fs3d_ctx *ctx);

/* One sweep = one pass of SolveSegments_GPU (AdiSolver3D.h:40-41; CPU semantics of
 * AdiSolver3D.cpp:593-603) over every segment of `dir`: reads layers l_cur and l_temp,
 * writes l_next.  merge_into_temp != 0 also performs the following
 * next->MergeLayerTo(grid, temp, NODE_IN) (AdiSolver3D.cpp:651).  Exposed for
 * kernel-level parity tests and profiling. */
fs3d_status fs3d_sweep(fs3d_ctx *ctx, int dir, double dt, int l_cur, int l_temp, int l_next,
                       int merge_into_temp);
/* TimeLayer3D::MergeLayerTo(grid, dest, NODE_IN), TimeLayer3D.h:664-683 / MergeFieldTo_GPU */
fs3d_status fs3d_merge(fs3d_ctx *ctx, int l_src, int l_dest);
/* TimeLayer3D::EvalDivError (TimeLayer3D.h:595-641): mean |div| over NODE_IN cells.
 * count_out (optional) receives the number of cells summed.  In a multi-GPU group the
 * sum and count are all-reduced (the reference's MPI_Reduce+Bcast, :630-637). */
fs3d_status fs3d_eval_div_error(fs3d_ctx *ctx, int layer, double *err_out, long long *count_out);

/* Solver3D::GetLayer (Solver3D.cpp:21-25): next->Clear(NODE_OUT -> MISSING_VALUE 99999)
 * then FilterToArrays (TimeLayer3D.h:819-924): nearest-neighbour down-sample into
 * outV (Vec3D[] = interleaved real x,y,z) and outT (double[]).  outdim == 0 means the
 * layer's own dim.  Reads `next`, which after the swap in TimeStep is the previous
 * step's layer -- the reference's one-step output lag is preserved. */
fs3d_status fs3d_get_layer(fs3d_ctx *ctx, void *outV, double *outT,
                           int outdimx, int outdimy, int outdimz);

/* ---- multi-GPU (one process per GPU, x-slabs) ---------------------------------
 * Replaces GPUplan peer copies / PARAplan MPI point-to-point (TimeLayer3D.h:47-247,
 * 272-335) with RCCL.  unique_id is the 128-byte ncclUniqueId produced by
 * fs3d_comm_unique_id() on rank 0 and broadcast by the caller (e.g. torch.distributed). */
fs3d_status fs3d_comm_unique_id(void *unique_id_128);
fs3d_status fs3d_comm_init(fs3d_ctx *ctx, const void *unique_id_128, int rank, int nranks);

/* The reference's other multi-GPU mode: ONE process driving several GPUs (GPUplan, Common/GPUplan.h:29-108,
 * `multiDev*` helpers).  Here: one slab context per GPU, each driven by its own host thread, joined by an
 * in-process group -- halos and carries move as device-to-device copies with a host rendezvous instead of
 * RCCL.  Every collective call (time_step, sweep, eval_div_error ...) must then be made by all the
 * group's threads concurrently, as with RCCL ranks.  Contexts may share a device (used to exercise the
 * slab protocol on a single card).  Destroy the contexts before the group. */
fs3d_status fs3d_local_group_create(int nranks, void **group_out);
void fs3d_local_group_destroy(void *group);
/* a driver thread that fails before it holds a context (Init threw) releases the others: every member's pending and later
 * exchange returns FS3D_ERR_COMM */
void fs3d_local_group_abort(void *group);
fs3d_status fs3d_comm_init_local(fs3d_ctx *ctx, void *group, int rank);
/* A rank / slab thread that cannot go on (its own call failed, its driver thread threw) takes the group down: the peers'
 * pending and later exchanges return FS3D_ERR_COMM instead of blocking (the reference: gpuSafeCall / mpiSafeCall throw,
 * main catches and calls MPI_Abort, GPUplan.cpp:173-193, FluidSolver3D.cpp:272-283).  The library calls it itself when a
 * multi-step exchange (the cross-slab X sweep) fails half way. */
fs3d_status fs3d_comm_abort(fs3d_ctx *ctx);
/* Wire check on one card: a communicator of ONE rank is created on the context's device and the three RCCL shapes the slab
 * protocol uses are run and verified -- a grouped ncclSend/ncclRecv pair (the halo-plane / carry-row shape, to the own rank,
 * on the exchange stream), ncclAllGather (the interface words of the cross-slab X solve) and the 2-double ncclAllReduce of
 * EvalDivError.  Says whether librccl loads, initialises and moves bytes inside this library (beside the caller's own RCCL);
 * it does not replace a run on several GPUs.  `elems` elements of the context's precision per message. */
fs3d_status fs3d_comm_selftest(fs3d_ctx *ctx, size_t elems);

/* ---- measurement ---------------------------------------------------------------
 * Wall time of the kernels of the last fs3d_time_step* call, measured with HIP events
 * on the context's stream: ms[0]=Z sweeps, [1]=Y sweeps, [2]=X sweeps, [3]=everything
 * else; n[] = number of launches in each class.  Profiler.h event names map onto these. */
fs3d_status fs3d_last_step_timing(fs3d_ctx *ctx, float ms[4], int n[4]);
/* The same device times under the event names of the reference's Profiler (Common/Profiler.h:44-134; StartEvent/StopEvent
 * sites AdiSolver3D.cpp:297-367, 555-680): SolveSegments_Z/_Y/_X, CopyLayer, MergeLayer (zero launches while the merge is fused
 * into the sweeps), EvalDivError, UpdateBoundaries, syncHalos, CreateSegments (host time of fs3d_upload_nodes).  names[] receives
 * static strings.  The driver prints them as the reference's PrintTimings table. */
#define FS3D_N_EVENTS 9
fs3d_status fs3d_profiler_events(fs3d_ctx *ctx, const char *names[FS3D_N_EVENTS], float ms[FS3D_N_EVENTS], int n[FS3D_N_EVENTS]);
/* per-class event timing: 0 off (default), 1 two events around every launch, N > 1 the same on every N-th time step only
 * (a sample: the events themselves cost a few microseconds per launch) */
fs3d_status fs3d_enable_timing(fs3d_ctx *ctx, int on);

/* One pipelined sweep (merge fused, result discarded into the spare temp buffer) with
 * in-kernel time stamps: for each of the first *n_blocks_out (<= max_blocks) workgroups and
 * each of its 8 waves, 8 shader-clock stamps (start, rows built, relay turn begins, forward
 * done, backward begins, backward done, relay drained, stores issued).
 * stamps_out holds max_blocks*64 values.  Measurement aid; no reference counterpart. */
fs3d_status fs3d_profile_sweep(fs3d_ctx *ctx, int dir, double dt, int l_cur, int l_temp, int l_next,
                               unsigned long long *stamps_out, int max_blocks, int *n_blocks_out);

/* Which kernel the last sweep of direction dir (FS3D_DIR_*) really ran: FS3D_SWEEP_LINE / _PIPE / _PART, with
 * *segmented_out (optional): bit 0 = PIPE / LINE ran as halves through the HBM scratch (long lines, x-slabs); for dir X
 * of a multi-GPU group bits 1-2 = the cross-slab form that ran (1 pipelined, 2 reduced interface, 3 reduced interface with
 * the slab's interface words from a first pass of the partition kernel instead of the per-line walk).  0 = no sweep yet.
 * FS3D_SWEEP_AUTO never falls back silently: callers (bench.py, fs3d_run) print this. */
fs3d_status fs3d_last_sweep_kernel(fs3d_ctx *ctx, int dir, int *kernel_out, int *segmented_out);

/* library / device identification */
const char *fs3d_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FS3D_H */
