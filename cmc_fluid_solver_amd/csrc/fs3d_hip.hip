// libfs3d_hip.so -- C ABI (include/fs3d.h) of the MI355X-native FluidSolver3D hot path:
// context, geometry tables, auxiliary kernels and the time-step orchestration.
// The line-sweep kernels live in kernels_line.hip / kernels_pipe.hip.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>

#include "fs3d_common.h"
#include "fs3d_comm.h"

#define FS3D_VERSION "fs3d-hip 0.1 (gfx950)"
static thread_local std::string g_create_err;

static fs3d_status fail(fs3d_ctx *c, fs3d_status st, const std::string &msg)
{
    if (c) c->err = msg; else g_create_err = msg;
    return st;
}

// gpuSafeCall (GPUplan.cpp:173-193): message carries the device id and the runtime's error text
#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            char b_[512];                                                                            \
            snprintf(b_, sizeof b_, "GPU %d: %s failed: %s", (c) ? (c)->device : -1, #call,          \
                     hipGetErrorString(e_));                                                         \
            return fail((c), FS3D_ERR_HIP, b_);                                                      \
        }                                                                                            \
    } while (0)

// ---------------------------------------------------------------------------------
// auxiliary kernels
// ---------------------------------------------------------------------------------

// TimeLayer3D::MergeLayerTo(grid, dest, NODE_IN) (TimeLayer3D.h:415-436, 664-683;
// GPU twin `merge`, TimeLayer3D.cu:133-146), all four fields in one pass.
template <typename R>
__global__ void __launch_bounds__(256) k_merge(const uint16_t *__restrict__ code, long long n,
                                                const R *s0, const R *s1, const R *s2, const R *s3,
                                                R *d0, R *d1, R *d2, R *d3)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (((code[i] >> CODE_TYPE_SHIFT) & 3) == FS3D_NODE_IN) {
            d0[i] = (d0[i] + s0[i]) / R(2);
            d1[i] = (d1[i] + s1[i]) / R(2);
            d2[i] = (d2[i] + s2[i]) / R(2);
            d3[i] = (d3[i] + s3[i]) / R(2);
        }
    }
}

// cur->CopyLayerTo(grid, next, NODE_BOUND / NODE_VALVE) (AdiSolver3D.cpp:310-311;
// TimeLayer3D.h:394-413): the BOUND/VALVE cells are a compact index list here.
template <typename R>
__global__ void __launch_bounds__(256) k_copy_list(const int *__restrict__ idx, int n,
                                                    const R *s0, const R *s1, const R *s2, const R *s3,
                                                    R *d0, R *d1, R *d2, R *d3)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { int i = idx[t]; d0[i] = s0[i]; d1[i] = s1[i]; d2[i] = s2[i]; d3[i] = s3[i]; }
}

// cur->CopyFromGrid(grid, NODE_BOUND / NODE_VALVE) (AdiSolver3D.cpp:292-293;
// TimeLayer3D.h:926-951): node values of the listed cells, stored compactly.
template <typename R>
__global__ void __launch_bounds__(256) k_impose_list(const int *__restrict__ idx, int n,
                                                      const R *v0, const R *v1, const R *v2, const R *v3,
                                                      R *d0, R *d1, R *d2, R *d3)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { int i = idx[t]; d0[i] = v0[t]; d1[i] = v1[t]; d2[i] = v2[t]; d3[i] = v3[t]; }
}

// both of the above in one pass, for UpdateBoundaries directly followed by TimeStep (fs3d_time_step_async): the node values of the
// listed cells into cur AND next (next[i] = cur[i] = v)
template <typename R>
__global__ void __launch_bounds__(256) k_impose_list2(const int *__restrict__ idx, int n,
                                                       const R *v0, const R *v1, const R *v2, const R *v3,
                                                       R *d0, R *d1, R *d2, R *d3, R *e0, R *e1, R *e2, R *e3)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        int i = idx[t];
        const R a = v0[t], b = v1[t], c = v2[t], d = v3[t];
        d0[i] = a; d1[i] = b; d2[i] = c; d3[i] = d; e0[i] = a; e1[i] = b; e2[i] = c; e3[i] = d;
    }
}

// TimeLayer3D::Clear(grid, NODE_OUT, MISSING_VALUE x4) (TimeLayer3D.h:974-999; `clear`, TimeLayer3D.cu:98-116)
template <typename R>
__global__ void __launch_bounds__(256) k_clear_type(const uint16_t *__restrict__ code, long long n, int type, R val,
                                                     R *d0, R *d1, R *d2, R *d3)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        if (((code[i] >> CODE_TYPE_SHIFT) & 3) == type) { d0[i] = val; d1[i] = val; d2[i] = val; d3[i] = val; }
}

// TimeLayer3D::EvalDivError (TimeLayer3D.h:595-641): per-cell FTYPE face sums, double
// accumulation.  Wave64 shuffle reduction, one partial (sum,count) pair per block;
// the partials are summed in a fixed order by k_div_final, so the result is
// run-to-run deterministic (it is NOT the reference's serial summation order: compare
// with a relative tolerance of ~1e-12).
// (r3) One block = one plane i x DIVE_JS rows j x all k: a thread walks its k column down the rows, keeping row j-1 in
// registers and taking the k-1 values from the lane next door, so every U/V/W value is fetched once per plane pair (6 loads
// per cell instead of 24 through the caches; no integer divisions).  The per-cell expression is the reference's, term for term.
#define DIVE_JS 16
template <typename R>
__global__ void __launch_bounds__(256) k_div_error(const uint16_t *__restrict__ code,
                                                    const R *__restrict__ U, const R *__restrict__ V,
                                                    const R *__restrict__ W, int dimx, int dimy, int dimz,
                                                    int i_end, int i_skip0, R dx, R dy, R dz, double *partial, int nj)
{
    const long long plane = (long long)dimy * dimz;
    const int i = (int)(blockIdx.x / nj), j0 = (int)(blockIdx.x % nj) * DIVE_JS;
    double err = 0.0, cnt = 0.0;
    if (i < i_end && !(i_skip0 && i == 0)) {
        const R *const F[3] = {U, V, W};
        for (int kb = 0; kb < dimz; kb += 256) {
            const int k = kb + (int)threadIdx.x;
            const bool kin = k < dimz;
            const int kc = kin ? k : dimz - 1;
            // values of row j-1: [field][plane i / i-1][k / k-1]
            R pv[3][2][2];
            auto load_row = [&](int j, R (&v)[3][2][2]) __attribute__((always_inline)) {
                const long long a = (long long)i * plane + (long long)j * dimz + kc;
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    const R hi = F[f][a], lo = F[f][a - plane];
                    R hm = __shfl_up(hi, 1, 64), lm = __shfl_up(lo, 1, 64);
                    if ((threadIdx.x & 63) == 0) { hm = kc > 0 ? F[f][a - 1] : hi; lm = kc > 0 ? F[f][a - plane - 1] : lo; }
                    v[f][0][0] = hi; v[f][0][1] = hm; v[f][1][0] = lo; v[f][1][1] = lm;
                }
            };
            const int jbeg = j0 > 0 ? j0 : 1;
            if (jbeg < dimy - 1 && jbeg < j0 + DIVE_JS) load_row(jbeg - 1, pv);
            for (int j = jbeg; j < j0 + DIVE_JS && j < dimy - 1; j++) {
                R cv[3][2][2];
                load_row(j, cv);
                const bool in = kin && k > 0 && k < dimz - 1 &&
                                ((code[(long long)i * plane + (long long)j * dimz + kc] >> CODE_TYPE_SHIFT) & 3) == FS3D_NODE_IN;
                if (in) {
                    // a = (j,k), b = (j-1,k), c = (j-1,k-1), d = (j,k-1); "- m" = plane i-1   (TimeLayer3D.h:610-632)
#define DV(f, row, pl, km) (row ? pv : cv)[f][pl][km]
                    const double ex = (double)((DV(0, 0, 0, 0) + DV(0, 1, 0, 0) + DV(0, 1, 0, 1) + DV(0, 0, 0, 1)
                                                - DV(0, 0, 1, 0) - DV(0, 1, 1, 0) - DV(0, 1, 1, 1) - DV(0, 0, 1, 1)) * dz * dy) / 4.0;
                    const double ey = (double)((DV(1, 0, 0, 0) + DV(1, 0, 1, 0) + DV(1, 0, 1, 1) + DV(1, 0, 0, 1)
                                                - DV(1, 1, 0, 0) - DV(1, 1, 1, 0) - DV(1, 1, 1, 1) - DV(1, 1, 0, 1)) * dx * dz) / 4.0;
                    const double ez = (double)((DV(2, 0, 0, 0) + DV(2, 1, 0, 0) + DV(2, 1, 1, 0) + DV(2, 0, 1, 0)
                                                - DV(2, 0, 0, 1) - DV(2, 1, 0, 1) - DV(2, 1, 1, 1) - DV(2, 0, 1, 1)) * dx * dy) / 4.0;
#undef DV
                    err += fabs(ex + ey + ez);
                    cnt += 1.0;
                }
#pragma unroll
                for (int f = 0; f < 3; f++)
#pragma unroll
                    for (int q = 0; q < 4; q++) pv[f][q >> 1][q & 1] = cv[f][q >> 1][q & 1];
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) { err += __shfl_down(err, off, 64); cnt += __shfl_down(cnt, off, 64); }
    __shared__ double se[4], sc[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { se[w] = err; sc[w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = (se[0] + se[1]) + (se[2] + se[3]);
        partial[2 * blockIdx.x + 1] = (sc[0] + sc[1]) + (sc[2] + sc[3]);
    }
}

__global__ void __launch_bounds__(256) k_div_final(const double *partial, int nblocks, double *out)
{
    __shared__ double se[256], sc[256];
    double e = 0.0, c = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) { e += partial[2 * b]; c += partial[2 * b + 1]; }
    se[threadIdx.x] = e; sc[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { se[threadIdx.x] += se[threadIdx.x + s]; sc[threadIdx.x] += sc[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = se[0]; out[1] = sc[0]; }
}

// ---------------------------------------------------------------------------------
// context helpers
// ---------------------------------------------------------------------------------

template <typename R> static R *fld(fs3d_ctx *c, int buf, int v) { return (R *)c->lay[buf] + (long long)v * c->fstride + c->plane; }
// byte pointer to the first owned cell of field v of layer buffer buf
static char *fptr(fs3d_ctx *c, int buf, int v) { return (char *)c->lay[buf] + ((size_t)v * c->fstride + c->plane) * c->esize; }

static inline unsigned grid_for(long long n, int bs, int cap = 4096)
{
    long long g = (n + bs - 1) / bs;
    return (unsigned)std::max(1LL, std::min<long long>(g, cap));
}

// per-launch HIP-event timing on the context's stream (events are pooled and reused)
static void rec_begin(fs3d_ctx *c, int cls)
{
    if (!c->timing) return;
    if (c->ev_used + 2 > c->ev.size()) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        c->ev.push_back(e0); c->ev.push_back(e1);
    }
    hipEventRecord(c->ev[c->ev_used], c->stream);
    c->ev_used += 2;
    c->ev_class.push_back(cls);
}
static void rec_end(fs3d_ctx *c)
{
    if (!c->timing) return;
    hipEventRecord(c->ev[c->ev_used - 1], c->stream);
}
// accumulates into t_ms/t_n (reset with fs3d_enable_timing)
static void rec_collect(fs3d_ctx *c)
{
    for (size_t i = 0; i < c->ev_class.size(); i++) {
        float ms = 0;
        hipEventSynchronize(c->ev[2 * i + 1]);
        hipEventElapsedTime(&ms, c->ev[2 * i], c->ev[2 * i + 1]);
        c->t_ms[c->ev_class[i]] += ms; c->t_n[c->ev_class[i]]++;
    }
    c->ev_used = 0; c->ev_class.clear();
}

// Device-side failures that cannot raise a HIP error (a relay hand-over of the pipe kernel that never arrived): the
// kernels set a bit in the context's pinned error word; every entry point that synchronises the stream checks it, so
// that wrong numbers are never returned with FS3D_OK (GPUplan.cpp:173-193: the reference throws on every device error).
static fs3d_status check_device_errors(fs3d_ctx *c)
{
    if (!c->errw_host || *c->errw_host == 0) return FS3D_OK;
    *c->errw_host = 0;
    return fail(c, FS3D_ERR_HIP, "GPU " + std::to_string(c->device) + ": sweep kernel: a relay hand-over between waves timed out; the fields of this step are invalid");
}

// ---------------------------------------------------------------------------------
// lifetime
// ---------------------------------------------------------------------------------

extern "C" const char *fs3d_version(void) { return FS3D_VERSION; }

extern "C" const char *fs3d_last_error(const fs3d_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" fs3d_status fs3d_create(fs3d_ctx **out, int device, fs3d_precision prec, int dimx, int dimy, int dimz,
                                   double dx, double dy, double dz, int x_offset, int dimx_global)
{
    if (!out) return fail(nullptr, FS3D_ERR_INVALID, "fs3d_create: out is NULL");
    *out = nullptr;
    if (dimx < 1 || dimy < 3 || dimz < 3 || dimx_global < 3 || x_offset < 0 || x_offset + dimx > dimx_global)
        return fail(nullptr, FS3D_ERR_INVALID, "fs3d_create: bad dimensions");
    if (prec != FS3D_F32 && prec != FS3D_F64) return fail(nullptr, FS3D_ERR_INVALID, "fs3d_create: bad precision");
    if ((long long)(dimx + 2) * dimy * dimz >= (1LL << 31))
        return fail(nullptr, FS3D_ERR_UNSUPPORTED, "fs3d_create: slab has more than 2^31 cells");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(nullptr, FS3D_ERR_HIP, "fs3d_create: no HIP device available");
    if (device < 0 || device >= ndev) return fail(nullptr, FS3D_ERR_INVALID, "fs3d_create: bad device ordinal");
    fs3d_ctx *c = new fs3d_ctx();
    c->device = device; c->prec = prec;
    c->dimx = dimx; c->dimy = dimy; c->dimz = dimz; c->x_offset = x_offset; c->dimx_global = dimx_global;
    c->gdx = dx; c->gdy = dy; c->gdz = dz;
    c->esize = prec == FS3D_F32 ? 4 : 8;
    c->plane = (long long)dimy * dimz; c->ncell = c->plane * dimx;
    if (const char *e = getenv("FS3D_TEST_DROP_HANDOFF")) c->test_drop = atoi(e) ? 1 : 0;
    if (const char *e = getenv("FS3D_DEFAULT_KERNEL")) {       // initial FS3D_OPT_SWEEP_KERNEL of new contexts (tests: 4 = bit-exact kernels only)
        const int v = atoi(e);
        if (v >= FS3D_SWEEP_AUTO && v <= FS3D_SWEEP_EXACT) c->opt_kernel = v;
    }
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { char b_[256]; snprintf(b_, sizeof b_, "GPU %d: %s failed: %s", device, #call, hipGetErrorString(e_)); g_create_err = b_; fs3d_destroy(c); return FS3D_ERR_HIP; } } while (0)
    CK(hipSetDevice(device));
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    // every field = haloSize + dimx*dimy*dimz + haloSize elements, data at +haloSize
    // (TimeLayer3D.h:354, GPUplan.h:79-108), zero-initialised
    // A sweep streams 16 arrays at once (4 fields of cur, temp, next, temp_out) at the same cell offset.  With power-of-two
    // grids the field stride and the allocation granularity put all 16 on the same low address bits (the same memory channel):
    // fields are padded and layers skewed against each other (FS3D_FIELD_PAD elements, FS3D_LAYER_SKEW bytes: experiments).
    // Measured (256^3 fp32, tools/pad_sweep.py, profiles/r2_padding.txt): 2338 -> 2770 Mcells/s; any pad of 576 .. 2112 elements does it.
    long long fpad = 576;                                      // 9 x 256 bytes in fp32
    size_t skew = 4864;                                        // 19 x 256 bytes per layer
    if (const char *e = getenv("FS3D_FIELD_PAD")) fpad = atoll(e) / 4 * 4;
    if (const char *e = getenv("FS3D_LAYER_SKEW")) skew = (size_t)atoll(e) / 256 * 256;
    if (fpad < 0) fpad = 0;
    c->fstride = c->ncell + 2 * c->plane + fpad;
    const size_t lbytes = (size_t)4 * c->fstride * c->esize;
    for (int l = 0; l < 5; l++) {
        CK(hipMalloc(&c->lay_raw[l], lbytes + 5 * skew));
        c->lay[l] = (char *)c->lay_raw[l] + (size_t)l * skew;
        CK(hipMemsetAsync(c->lay[l], 0, lbytes, c->stream));
    }
    // the 4 node-value fields and the 6 row fields of the exact kernels' scratch: padded the same way
    c->nstride = c->ncell + (getenv("FS3D_FIELD_PAD") ? fpad : 832);
    CK(hipMalloc((void **)&c->code, (size_t)c->nstride * sizeof(uint16_t)));
    CK(hipMemsetAsync(c->code, 0, (size_t)c->nstride * sizeof(uint16_t), c->stream));
    CK(hipMalloc(&c->node, (size_t)4 * c->nstride * c->esize));
    CK(hipMemsetAsync(c->node, 0, (size_t)4 * c->nstride * c->esize, c->stream));
    c->red_blocks = dimx * ((dimy + DIVE_JS - 1) / DIVE_JS);      // k_div_error: one block per plane and DIVE_JS rows
    CK(hipMalloc((void **)&c->red_buf, sizeof(double) * 2 * (c->red_blocks + 1)));
    CK(hipHostMalloc((void **)&c->red_host, sizeof(double) * 2, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&c->errw_host, sizeof(int), hipHostMallocMapped));
    *c->errw_host = 0;
    CK(hipHostGetDevicePointer((void **)&c->errw_dev, c->errw_host, 0));
    CK(hipStreamSynchronize(c->stream));
#undef CK
    *out = c;
    return FS3D_OK;
}

extern "C" void fs3d_destroy(fs3d_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    fs3d_comm_destroy(c);
    for (int l = 0; l < 5; l++) if (c->lay_raw[l]) hipFree(c->lay_raw[l]);
    if (c->redo) hipFree(c->redo);
    for (int i = 0; i < 2; i++) if (c->seg_carry[i]) hipFree(c->seg_carry[i]);
    if (c->code) hipFree(c->code);
    for (int d = 0; d < 3; d++) if (c->dead[d]) hipFree(c->dead[d]);
    for (int d = 0; d < 2; d++) { if (c->ucol[d]) hipFree(c->ucol[d]); if (c->uflag[d]) hipFree(c->uflag[d]); }
    if (c->node) hipFree(c->node);
    if (c->scr) hipFree(c->scr);
    if (c->xif_send) hipFree(c->xif_send);
    if (c->xif_all) hipFree(c->xif_all);
    for (int v = 0; v < 4; v++) if (c->bnd_val[v]) hipFree(c->bnd_val[v]);
    if (c->bnd_idx) hipFree(c->bnd_idx);
    if (c->red_buf) hipFree(c->red_buf);
    if (c->stamps) hipFree(c->stamps);
    if (c->red_host) hipHostFree(c->red_host);
    if (c->errw_host) hipHostFree(c->errw_host);
    for (auto e : c->ev) hipEventDestroy(e);
    if (c->ev_src) hipEventDestroy(c->ev_src);
    if (c->ev_halo) hipEventDestroy(c->ev_halo);
    if (c->comm_stream) hipStreamDestroy(c->comm_stream);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" fs3d_status fs3d_set_params(fs3d_ctx *c, double v_T, double v_vis, double t_vis, double t_phi)
{
    if (!c) return FS3D_ERR_INVALID;
    c->v_T = v_T; c->v_vis = v_vis; c->t_vis = t_vis; c->t_phi = t_phi; c->have_params = true;
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_set_option(fs3d_ctx *c, int option, int value)
{
    if (!c) return FS3D_ERR_INVALID;
    switch (option) {
    case FS3D_OPT_SWEEP_KERNEL:
        if (value < FS3D_SWEEP_AUTO || value > FS3D_SWEEP_EXACT) return fail(c, FS3D_ERR_INVALID, "bad sweep kernel id");
        c->opt_kernel = value; return FS3D_OK;
    case FS3D_OPT_FUSE_MERGE: c->opt_fuse = value ? 1 : 0; return FS3D_OK;
    case FS3D_OPT_DIV_CORE: c->opt_div_core = value ? 1 : 0; return FS3D_OK;
    case FS3D_OPT_OVERLAP: c->opt_overlap = value ? 1 : 0; return FS3D_OK;
    case FS3D_OPT_KEEP_TEMP: c->opt_keep_temp = value ? 1 : 0; return FS3D_OK;
    case FS3D_OPT_XSOLVE:
        if (value < 0 || value > 3) return fail(c, FS3D_ERR_INVALID, "bad cross-slab X solve id");
        c->opt_xsolve = value; return FS3D_OK;
    default: return fail(c, FS3D_ERR_INVALID, "unknown option");
    }
}

extern "C" fs3d_status fs3d_enable_timing(fs3d_ctx *c, int on)
{
    if (!c) return FS3D_ERR_INVALID;
    c->timing = on != 0;
    c->timing_period = on > 0 ? on : 0; c->timing_steps = 0;
    for (int k = 0; k < 8; k++) { c->t_ms[k] = 0; c->t_n[k] = 0; }
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_last_step_timing(fs3d_ctx *c, float ms[4], int n[4])
{
    if (!c) return FS3D_ERR_INVALID;
    for (int k = 0; k < 3; k++) { if (ms) ms[k] = c->t_ms[k]; if (n) n[k] = c->t_n[k]; }
    float rest = 0; int nrest = 0;
    for (int k = 3; k < 8; k++) { rest += c->t_ms[k]; nrest += c->t_n[k]; }
    if (ms) ms[3] = rest;
    if (n) n[3] = nrest;
    return FS3D_OK;
}

static const char *const k_event_names[FS3D_N_EVENTS] = {"SolveSegments_Z", "SolveSegments_Y", "SolveSegments_X", "CopyLayer", "MergeLayer",
                                                         "EvalDivError", "UpdateBoundaries", "syncHalos", "CreateSegments"};
extern "C" fs3d_status fs3d_profiler_events(fs3d_ctx *c, const char *names[FS3D_N_EVENTS], float ms[FS3D_N_EVENTS], int n[FS3D_N_EVENTS])
{
    if (!c) return FS3D_ERR_INVALID;
    for (int k = 0; k < 8; k++) { if (names) names[k] = k_event_names[k]; if (ms) ms[k] = c->t_ms[k]; if (n) n[k] = c->t_n[k]; }
    if (names) names[8] = k_event_names[8];
    if (ms) ms[8] = (float)c->t_create_segments_ms;
    if (n) n[8] = c->have_nodes ? 1 : 0;
    return FS3D_OK;
}

// ---------------------------------------------------------------------------------
// geometry: row codes from the node-type array
// ---------------------------------------------------------------------------------

// Per-line restatement of Grid3D::GenerateListSegments (Grid3D.cpp:47-127, nblockZ = 1):
// walk the line; a run of NODE_IN cells opened at pos+1 takes the cell at pos as its
// first node and the first non-IN cell after it as its last node; a run that reaches the
// end of the line without a closing cell is dropped.  kinds[] gets START/INTERIOR/END.
// Returns the number of segments; *shared_free is set when a cell closes one segment and
// opens the next while carrying a FREE boundary condition (two different rows on one cell).
static int line_kinds(const uint8_t *type, long long base, long long stride, int n, uint8_t *kinds,
                      const uint8_t *bc_vel, const uint8_t *bc_temp, bool *shared_free)
{
    int nseg = 0, state = 0, start = 0;
    for (int s = 0; s < n; s++) kinds[s] = ROW_SKIP;
    for (int pos = 0; pos + 1 < n; pos++) {
        if (type[base + (long long)(pos + 1) * stride] == FS3D_NODE_IN) {
            if (state == 0) start = pos;
            state = 1;
        } else if (state == 1) {
            const int end = pos + 1;
            if (kinds[start] == ROW_END) {   // closes the previous segment and opens this one
                const long long id = base + (long long)start * stride;
                if (bc_vel[id] == FS3D_BC_FREE || bc_temp[id] == FS3D_BC_FREE) *shared_free = true;
            }
            kinds[start] = ROW_START;
            for (int s = start + 1; s < end; s++) kinds[s] = ROW_INTERIOR;
            kinds[end] = ROW_END;
            nseg++;
            state = 0;
        }
    }
    return nseg;
}

template <typename R>
static fs3d_status upload_nodes_impl(fs3d_ctx *c, const uint8_t *type, const uint8_t *bc_vel, const uint8_t *bc_temp,
                                     const R *vx, const R *vy, const R *vz, const R *T, int n_seg_out[3])
{
    const int gx = c->dimx_global, dy = c->dimy, dz = c->dimz, x0 = c->x_offset, nx = c->dimx;
    const long long plane = c->plane;
    std::vector<uint16_t> code((size_t)c->ncell, 0);
    std::vector<uint8_t> kinds((size_t)std::max(gx, std::max(dy, dz)));
    bool shared_free = false;
    long long nseg[3] = {0, 0, 0};
    auto put = [&](int dir, long long gid, int kind) {
        const int gi = (int)(gid / plane);
        if (gi < x0 || gi >= x0 + nx) return;
        int rc = kind;
        if (kind == ROW_START || kind == ROW_END) {
            if (bc_vel[gid] == FS3D_BC_FREE) rc |= ROW_VELFREE;
            if (bc_temp[gid] == FS3D_BC_FREE) rc |= ROW_TEMPFREE;
        }
        code[(size_t)(gid - (long long)x0 * plane)] |= (uint16_t)(rc << (4 * dir));
    };
    // X lines span all slabs: kinds come from the global line (as the reference builds
    // global segments and clips them per device, AdiSolver3D.cpp:475-524)
    for (int j = 0; j < dy; j++)
        for (int k = 0; k < dz; k++) {
            const long long base = (long long)j * dz + k;
            nseg[0] += line_kinds(type, base, plane, gx, kinds.data(), bc_vel, bc_temp, &shared_free);
            for (int s = x0; s < x0 + nx; s++) put(0, base + (long long)s * plane, kinds[s]);
        }
    for (int i = x0; i < x0 + nx; i++)
        for (int k = 0; k < dz; k++) {
            const long long base = (long long)i * plane + k;
            nseg[1] += line_kinds(type, base, dz, dy, kinds.data(), bc_vel, bc_temp, &shared_free);
            for (int s = 0; s < dy; s++) put(1, base + (long long)s * dz, kinds[s]);
        }
    for (int i = x0; i < x0 + nx; i++)
        for (int j = 0; j < dy; j++) {
            const long long base = (long long)i * plane + (long long)j * dz;
            nseg[2] += line_kinds(type, base, 1, dz, kinds.data(), bc_vel, bc_temp, &shared_free);
            for (int s = 0; s < dz; s++) put(2, base + s, kinds[s]);
        }
    if (shared_free)
        return fail(c, FS3D_ERR_UNSUPPORTED,
                    "fs3d_upload_nodes: a cell with a FREE boundary condition closes one segment and opens the next "
                    "on the same line (two rows on one cell; the reference's result there depends on thread timing)");
    std::vector<int> bidx;
    std::vector<R> bval[4];
    std::vector<R> nv[4];
    for (int v = 0; v < 4; v++) nv[v].resize((size_t)c->ncell);
    for (long long l = 0; l < c->ncell; l++) {
        const long long g = l + (long long)x0 * plane;
        code[(size_t)l] |= (uint16_t)((type[g] & 3) << CODE_TYPE_SHIFT);
        nv[0][l] = vx[g]; nv[1][l] = vy[g]; nv[2][l] = vz[g]; nv[3][l] = T[g];
        if (type[g] == FS3D_NODE_BOUND || type[g] == FS3D_NODE_VALVE) {
            bidx.push_back((int)l);
            bval[0].push_back(vx[g]); bval[1].push_back(vy[g]); bval[2].push_back(vz[g]); bval[3].push_back(T[g]);
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(c->code, code.data(), code.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    {
        // dead lines: no cell of the (local part of the) line is on a segment of that direction or NODE_IN -- nothing a
        // sweep computes for such a line is ever stored; the partition kernels keep them off the row-kind paths
        const long long nl[3] = {(long long)dy * dz, (long long)nx * dz, (long long)nx * dy};
        // NODE_IN cells that lie on no segment of some direction (a run without a closing cell, Grid3D.cpp:87-117): the reference merges
        // the STALE `next` value there -- whatever an earlier sweep left.  Only a geometry without such cells lets the time step drop
        // stores of `next` that nothing but they could read (time_step_enqueue).
        long long stale = 0;
        for (long long l = 0; l < c->ncell; l++)
            if (((code[(size_t)l] >> CODE_TYPE_SHIFT) & 3) == FS3D_NODE_IN)
                for (int d = 0; d < 3; d++) stale += ((code[(size_t)l] >> (4 * d)) & 3) == ROW_SKIP;
        c->stale_in_cells = stale;
        for (int d = 0; d < 3; d++) {
            std::vector<uint8_t> dead((size_t)nl[d], 1);
            for (long long l = 0; l < c->ncell; l++) {
                const int i = (int)(l / plane), rem = (int)(l - (long long)i * plane), j = rem / dz, k = rem - j * dz;
                const bool live = ((code[(size_t)l] >> (4 * d)) & 3) != ROW_SKIP || ((code[(size_t)l] >> CODE_TYPE_SHIFT) & 3) == FS3D_NODE_IN;
                if (live) dead[(size_t)(d == 0 ? (long long)j * dz + k : (d == 1 ? (long long)i * dz + k : (long long)i * dy + j))] = 0;
            }
            if (c->dead[d]) { hipFree(c->dead[d]); c->dead[d] = nullptr; }
            HIPCHK(c, hipMalloc((void **)&c->dead[d], (size_t)nl[d]));
            HIPCHK(c, hipMemcpy(c->dead[d], dead.data(), (size_t)nl[d], hipMemcpyHostToDevice));
            if (d < 2) {
                // shared code columns (X: o = j, line cells along i; Y: o = i, cells along j; lanes k in groups of 32): a group is
                // uniform when all its live lines carry the same (row code of this direction, node type) on every cell
                if (c->ucol[d]) { hipFree(c->ucol[d]); c->ucol[d] = nullptr; }
                if (c->uflag[d]) { hipFree(c->uflag[d]); c->uflag[d] = nullptr; }
                const int n_o = d == 0 ? dy : nx, n = d == 0 ? nx : dy, ng = (dz + 31) / 32;
                if (n <= UCOL_PITCH) {
                    const uint16_t keep = (uint16_t)((0xF << (4 * d)) | (3 << CODE_TYPE_SHIFT));
                    const long long ss = d == 0 ? plane : dz, os = d == 0 ? (long long)dz : plane;
                    std::vector<uint16_t> col((size_t)n_o * ng * UCOL_PITCH, 0);
                    std::vector<uint8_t> flag((size_t)n_o * ng, 0);
                    for (int o = 0; o < n_o; o++)
                        for (int g = 0; g < ng; g++) {
                            uint16_t *cc = &col[((size_t)o * ng + g) * UCOL_PITCH];
                            int k0 = -1;
                            bool uni = true;
                            for (int k = 32 * g; k < std::min(32 * g + 32, dz) && uni; k++) {
                                if (dead[(size_t)o * dz + k]) continue;
                                const uint16_t *src = &code[(size_t)((long long)o * os + k)];
                                if (k0 < 0) { k0 = k; for (int s2 = 0; s2 < n; s2++) cc[s2] = (uint16_t)(src[(size_t)s2 * ss] & keep); }
                                else for (int s2 = 0; s2 < n; s2++) if ((uint16_t)(src[(size_t)s2 * ss] & keep) != cc[s2]) { uni = false; break; }
                            }
                            flag[(size_t)o * ng + g] = uni ? 1 : 0;
                        }
                    for (int o = 0; o < n_o; o++)
                        for (int g = 0; g < ng; g += 2) {
                            bool pair = flag[(size_t)o * ng + g] & 1;
                            if (pair && g + 1 < ng) {
                                pair = flag[(size_t)o * ng + g + 1] & 1;
                                // an all-dead group holds zeros: it takes the other group's column
                                const uint16_t *a = &col[((size_t)o * ng + g) * UCOL_PITCH], *b = a + UCOL_PITCH;
                                bool da = true, db = true;
                                for (int k = 32 * g; k < std::min(32 * g + 32, dz); k++) da = da && dead[(size_t)o * dz + k];
                                for (int k = 32 * g + 32; k < std::min(32 * g + 64, dz); k++) db = db && dead[(size_t)o * dz + k];
                                if (pair && da && !db) std::copy(b, b + UCOL_PITCH, &col[((size_t)o * ng + g) * UCOL_PITCH]);
                                else if (pair && !da && !db) pair = std::equal(a, a + n, b);
                            }
                            if (pair) flag[(size_t)o * ng + g] |= 2;
                        }
                    // the distinct columns only (a box has three: interior lines, and the rows / planes at the faces): they stay in the caches
                    std::map<std::string, unsigned> ids;
                    std::vector<uint16_t> uniq;
                    std::vector<unsigned> fl(flag.size(), 0);
                    for (size_t q = 0; q < flag.size(); q++) {
                        if (!flag[q]) continue;
                        const uint16_t *cc = &col[q * UCOL_PITCH];
                        const std::string key((const char *)cc, (size_t)n * sizeof(uint16_t));
                        auto it = ids.find(key);
                        if (it == ids.end()) { it = ids.emplace(key, (unsigned)(uniq.size() / UCOL_PITCH)).first; uniq.insert(uniq.end(), cc, cc + UCOL_PITCH); }
                        fl[q] = (unsigned)flag[q] | (it->second << 2);
                    }
                    if (uniq.empty()) uniq.resize(UCOL_PITCH, 0);
                    HIPCHK(c, hipMalloc((void **)&c->ucol[d], uniq.size() * sizeof(uint16_t)));
                    HIPCHK(c, hipMemcpy(c->ucol[d], uniq.data(), uniq.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
                    HIPCHK(c, hipMalloc((void **)&c->uflag[d], fl.size() * sizeof(unsigned)));
                    HIPCHK(c, hipMemcpy(c->uflag[d], fl.data(), fl.size() * sizeof(unsigned), hipMemcpyHostToDevice));
                }
            }
        }
    }
    for (int v = 0; v < 4; v++)
        HIPCHK(c, hipMemcpy((R *)c->node + (size_t)v * c->nstride, nv[v].data(), (size_t)c->ncell * sizeof(R), hipMemcpyHostToDevice));
    if (c->bnd_idx) { hipFree(c->bnd_idx); c->bnd_idx = nullptr; }
    for (int v = 0; v < 4; v++) if (c->bnd_val[v]) { hipFree(c->bnd_val[v]); c->bnd_val[v] = nullptr; }
    c->n_bnd = (int)bidx.size();
    if (c->n_bnd) {
        HIPCHK(c, hipMalloc((void **)&c->bnd_idx, sizeof(int) * bidx.size()));
        HIPCHK(c, hipMemcpy(c->bnd_idx, bidx.data(), sizeof(int) * bidx.size(), hipMemcpyHostToDevice));
        for (int v = 0; v < 4; v++) {
            HIPCHK(c, hipMalloc(&c->bnd_val[v], sizeof(R) * bidx.size()));
            HIPCHK(c, hipMemcpy(c->bnd_val[v], bval[v].data(), sizeof(R) * bidx.size(), hipMemcpyHostToDevice));
        }
    }
    for (int d = 0; d < 3; d++) { c->nseg[d] = (int)nseg[d]; if (n_seg_out) n_seg_out[d] = (int)nseg[d]; }
    c->have_nodes = true;
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_upload_nodes(fs3d_ctx *c, const uint8_t *type, const uint8_t *bc_vel, const uint8_t *bc_temp,
                                         const void *vx, const void *vy, const void *vz, const void *T, int n_seg_out[3])
{
    if (!c) return FS3D_ERR_INVALID;
    if (!type || !bc_vel || !bc_temp || !vx || !vy || !vz || !T) return fail(c, FS3D_ERR_INVALID, "fs3d_upload_nodes: NULL array");
    const auto t0 = std::chrono::steady_clock::now();
    const fs3d_status st = c->prec == FS3D_F32
        ? upload_nodes_impl<float>(c, type, bc_vel, bc_temp, (const float *)vx, (const float *)vy, (const float *)vz, (const float *)T, n_seg_out)
        : upload_nodes_impl<double>(c, type, bc_vel, bc_temp, (const double *)vx, (const double *)vy, (const double *)vz, (const double *)T, n_seg_out);
    c->t_create_segments_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st;
}

// ---------------------------------------------------------------------------------
// layers
// ---------------------------------------------------------------------------------

static fs3d_status check_layer(fs3d_ctx *c, int layer)
{
    if (layer < 0 || layer > 3) return fail(c, FS3D_ERR_INVALID, "bad layer id");
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_init_layers_from_nodes(fs3d_ctx *c)
{
    if (!c) return FS3D_ERR_INVALID;
    if (!c->have_nodes) return fail(c, FS3D_ERR_INVALID, "fs3d_init_layers_from_nodes: upload nodes first");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t lbytes = (size_t)4 * c->fstride * c->esize;
    for (int l = 0; l < 5; l++) HIPCHK(c, hipMemsetAsync(c->lay[l], 0, lbytes, c->stream));
    for (int l = 0; l < 4; l++) c->slot[l] = l;
    c->spare = 4;
    for (int v = 0; v < 4; v++)
        HIPCHK(c, hipMemcpyAsync(fptr(c, c->slot[FS3D_LAYER_CUR], v), (char *)c->node + (size_t)v * c->nstride * c->esize,
                                 (size_t)c->ncell * c->esize, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_upload_layer(fs3d_ctx *c, int layer, const void *u, const void *v, const void *w, const void *T)
{
    if (!c) return FS3D_ERR_INVALID;
    if (check_layer(c, layer)) return FS3D_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const void *src[4] = {u, v, w, T};
    for (int k = 0; k < 4; k++)
        if (src[k])
            HIPCHK(c, hipMemcpyAsync(fptr(c, c->slot[layer], k), src[k],
                                     (size_t)c->ncell * c->esize, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_download_layer(fs3d_ctx *c, int layer, void *u, void *v, void *w, void *T)
{
    if (!c) return FS3D_ERR_INVALID;
    if (check_layer(c, layer)) return FS3D_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    void *dst[4] = {u, v, w, T};
    for (int k = 0; k < 4; k++)
        if (dst[k])
            HIPCHK(c, hipMemcpyAsync(dst[k], fptr(c, c->slot[layer], k),
                                     (size_t)c->ncell * c->esize, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_field_dev_ptr(fs3d_ctx *c, int layer, int var, void **dev_ptr)
{
    if (!c || !dev_ptr) return FS3D_ERR_INVALID;
    if (check_layer(c, layer) || var < 0 || var > 3) return fail(c, FS3D_ERR_INVALID, "bad layer/var id");
    *dev_ptr = fptr(c, c->slot[layer], var);
    return FS3D_OK;
}

// ---------------------------------------------------------------------------------
// hot path
// ---------------------------------------------------------------------------------

template <typename R>
static void fill_params(fs3d_ctx *c, SweepParams<R> &p, int dir, double dt_, int b_cur, int b_temp, int b_next, int b_tout, int merge)
{
    p.dimx = c->dimx; p.dimy = c->dimy; p.dimz = c->dimz; p.plane = c->plane;
    p.cur_ = fld<R>(c, b_cur, 0); p.temp_ = fld<R>(c, b_temp, 0);
    p.next_ = fld<R>(c, b_next, 0); p.temp_out_ = fld<R>(c, b_tout, 0);
    p.fstride = c->fstride;
    p.node_ = (const R *)c->node; p.scr_ = (R *)c->scr; p.nstride = c->nstride;
    p.code = c->code;
    p.dead = c->dead[dir];
    p.ucol = dir < 2 ? c->ucol[dir] : nullptr; p.uflag = dir < 2 ? c->uflag[dir] : nullptr; p.ung = (c->dimz + 31) / 32;
    // every constant below is evaluated in FTYPE exactly as the reference writes it
    const R dx = (R)c->gdx, dy = (R)c->gdy, dz = (R)c->gdz;          // TimeLayer3D.h:1078-1080
    const R ds = dir == 0 ? dx : (dir == 1 ? dy : dz);
    const R dt = (R)dt_;                                             // FluidSolver3D.cpp:242 (FTYPE)dt
    const R v_vis = (R)c->v_vis, t_vis = (R)c->t_vis;
    p.two_ds[0] = 2 * dx; p.two_ds[1] = 2 * dy; p.two_ds[2] = 2 * dz;   // TimeLayer3D.h:338-340
    p.vis_v = v_vis / (ds * ds);                                     // AdiSolver3D.cpp:744-746
    p.vis_t = t_vis / (ds * ds);                                     // AdiSolver3D.cpp:749-751
    p.b_v = 3 / dt + 2 * p.vis_v;                                    // AdiSolver3D.cpp:761
    p.b_t = 3 / dt + 2 * p.vis_t;
    p.dt = dt; p.v_T = (R)c->v_T; p.t_phi = (R)c->t_phi;
    p.merge = merge & 3;
    p.store_next = (merge & 4) ? 0 : 1;      // merge | 4: the caller never reads this sweep's `next` (time_step_enqueue)
    p.stamps = nullptr;
    p.errw = c->errw_dev;
    p.o_begin = 0; p.o_count = 0;
    p.test_drop = c->test_drop;
    p.xiface_pass = 0;
    p.carry_in = nullptr; p.carry_out = nullptr; p.xcarry_in = nullptr; p.xcarry_out = nullptr; p.bundle0 = 0;
    p.seg_begin = 0; p.seg_len = 0; p.carry_pitch = c->plane; p.seg_index = 0; p.scr_bundles = 0;
    p.ghost_lo = c->x_offset > 0; p.ghost_hi = c->x_offset + c->dimx < c->dimx_global;
    // division core (fp32 pipe kernel): the constant divisors must be plain numbers in [2^-30, 2^60)
    // [2^-30, 2^26): with a divisor beyond 2^26 a tiny numerator gives a denormal quotient, which the core would double-round
    auto plain = [](double v) { v = v < 0 ? -v : v; return v >= 9.313225746154785e-10 && v < 67108864.0; };
    p.fast_div = c->opt_div_core && plain((double)p.two_ds[0]) && plain((double)p.two_ds[1]) && plain((double)p.two_ds[2]) && plain((double)p.dt);
}

static fs3d_status ensure_scratch(fs3d_ctx *c)
{
    if (c->scr) return FS3D_OK;
    HIPCHK(c, hipMalloc(&c->scr, (size_t)6 * c->nstride * c->esize));
    c->scr_bytes = (size_t)6 * c->nstride * c->esize;
    return FS3D_OK;
}

// carry buffers of the cross-slab X sweeps ([value][line]: 6 forward, 4 backward words per line); failure is a status, not an
// early return past the callers' abort guard
static fs3d_status ensure_carries(fs3d_ctx *c)
{
    if (c->carry[0] && c->carry[1] && c->carry[2] && c->carry[3]) return FS3D_OK;
    const size_t pl = (size_t)c->plane;
    const size_t words[4] = {6, 6, 4, 4};
    for (int k = 0; k < 4; k++)
        if (!c->carry[k] && hipMalloc(&c->carry[k], words[k] * pl * c->esize) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, FS3D_ERR_HIP, "cross-slab X sweep: hipMalloc of the carry buffers failed");
        }
    return FS3D_OK;
}

// Cross-slab X sweep (nranks > 1): forward over the slabs 0 -> R-1, backward R-1 -> 0, carries over RCCL.
// The lines of the plane are cut into `xblocks` blocks that travel through the ranks as a pipeline (rank r
// works on block b while rank r+1 works on block b-1) -- the reference's `blocking` idea (AdiSolver3D.cu:642-881).
// Carries are [value][line]; a block moves as one grouped transfer of its 6 (forward) / 4 (backward) row pieces.
template <typename R>
static fs3d_status xsweep_multi(fs3d_ctx *c, SweepParams<R> &p)
{
    // a rank that fails anywhere in here -- allocations included -- must not leave its neighbours blocked in their receives: it
    // aborts the group (fs3d_comm_abort), the peers' pending and later exchanges return FS3D_ERR_COMM
    fs3d_status st = FS3D_OK;
    struct AbortOnError { fs3d_ctx *c; fs3d_status *st; ~AbortOnError() { if (*st != FS3D_OK) fs3d_comm_abort(c); } } guard{c, &st};
    if ((st = ensure_scratch(c))) return st;
    p.scr_ = (R *)c->scr;
    const size_t pl = (size_t)c->plane;
    if ((st = ensure_carries(c))) return st;
    const bool first = c->rank == 0, last = c->rank == c->nranks - 1;
    const int nb = c->xblocks < 1 ? 1 : c->xblocks;
    // per-slab halves: the pipe kernel (rows on chip, 64 lines per bundle) where the slab allows it, else thread-per-line
    const bool pipe = c->opt_kernel != FS3D_SWEEP_LINE && xslab_pipe_supported<R>(p);
    c->ran_kernel[0] = pipe ? FS3D_SWEEP_PIPE : FS3D_SWEEP_LINE; c->ran_segmented[0] = 1;
    if (c->opt_kernel == FS3D_SWEEP_PIPE && !pipe) return fail(c, FS3D_ERR_UNSUPPORTED, "pipe kernel: slab dims unsupported for the X sweep");
    auto range = [&](int b, long long &l0, long long &l1) {
        const long long per = ((long long)pl / 64 + nb - 1) / nb * 64;      // whole waves / bundles per block
        l0 = std::min<long long>((long long)b * per, (long long)pl); l1 = std::min<long long>(l0 + per, (long long)pl);
    };
    p.carry_in = first ? nullptr : (const R *)c->carry[0]; p.carry_out = (R *)c->carry[1];
    p.xcarry_in = last ? nullptr : (const R *)c->carry[2]; p.xcarry_out = (R *)c->carry[3];
    for (int b = 0; b < nb; b++) {
        long long l0, l1; range(b, l0, l1);
        if (l1 <= l0) continue;
        if (!first && (st = fs3d_comm_xfer_rows(c, c->carry[0], 6, pl, l0, l1, c->rank - 1, false))) return st;
        if (pipe) { if (!launch_xslab_pipe<R>(c, p, 1, (int)(l0 / 64), (int)(l1 / 64))) return st = fail(c, FS3D_ERR_HIP, "pipe kernel launch (forward half)"); }
        else launch_xsweep_fwd<R>(c, p, first ? nullptr : c->carry[0], c->carry[1], l0, l1);
        if (!last && (st = fs3d_comm_xfer_rows(c, c->carry[1], 6, pl, l0, l1, c->rank + 1, true))) return st;
    }
    for (int b = 0; b < nb; b++) {
        long long l0, l1; range(b, l0, l1);
        if (l1 <= l0) continue;
        if (!last && (st = fs3d_comm_xfer_rows(c, c->carry[2], 4, pl, l0, l1, c->rank + 1, false))) return st;
        if (pipe) { if (!launch_xslab_pipe<R>(c, p, 2, (int)(l0 / 64), (int)(l1 / 64))) return st = fail(c, FS3D_ERR_HIP, "pipe kernel launch (backward half)"); }
        else launch_xsweep_bwd<R>(c, p, last ? nullptr : c->carry[2], c->carry[3], l0, l1);
        if (!first && (st = fs3d_comm_xfer_rows(c, c->carry[3], 4, pl, l0, l1, c->rank - 1, true))) return st;
    }
    return FS3D_OK;
}

// Cross-slab X sweep, reduced-interface form (kernels_line.hip: k_xiface / k_xreduce): every rank eliminates its slab at
// once, ONE all-gather of 18 words per line, then the slab's halves run with the two boundary values given -- all lines
// in one launch each, no pipeline over the ranks.
template <typename R>
static fs3d_status xsweep_reduced(fs3d_ctx *c, SweepParams<R> &p)
{
    fs3d_status st = FS3D_OK;
    struct AbortOnError { fs3d_ctx *c; fs3d_status *st; ~AbortOnError() { if (*st != FS3D_OK) fs3d_comm_abort(c); } } guard{c, &st};   // peers never block on a rank that failed
    if (c->nranks > FS3D_XREDUCE_MAX_RANKS) return st = fail(c, FS3D_ERR_UNSUPPORTED, "reduced-interface X solve: more than 64 slabs (k_xreduce holds the slab system in registers); use FS3D_XSOLVE_PIPELINED");
    if ((st = ensure_scratch(c))) return st;
    p.scr_ = (R *)c->scr;
    const size_t pl = (size_t)c->plane;
    if ((st = ensure_carries(c))) return st;
    if (!c->xif_send) {
        if (hipMalloc(&c->xif_send, 18 * pl * c->esize) != hipSuccess || hipMalloc(&c->xif_all, (size_t)c->nranks * 18 * pl * c->esize) != hipSuccess) {
            (void)hipGetLastError();
            return st = fail(c, FS3D_ERR_HIP, "reduced-interface X solve: hipMalloc of the interface buffers failed");
        }
    }
    // the slab's interface words: a first pass of the X partition kernel (rows and chunk elimination on chip, 8 words per cell
    // read, 18 words per line written) where it applies, else the thread-per-line walk over the planes
    bool iface_done = false;
    if (c->opt_kernel == FS3D_SWEEP_AUTO || c->opt_kernel == FS3D_SWEEP_PART) {
        SweepParams<R> pa = p;
        pa.xiface_pass = 1; pa.carry_in = nullptr; pa.xcarry_in = nullptr; pa.carry_out = (R *)c->xif_send; pa.merge = 0; pa.store_next = 0;
        iface_done = launch_sweep_part<R>(c, 0, pa);
    }
    if (!iface_done) launch_xiface<R>(c, p, c->xif_send);
    // the R x R interface systems: every rank solves every line's (one all-gather), or -- three ranks and more -- every rank solves
    // the lines it owns and hands every rank its two boundary values (two all-to-alls of point-to-point transfers: (R-1)/R x 26
    // words per line on the wires instead of (R-1) x 18; same operations on the same values: bit-identical fields)
    const bool a2a = c->opt_xsolve == 3 || (c->opt_xsolve != 2 && c->nranks >= 3);
    c->ran_xa2a = a2a ? 1 : 0;
    if (a2a) {
        const long long lp = (((long long)pl + c->nranks - 1) / c->nranks + 63) / 64 * 64;       // lines per owner
        const size_t w1 = (size_t)c->nranks * 18 * lp * c->esize, w2 = (size_t)c->nranks * 8 * lp * c->esize;
        if (!c->xa2a[0]) {
            const size_t sz[4] = {w1, w1, w2, w2};
            for (int k = 0; k < 4; k++)
                if (hipMalloc(&c->xa2a[k], sz[k]) != hipSuccess || hipMemsetAsync(c->xa2a[k], 0, sz[k], c->stream) != hipSuccess) {
                    (void)hipGetLastError();
                    return st = fail(c, FS3D_ERR_HIP, "distributed interface solve: hipMalloc of the exchange buffers failed");
                }
        }
        launch_xpack<R>(c, c->xif_send, (long long)pl, lp, c->xa2a[0]);
        if ((st = fs3d_comm_alltoall(c, c->xa2a[0], c->xa2a[1], (size_t)18 * lp))) return st;
        launch_xreduce_a2a<R>(c, c->xa2a[1], (long long)pl, lp, c->nranks, c->rank, c->xa2a[2]);
        if ((st = fs3d_comm_alltoall(c, c->xa2a[2], c->xa2a[3], (size_t)8 * lp))) return st;
        launch_xunpack<R>(c, c->xa2a[3], (long long)pl, lp, c->carry[0], c->carry[2]);
    } else {
        if ((st = fs3d_comm_allgather(c, c->xif_send, c->xif_all, 18 * pl))) return st;
        launch_xreduce<R>(c, c->xif_all, (long long)pl, c->nranks, c->rank, c->carry[0], c->carry[2]);
    }
    p.carry_in = (const R *)c->carry[0]; p.carry_out = (R *)c->carry[1];
    p.xcarry_in = (const R *)c->carry[2]; p.xcarry_out = (R *)c->carry[3];
    c->ran_xsolve = iface_done ? 3 : 2;
    // the slab with both boundary values given: the X partition kernel (rows on chip, 16 words per cell) where it applies ...
    if ((c->opt_kernel == FS3D_SWEEP_AUTO || c->opt_kernel == FS3D_SWEEP_PART) && launch_sweep_part<R>(c, 0, p)) {
        c->ran_kernel[0] = FS3D_SWEEP_PART; c->ran_segmented[0] = 0;
        return FS3D_OK;
    }
    // ... else the exact halves (rows through the HBM scratch)
    const bool pipe = c->opt_kernel != FS3D_SWEEP_LINE && xslab_pipe_supported<R>(p);
    c->ran_kernel[0] = pipe ? FS3D_SWEEP_PIPE : FS3D_SWEEP_LINE; c->ran_segmented[0] = 1;
    if (pipe) {
        if (!launch_xslab_pipe<R>(c, p, 1, 0, (int)(pl / 64))) return st = fail(c, FS3D_ERR_HIP, "pipe kernel launch (forward half)");
        if (!launch_xslab_pipe<R>(c, p, 2, 0, (int)(pl / 64))) return st = fail(c, FS3D_ERR_HIP, "pipe kernel launch (backward half)");
    } else {
        launch_xsweep_fwd<R>(c, p, c->carry[0], c->carry[1], 0, (long long)pl);
        launch_xsweep_bwd<R>(c, p, c->carry[2], c->carry[3], 0, (long long)pl);
    }
    return FS3D_OK;
}

// Y / Z sweep of an x-slab with the halo planes of temp travelling BESIDE the sweep (north_star: "ghost-cell halo exchange ...
// overlapped with interior-cell compute on a second HIP stream"; the reference exchanges first, TimeLayer3D.h:272-335 /
// AdiSolver3D.cpp:608).  Only the first and the last owned plane read a ghost plane (the o+-1 neighbours of the stencils):
//   stream        : interior planes [1, nx-1)  ........ wait(ev_halo) -> plane 0, plane nx-1
//   comm_stream   : wait(ev_src) -> send / recv of the 4 x 2 halo planes -> ev_halo
// ev_src marks the point where the planes to be sent are final (everything enqueued before this sweep).  Same kernels, same
// cells, same values as the plain order -- only the launches are cut differently.  false: not applicable (caller exchanges
// first and sweeps in one launch).
template <typename R>
static bool sweep_overlapped(fs3d_ctx *c, int dir, SweepParams<R> &p, int b_temp, fs3d_status &st)
{
    st = FS3D_OK;
    if (!c->opt_overlap || c->nranks < 2 || dir == 0 || c->dimx < 3) return false;
    if (c->opt_kernel != FS3D_SWEEP_AUTO && c->opt_kernel != FS3D_SWEEP_PART) return false;
    if (std::is_same<R, double>::value) return false;
    if (!c->comm_stream) {
        if (hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_src, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming) != hipSuccess) { st = fail(c, FS3D_ERR_HIP, "halo overlap: stream / event creation failed"); return true; }
    }
    if (hipEventRecord(c->ev_src, c->stream) != hipSuccess) { st = fail(c, FS3D_ERR_HIP, "halo overlap: event"); return true; }
    p.o_begin = 1; p.o_count = c->dimx - 2;
    if (!launch_sweep_part<R>(c, dir, p)) { p.o_begin = 0; p.o_count = 0; return false; }      // dims outside the partition kernels
    c->ran_kernel[dir] = FS3D_SWEEP_PART; c->ran_segmented[dir] = 0;
    // the interior planes are running; now the exchange, on its own stream, behind everything that was enqueued before them
    if (hipStreamWaitEvent(c->comm_stream, c->ev_src, 0) != hipSuccess) { st = fail(c, FS3D_ERR_HIP, "halo overlap: event"); return true; }
    c->xstream = c->comm_stream;
    st = fs3d_comm_halo_exchange(c, b_temp, 4);
    c->xstream = nullptr;
    if (st) return true;
    if (hipEventRecord(c->ev_halo, c->comm_stream) != hipSuccess || hipStreamWaitEvent(c->stream, c->ev_halo, 0) != hipSuccess) { st = fail(c, FS3D_ERR_HIP, "halo overlap: event"); return true; }
    p.o_begin = 0; p.o_count = 1;
    launch_sweep_part<R>(c, dir, p);
    p.o_begin = c->dimx - 1; p.o_count = 1;
    launch_sweep_part<R>(c, dir, p);
    p.o_begin = 0; p.o_count = 0;
    return true;
}

// one sweep on explicit buffers; merge: 0 none, 1 fused merge, 2 fused merge twice
// halo: the temp layer's ghost planes are exchanged first -- or beside the interior planes (sweep_overlapped)
template <typename R>
static fs3d_status sweep_buffers(fs3d_ctx *c, int dir, double dt, int b_cur, int b_temp, int b_next, int b_tout, int merge, bool halo = false)
{
    SweepParams<R> p;
    fill_params<R>(c, p, dir, dt, b_cur, b_temp, b_next, b_tout, merge);
    const int cls = dir == 2 ? 0 : (dir == 1 ? 1 : 2);
    if (halo && c->nranks > 1) {
        fs3d_status so;
        rec_begin(c, cls);
        if (sweep_overlapped<R>(c, dir, p, b_temp, so)) {      // the exchange runs beside the interior planes: inside the sweep's time
            rec_end(c);
            if (so) return so;
            HIPCHK(c, hipGetLastError());
            return FS3D_OK;
        }
        if (c->timing) { c->ev_used -= 2; c->ev_class.pop_back(); }      // nothing was launched: drop the event pair
        rec_begin(c, 7);                                       // syncHalos_{X,Y,Z} (AdiSolver3D.cpp:608)
        so = fs3d_comm_halo_exchange(c, b_temp, 4);
        rec_end(c);
        if (so) return so;
    }
    rec_begin(c, cls);
    // (a slab context without peers that asks for the reduced form runs its kernels on the slab alone: tools/slab_cost.py times
    // what one rank computes)
    if (dir == 0 && (c->nranks > 1 || ((c->opt_xsolve == 2 || c->opt_xsolve == 3) && (p.ghost_lo || p.ghost_hi)))) {
        // reduced-interface form (all ranks at once) unless bit-equality with the sequential recurrence was asked for
        const bool reduced = c->opt_xsolve == 2 || c->opt_xsolve == 3 || (c->opt_xsolve == 0 && c->nranks <= FS3D_XREDUCE_MAX_RANKS && (c->opt_kernel == FS3D_SWEEP_AUTO || c->opt_kernel == FS3D_SWEEP_PART));
        fs3d_status st = reduced ? xsweep_reduced<R>(c, p) : xsweep_multi<R>(c, p);
        rec_end(c);
        if (st) return st;
        HIPCHK(c, hipGetLastError());
        return FS3D_OK;
    }
    bool done = false;
    const int ok = c->opt_kernel;
    c->ran_segmented[dir] = 0;
    if (ok == FS3D_SWEEP_AUTO || ok == FS3D_SWEEP_PART) {
        done = launch_sweep_part<R>(c, dir, p);
        if (done) c->ran_kernel[dir] = FS3D_SWEEP_PART;
        else if (ok == FS3D_SWEEP_PART) { rec_end(c); return fail(c, FS3D_ERR_UNSUPPORTED, "partition sweep kernel does not support these dims / this precision"); }
    }
    const bool exact_fast = ok == FS3D_SWEEP_AUTO || ok == FS3D_SWEEP_EXACT || ok == FS3D_SWEEP_PIPE;
    if (!done && exact_fast) { done = launch_sweep_pipe<R>(c, dir, p); if (done) c->ran_kernel[dir] = FS3D_SWEEP_PIPE; }
    if (!done && exact_fast) {
        // lines longer than one launch holds on chip: segment by segment, rows through the HBM scratch
        fs3d_status st = ensure_scratch(c);
        if (st) return st;
        p.scr_ = (R *)c->scr;
        done = launch_sweep_pipe_segmented<R>(c, dir, p);
        if (done) { c->ran_kernel[dir] = FS3D_SWEEP_PIPE; c->ran_segmented[dir] = 1; }
    }
    if (!done) {
        if (ok == FS3D_SWEEP_PIPE) { rec_end(c); return fail(c, FS3D_ERR_UNSUPPORTED, "pipelined sweep kernel does not support these dims (" + c->err + ")"); }
        fs3d_status st = ensure_scratch(c);
        if (st) return st;
        fill_params<R>(c, p, dir, dt, b_cur, b_temp, b_next, b_tout, merge);
        launch_sweep_line<R>(c, dir, p);
        c->ran_kernel[dir] = FS3D_SWEEP_LINE;
    }
    rec_end(c);
    HIPCHK(c, hipGetLastError());
    return FS3D_OK;
}

// measurement: one pipelined sweep with per-wave phase stamps
extern "C" fs3d_status fs3d_profile_sweep(fs3d_ctx *c, int dir, double dt, int l_cur, int l_temp, int l_next,
                                          unsigned long long *stamps_out, int max_blocks, int *n_blocks_out)
{
    if (!c || !stamps_out || !n_blocks_out) return FS3D_ERR_INVALID;
    if (dir < 0 || dir > 2 || check_layer(c, l_cur) || check_layer(c, l_temp) || check_layer(c, l_next) || l_next == l_cur || l_next == l_temp)
        return fail(c, FS3D_ERR_INVALID, "fs3d_profile_sweep: bad direction or layer id");
    if (!c->have_nodes || !c->have_params) return fail(c, FS3D_ERR_INVALID, "fs3d_profile_sweep: upload nodes and set params first");
    HIPCHK(c, hipSetDevice(c->device));
    const int la = dir == 2 ? c->dimy : c->dimz, n_o = dir == 0 ? c->dimy : c->dimx;
    const int nb = n_o * ((la + 31) / 32);      // enough for the partition kernel's 32-line workgroups too
    if (c->stamps_cap < nb) {
        if (c->stamps) hipFree(c->stamps);
        HIPCHK(c, hipMalloc((void **)&c->stamps, sizeof(unsigned long long) * 64 * (size_t)nb));
        c->stamps_cap = nb;
    }
    HIPCHK(c, hipMemsetAsync(c->stamps, 0, sizeof(unsigned long long) * 64 * (size_t)nb, c->stream));
    bool ok;
    if (c->prec == FS3D_F32) {
        SweepParams<float> p; fill_params<float>(c, p, dir, dt, c->slot[l_cur], c->slot[l_temp], c->slot[l_next], c->spare, 1);
        p.stamps = c->stamps;
        ok = (c->opt_kernel == FS3D_SWEEP_AUTO || c->opt_kernel == FS3D_SWEEP_PART) && launch_sweep_part<float>(c, dir, p);
        if (!ok) ok = launch_sweep_pipe<float>(c, dir, p);
    } else {
        SweepParams<double> p; fill_params<double>(c, p, dir, dt, c->slot[l_cur], c->slot[l_temp], c->slot[l_next], c->spare, 1);
        p.stamps = c->stamps; ok = launch_sweep_pipe<double>(c, dir, p);
    }
    if (!ok) return fail(c, FS3D_ERR_UNSUPPORTED, "fs3d_profile_sweep: pipelined kernel does not support these dims");
    HIPCHK(c, hipGetLastError());
    const int nout = nb < max_blocks ? nb : max_blocks;
    HIPCHK(c, hipMemcpyAsync(stamps_out, c->stamps, sizeof(unsigned long long) * 64 * (size_t)nout, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n_blocks_out = nout;
    return FS3D_OK;
}

template <typename R>
static fs3d_status merge_buffers(fs3d_ctx *c, int b_src, int b_dest)
{
    rec_begin(c, 4);
    hipLaunchKernelGGL((k_merge<R>), dim3(grid_for(c->ncell, 256)), dim3(256), 0, c->stream, c->code, c->ncell,
                       fld<R>(c, b_src, 0), fld<R>(c, b_src, 1), fld<R>(c, b_src, 2), fld<R>(c, b_src, 3),
                       fld<R>(c, b_dest, 0), fld<R>(c, b_dest, 1), fld<R>(c, b_dest, 2), fld<R>(c, b_dest, 3));
    rec_end(c);
    HIPCHK(c, hipGetLastError());
    return FS3D_OK;
}

template <typename R>
static fs3d_status update_boundaries_impl(fs3d_ctx *c, bool also_next = false)
{
    if (!c->n_bnd) return FS3D_OK;
    const int b = c->slot[FS3D_LAYER_CUR];
    rec_begin(c, 6);
    if (also_next) {
        const int bn = c->slot[FS3D_LAYER_NEXT];
        hipLaunchKernelGGL((k_impose_list2<R>), dim3((c->n_bnd + 255) / 256), dim3(256), 0, c->stream, c->bnd_idx, c->n_bnd,
                           (const R *)c->bnd_val[0], (const R *)c->bnd_val[1], (const R *)c->bnd_val[2], (const R *)c->bnd_val[3],
                           fld<R>(c, b, 0), fld<R>(c, b, 1), fld<R>(c, b, 2), fld<R>(c, b, 3), fld<R>(c, bn, 0), fld<R>(c, bn, 1), fld<R>(c, bn, 2), fld<R>(c, bn, 3));
    } else
    hipLaunchKernelGGL((k_impose_list<R>), dim3((c->n_bnd + 255) / 256), dim3(256), 0, c->stream, c->bnd_idx, c->n_bnd,
                       (const R *)c->bnd_val[0], (const R *)c->bnd_val[1], (const R *)c->bnd_val[2], (const R *)c->bnd_val[3],
                       fld<R>(c, b, 0), fld<R>(c, b, 1), fld<R>(c, b, 2), fld<R>(c, b, 3));
    rec_end(c);
    HIPCHK(c, hipGetLastError());
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_update_boundaries(fs3d_ctx *c)
{
    if (!c) return FS3D_ERR_INVALID;
    if (!c->have_nodes) return fail(c, FS3D_ERR_INVALID, "fs3d_update_boundaries: upload nodes first");
    HIPCHK(c, hipSetDevice(c->device));
    fs3d_status st = c->prec == FS3D_F32 ? update_boundaries_impl<float>(c) : update_boundaries_impl<double>(c);
    if (st) return st;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FS3D_OK;
}

template <typename R>
static fs3d_status div_error_enqueue(fs3d_ctx *c, int layer)
{
    const int b = c->slot[layer];
    // the last slab skips its final plane (TimeLayer3D.h:606); inner slabs need the i-1 ghost (halo exchanged by caller)
    const int i_end = (c->x_offset + c->dimx == c->dimx_global) ? c->dimx - 1 : c->dimx;
    rec_begin(c, 5);
    hipLaunchKernelGGL((k_div_error<R>), dim3(c->red_blocks), dim3(256), 0, c->stream, c->code,
                       (const R *)fld<R>(c, b, 0), (const R *)fld<R>(c, b, 1), (const R *)fld<R>(c, b, 2),
                       c->dimx, c->dimy, c->dimz, i_end, c->x_offset == 0 ? 1 : 0, (R)c->gdx, (R)c->gdy, (R)c->gdz, c->red_buf + 2,
                       (c->dimy + DIVE_JS - 1) / DIVE_JS);
    hipLaunchKernelGGL(k_div_final, dim3(1), dim3(256), 0, c->stream, c->red_buf + 2, c->red_blocks, c->red_buf);
    rec_end(c);
    HIPCHK(c, hipGetLastError());
    return FS3D_OK;
}

static fs3d_status div_error_finish(fs3d_ctx *c, double *err, long long *count)
{
    fs3d_status st = fs3d_comm_allreduce_sum2(c, c->red_buf);   // no-op for a single rank
    if (st) return st;
    HIPCHK(c, hipMemcpyAsync(c->red_host, c->red_buf, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (count) *count = (long long)c->red_host[1];
    if (err) *err = c->red_host[0] / c->red_host[1];   // err / count (0/0 = NaN as in the reference)
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_eval_div_error(fs3d_ctx *c, int layer, double *err_out, long long *count_out)
{
    if (!c) return FS3D_ERR_INVALID;
    if (check_layer(c, layer)) return FS3D_ERR_INVALID;
    if (!c->have_nodes) return fail(c, FS3D_ERR_INVALID, "fs3d_eval_div_error: upload nodes first");
    HIPCHK(c, hipSetDevice(c->device));
    fs3d_status st = fs3d_comm_halo_exchange(c, c->slot[layer], 3);   // U,V,W ghosts (TimeLayer3D.h:601-603)
    if (st) return st;
    st = c->prec == FS3D_F32 ? div_error_enqueue<float>(c, layer) : div_error_enqueue<double>(c, layer);
    if (st) return st;
    return div_error_finish(c, err_out, count_out);
}

extern "C" fs3d_status fs3d_sweep(fs3d_ctx *c, int dir, double dt, int l_cur, int l_temp, int l_next, int merge_into_temp)
{
    if (!c) return FS3D_ERR_INVALID;
    if (dir < 0 || dir > 2 || check_layer(c, l_cur) || check_layer(c, l_temp) || check_layer(c, l_next))
        return fail(c, FS3D_ERR_INVALID, "fs3d_sweep: bad direction or layer id");
    if (!c->have_nodes || !c->have_params) return fail(c, FS3D_ERR_INVALID, "fs3d_sweep: upload nodes and set params first");
    if (l_next == l_cur || l_next == l_temp) return fail(c, FS3D_ERR_INVALID, "fs3d_sweep: next must differ from cur and temp");
    HIPCHK(c, hipSetDevice(c->device));
    fs3d_status st = fs3d_comm_halo_exchange(c, c->slot[l_temp], 4);   // temp->syncHalos (AdiSolver3D.cpp:608)
    if (st) return st;
    const bool f32 = c->prec == FS3D_F32;
    if (merge_into_temp && c->opt_fuse) {
        const int bt = c->slot[l_temp], bo = c->spare;
        st = f32 ? sweep_buffers<float>(c, dir, dt, c->slot[l_cur], bt, c->slot[l_next], bo, 1)
                 : sweep_buffers<double>(c, dir, dt, c->slot[l_cur], bt, c->slot[l_next], bo, 1);
        if (st) return st;
        c->slot[l_temp] = bo; c->spare = bt;
    } else {
        const int bt = c->slot[l_temp];
        st = f32 ? sweep_buffers<float>(c, dir, dt, c->slot[l_cur], bt, c->slot[l_next], bt, 0)
                 : sweep_buffers<double>(c, dir, dt, c->slot[l_cur], bt, c->slot[l_next], bt, 0);
        if (st) return st;
        if (merge_into_temp) {
            st = f32 ? merge_buffers<float>(c, c->slot[l_next], bt) : merge_buffers<double>(c, c->slot[l_next], bt);
            if (st) return st;
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->ev_used) rec_collect(c);
    return check_device_errors(c);
}

extern "C" fs3d_status fs3d_merge(fs3d_ctx *c, int l_src, int l_dest)
{
    if (!c) return FS3D_ERR_INVALID;
    if (check_layer(c, l_src) || check_layer(c, l_dest) || l_src == l_dest) return fail(c, FS3D_ERR_INVALID, "fs3d_merge: bad layers");
    if (!c->have_nodes) return fail(c, FS3D_ERR_INVALID, "fs3d_merge: upload nodes first");
    HIPCHK(c, hipSetDevice(c->device));
    fs3d_status st = c->prec == FS3D_F32 ? merge_buffers<float>(c, c->slot[l_src], c->slot[l_dest])
                                         : merge_buffers<double>(c, c->slot[l_src], c->slot[l_dest]);
    if (st) return st;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FS3D_OK;
}

// AdiSolver3D::TimeStep, AdiSolver3D.cpp:306-391, enqueued on the context's stream.
template <typename R>
static fs3d_status time_step_enqueue(fs3d_ctx *c, double dt, int G, int L, bool compute_error, bool bnd_copied = false)
{
    const int bCur = c->slot[FS3D_LAYER_CUR], bNext = c->slot[FS3D_LAYER_NEXT], bHalf = c->slot[FS3D_LAYER_HALF];
    fs3d_status st;
    // :310-311  cur -> next on NODE_BOUND and NODE_VALVE
    if (c->n_bnd && !bnd_copied) {
        rec_begin(c, 3);
        hipLaunchKernelGGL((k_copy_list<R>), dim3((c->n_bnd + 255) / 256), dim3(256), 0, c->stream, c->bnd_idx, c->n_bnd,
                           (const R *)fld<R>(c, bCur, 0), (const R *)fld<R>(c, bCur, 1), (const R *)fld<R>(c, bCur, 2), (const R *)fld<R>(c, bCur, 3),
                           fld<R>(c, bNext, 0), fld<R>(c, bNext, 1), fld<R>(c, bNext, 2), fld<R>(c, bNext, 3));
        rec_end(c);
    }
    const bool fuse = c->opt_fuse && L >= 1 && G >= 1;
    if (!fuse) {
        // :320 cur->CopyLayerTo(temp)
        rec_begin(c, 3);
        for (int v = 0; v < 4; v++)
            HIPCHK(c, hipMemcpyAsync(fld<R>(c, c->slot[FS3D_LAYER_TEMP], v), fld<R>(c, bCur, v), (size_t)c->ncell * sizeof(R),
                                     hipMemcpyDeviceToDevice, c->stream));
        rec_end(c);
        for (int it = 0; it < G; it++) {
            const int bT = c->slot[FS3D_LAYER_TEMP];
            const int plan[3][3] = {{2, bCur, bNext}, {1, bNext, bHalf}, {0, bHalf, bNext}};   // :338, :342, :343
            for (int d = 0; d < 3; d++)
                for (int l = 0; l < L; l++) {
                    if ((st = fs3d_comm_halo_exchange(c, bT, 4))) return st;
                    if ((st = sweep_buffers<R>(c, plan[d][0], dt, plan[d][1], bT, plan[d][2], bT, 0))) return st;
                    if ((st = merge_buffers<R>(c, plan[d][2], bT))) return st;                 // :651
                }
            if ((st = merge_buffers<R>(c, bNext, bT))) return st;                               // :354
        }
    } else {
        // Fused form.  The sweep kernel writes next and the merged temp in one pass
        // (temp is double-buffered so cross-line stencil reads still see the un-merged temp);
        // the first sweep reads temp straight from cur (saves the :320 copy); the last X
        // sweep of each global iteration applies the :354 merge as a second averaging step.
        int bTin = bCur;                       // temp == cur until the first merge
        int bTout = c->slot[FS3D_LAYER_TEMP];
        int bSpare = c->spare;
        for (int it = 0; it < G; it++) {
            const int plan[3][3] = {{2, bCur, bNext}, {1, bNext, bHalf}, {0, bHalf, bNext}};
            for (int d = 0; d < 3; d++)
                for (int l = 0; l < L; l++) {
                    // `next` of a local iteration that is not the last of its direction is overwritten by the following
                    // one without having been read (the merge into temp is fused into the kernel): not stored
                    // (r3) ... and so is the `next` of the X sweep that closes a global iteration but the last: the Z sweep of the
                    // following iteration writes `next` on every segment cell before anything reads it (Y reads it as its `cur` on
                    // interior rows only; the result layer is the LAST iteration's) -- provided no NODE_IN cell relies on stale values
                    const bool x_dead = d == 2 && l == L - 1 && it < G - 1 && c->stale_in_cells == 0;
                    int merge = ((d == 2 && l == L - 1) ? 2 : 1) | ((l < L - 1 || x_dead) ? 4 : 0);
                    // (r3) the merged temp of the step's very last sweep is dead too: the next step starts from temp := cur (:320),
                    // GetLayer / EvalDivError read `next` (FS3D_OPT_KEEP_TEMP 1 stores it, as the reference's private member holds it)
                    const bool last_sweep = it == G - 1 && d == 2 && l == L - 1;
                    if (last_sweep && !c->opt_keep_temp && bTin != bCur) merge = 0;
                    if ((st = sweep_buffers<R>(c, plan[d][0], dt, plan[d][1], bTin, plan[d][2], merge ? bTout : bTin, merge, true))) return st;
                    if (merge == 0) continue;                                  // temp stays where it is
                    if (bTin == bCur) { bTin = bTout; bTout = bSpare; }
                    else { int t = bTin; bTin = bTout; bTout = t; }
                }
        }
        c->slot[FS3D_LAYER_TEMP] = bTin; c->spare = bTout;
    }
    if (compute_error) {
        if ((st = fs3d_comm_halo_exchange(c, bNext, 3))) return st;
        if ((st = div_error_enqueue<R>(c, FS3D_LAYER_NEXT))) return st;
    }
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_time_step(fs3d_ctx *c, double dt, int G, int L, int compute_error, double *err_out)
{
    if (!c) return FS3D_ERR_INVALID;
    if (!c->have_nodes || !c->have_params) return fail(c, FS3D_ERR_INVALID, "fs3d_time_step: upload nodes and set params first");
    if (G < 0 || L < 0 || !(dt > 0)) return fail(c, FS3D_ERR_INVALID, "fs3d_time_step: bad dt / iteration counts");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->timing_period > 1) c->timing = (c->timing_steps++ % c->timing_period) == 0;      // sampled timing: every N-th time step carries the events
    fs3d_status st = c->prec == FS3D_F32 ? time_step_enqueue<float>(c, dt, G, L, compute_error != 0)
                                         : time_step_enqueue<double>(c, dt, G, L, compute_error != 0);
    if (st) return st;
    if (compute_error) {
        double e = 0;
        if ((st = div_error_finish(c, &e, nullptr))) return st;
        c->diffError = e;                                                    // :366
    } else {
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (c->ev_used) rec_collect(c);
    if ((st = check_device_errors(c))) return st;
    if (err_out) *err_out = c->diffError;
    if (c->diffError > 0.01) {                                               // :371-374 (ERR_THRESHOLD, AdiSolver3D.h:32)
        char b[128]; snprintf(b, sizeof b, "Error is too big! %f", c->diffError);
        return fail(c, FS3D_ERR_DIVERGED, b);
    }
    std::swap(c->slot[FS3D_LAYER_CUR], c->slot[FS3D_LAYER_NEXT]);            // :388-390
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_time_step_async(fs3d_ctx *c, double dt, int G, int L)
{
    if (!c) return FS3D_ERR_INVALID;
    if (!c->have_nodes || !c->have_params) return fail(c, FS3D_ERR_INVALID, "fs3d_time_step_async: upload nodes and set params first");
    if (G < 0 || L < 0 || !(dt > 0)) return fail(c, FS3D_ERR_INVALID, "fs3d_time_step_async: bad dt / iteration counts");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->timing_period > 1) c->timing = (c->timing_steps++ % c->timing_period) == 0;
    // UpdateBoundaries and the cur -> next copy of the boundary cells that opens TimeStep: one pass over the list
    fs3d_status st = c->prec == FS3D_F32 ? update_boundaries_impl<float>(c, true) : update_boundaries_impl<double>(c, true);
    if (st) return st;
    st = c->prec == FS3D_F32 ? time_step_enqueue<float>(c, dt, G, L, false, true) : time_step_enqueue<double>(c, dt, G, L, false, true);
    if (st) return st;
    std::swap(c->slot[FS3D_LAYER_CUR], c->slot[FS3D_LAYER_NEXT]);
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_last_sweep_kernel(fs3d_ctx *c, int dir, int *kernel_out, int *segmented_out)
{
    if (!c || dir < 0 || dir > 2 || !kernel_out) return FS3D_ERR_INVALID;
    *kernel_out = c->ran_kernel[dir];
    if (segmented_out) *segmented_out = c->ran_segmented[dir] | (dir == 0 ? (c->ran_xsolve << 1) | (c->ran_xa2a << 3) : 0);
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_synchronize(fs3d_ctx *c)
{
    if (!c) return FS3D_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->ev_used) rec_collect(c);
    return check_device_errors(c);
}

// Solver3D::GetLayer, Solver3D.cpp:21-25
template <typename R>
static fs3d_status get_layer_impl(fs3d_ctx *c, R *outV, double *outT, int odx, int ody, int odz)
{
    const int b = c->slot[FS3D_LAYER_NEXT];
    hipLaunchKernelGGL((k_clear_type<R>), dim3(grid_for(c->ncell, 256)), dim3(256), 0, c->stream, c->code, c->ncell,
                       (int)FS3D_NODE_OUT, (R)99999.0f, fld<R>(c, b, 0), fld<R>(c, b, 1), fld<R>(c, b, 2), fld<R>(c, b, 3));
    HIPCHK(c, hipGetLastError());
    std::vector<R> h[4];
    for (int v = 0; v < 4; v++) {
        h[v].resize((size_t)c->ncell);
        HIPCHK(c, hipMemcpyAsync(h[v].data(), fld<R>(c, b, v), (size_t)c->ncell * sizeof(R), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (odx == 0) odx = c->dimx;
    if (ody == 0) ody = c->dimy;
    if (odz == 0) odz = c->dimz;
    // FilterToArrays, TimeLayer3D.h:819-924 (single slab; a multi-GPU caller gathers the slabs)
    for (int i = 0; i < odx; i++)
        for (int j = 0; j < ody; j++)
            for (int k = 0; k < odz; k++) {
                const int x = i * c->dimx / odx, y = j * c->dimy / ody, z = k * c->dimz / odz;
                const size_t ind = (size_t)i * ody * odz + (size_t)j * odz + k;
                const size_t id = (size_t)x * c->plane + (size_t)y * c->dimz + z;
                outV[3 * ind] = h[0][id]; outV[3 * ind + 1] = h[1][id]; outV[3 * ind + 2] = h[2][id];
                outT[ind] = h[3][id];
            }
    return FS3D_OK;
}

extern "C" fs3d_status fs3d_get_layer(fs3d_ctx *c, void *outV, double *outT, int odx, int ody, int odz)
{
    if (!c || !outV || !outT) return FS3D_ERR_INVALID;
    if (!c->have_nodes) return fail(c, FS3D_ERR_INVALID, "fs3d_get_layer: upload nodes first");
    if (odx < 0 || ody < 0 || odz < 0) return fail(c, FS3D_ERR_INVALID, "fs3d_get_layer: negative output dims");
    HIPCHK(c, hipSetDevice(c->device));
    return c->prec == FS3D_F32 ? get_layer_impl<float>(c, (float *)outV, outT, odx, ody, odz)
                               : get_layer_impl<double>(c, (double *)outV, outT, odx, ody, odz);
}
