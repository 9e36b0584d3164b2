// Streaming-rate micro-benchmark for the access shapes of the sweep kernels (MI355X).
// A "sheet" kernel moves NF fields in and NF fields out with the partition kernel's shape: a workgroup of
// LT x NCH threads, thread (kk, ch) walks M consecutive rows of its chunk and touches 4 bytes of each row,
// so a wave-wide access is 64/LT row pieces of LT*4 bytes; rows are `pitch` elements apart (Y sweep: dimz,
// X sweep: dimy*dimz).  No arithmetic: what the memory system delivers for the shape, at the kernel's occupancy.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream.hip -o tools/ubench/stream && tools/ubench/stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NF>
__global__ void __launch_bounds__(256) k_copy4(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n4_per_field, size_t fstride4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4_per_field; i += (size_t)gridDim.x * blockDim.x) {
        float4 v[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) v[f] = in[f * fstride4 + i];
#pragma unroll
        for (int f = 0; f < NF; f++) out[f * fstride4 + i] = v[f];
    }
}

// dir 0: X-like (sweep stride = plane, o stride = dimz); dir 1: Y-like (sweep stride = dimz, o stride = plane)
template <int LT, int M, int NCH, int NFI, int NFO, int WPS, int NRR = 0, int NNB = 0>
__global__ void __launch_bounds__(LT * NCH, WPS) k_sheet(const float *__restrict__ in, float *__restrict__ out, int dim, int dir, size_t fstride, int n_o, int n_tiles, int order)
{
    int lb = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = lb & 7, slot = lb >> 3;
        lb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + slot;
    }
    const int tile = order ? lb % n_tiles : lb / n_o, o = order ? lb / n_tiles : lb - tile * n_o;
    const int t = threadIdx.x, kk = t % LT, ch = t / LT;
    const size_t plane = (size_t)dim * dim;
    const size_t ss = dir == 0 ? plane : dim, os = dir == 0 ? dim : plane;
    const size_t base = (size_t)o * os + (size_t)(ch * M) * ss + tile * LT + kk;
    float keep = 0.f;
    constexpr int PF = 2;
    float v[PF + 1][NFI];
#pragma unroll
    for (int i = 0; i < PF; i++)
#pragma unroll
        for (int f = 0; f < NFI; f++) v[i][f] = in[f * fstride + base + i * ss];
    float res[M][NFO > 0 ? NFO : 1];
#pragma unroll
    for (int i = 0; i < M; i++) {
        if (i + PF < M) {
#pragma unroll
            for (int f = 0; f < NFI; f++) v[(i + PF) % (PF + 1)][f] = in[f * fstride + base + (i + PF) * ss];
        }
        // NNB: the rows of the neighbouring workgroups (o +- 1) of field 0, as the stencils of the real kernels read them
        float nb = 0.f;
        if (NNB) { nb = in[base + i * ss + os] + in[base + i * ss - os]; }
        __builtin_amdgcn_sched_barrier(0);
        float s = 0.f;
#pragma unroll
        for (int f = 0; f < NFI; f++) s += v[i % (PF + 1)][f];
#pragma unroll
        for (int f = 0; f < NFO; f++) res[i][f] = s + (float)f;
        keep += s + nb;
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (NFO == 0 && keep == 12345.678f) out[base] = keep;      // loads-only variant: keep the loads alive
#pragma unroll
    for (int i = 0; i < M; i++) {
        // NRR: fields read a second time by the store phase (the merge's temp values in the real kernels)
        float rr = 0.f;
#pragma unroll
        for (int f = 0; f < NRR; f++) rr += in[(4 + f) * fstride + base + i * ss];
#pragma unroll
        for (int f = 0; f < NFO; f++) out[f * fstride + base + i * ss] = res[i][f] + rr;
        __builtin_amdgcn_sched_barrier(0);
    }
}

static float time_ms(hipStream_t st, int reps, void (*fn)(void *), void *ctx)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    fn(ctx); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; i++) fn(ctx);
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

struct Ctx { const float *in; float *out; int dim; size_t fstride; hipStream_t st; int dir, order; };

template <int LT, int M, int NCH, int NFI, int NFO, int WPS, int NRR = 0, int NNB = 0>
static void run_sheet(Ctx &c, const char *name)
{
    for (int dir = 0; dir < 2; dir++) for (int order = 0; order < 2; order++) {
        c.dir = dir; c.order = order;
        auto fn = [](void *p) {
            Ctx &c = *(Ctx *)p;
            const int n_o = c.dim, n_tiles = c.dim / LT;
            hipLaunchKernelGGL((k_sheet<LT, M, NCH, NFI, NFO, WPS, NRR, NNB>), dim3(n_o * n_tiles), dim3(LT * NCH), 0, c.st, c.in, c.out, c.dim, c.dir, c.fstride, n_o, n_tiles, c.order);
        };
        const float ms = time_ms(c.st, 10, fn, &c);
        const double bytes = (double)c.dim * c.dim * c.dim * 4.0 * (NFI + NFO);
        printf("%-44s dir %c order %d: %.4f ms  %.2f TB/s\n", name, dir == 0 ? 'X' : 'Y', order, ms, bytes / ms / 1e9);
    }
}

int main()
{
    const int dim = 256;
    const size_t ncell = (size_t)dim * dim * dim, fstride = ncell + 2 * (size_t)dim * dim;
    float *in, *out;
    CK(hipMalloc(&in, 8 * fstride * 4)); CK(hipMalloc(&out, 8 * fstride * 4));
    CK(hipMemset(in, 0, 8 * fstride * 4)); CK(hipMemset(out, 0, 8 * fstride * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    Ctx c{in + dim * dim, out + dim * dim, dim, fstride, st, 0, 0};
    {
        auto fn = [](void *p) { Ctx &c = *(Ctx *)p; hipLaunchKernelGGL((k_copy4<8>), dim3(4096), dim3(256), 0, c.st, (const float4 *)c.in, (float4 *)c.out, (size_t)c.dim * c.dim * c.dim / 4, c.fstride / 4); };
        const float ms = time_ms(st, 10, fn, &c);
        printf("%-44s: %.4f ms  %.2f TB/s\n", "float4 copy, 8 fields in / 8 out", ms, (double)ncell * 4 * 16 / ms / 1e9);
        auto fn2 = [](void *p) { Ctx &c = *(Ctx *)p; hipLaunchKernelGGL((k_copy4<4>), dim3(8192), dim3(256), 0, c.st, (const float4 *)c.in, (float4 *)c.out, (size_t)c.dim * c.dim * c.dim / 4, c.fstride / 4); };
        const float ms2 = time_ms(st, 10, fn2, &c);
        printf("%-44s: %.4f ms  %.2f TB/s\n", "float4 copy, 4 fields in / 4 out", ms2, (double)ncell * 4 * 8 / ms2 / 1e9);
    }
    run_sheet<32, 16, 16, 8, 8, 4>(c, "sheet LT 32 M 16 (512 thr, 2/CU) 8 in 8 out");
    run_sheet<64, 16, 16, 8, 8, 4>(c, "sheet LT 64 M 16 (1024 thr, 1/CU) 8 in 8 out");
    run_sheet<64, 16, 16, 8, 8, 4, 4, 0>(c, "sheet LT 64: 8 in, 4 read again, 8 out");
    run_sheet<64, 16, 16, 8, 8, 4, 4, 1>(c, "sheet LT 64: 8 in + o+-1 rows, 4 again, 8 out");
    run_sheet<32, 16, 16, 8, 0, 4>(c, "sheet LT 32 M 16 loads only (8 in)");
    run_sheet<64, 16, 16, 8, 0, 4>(c, "sheet LT 64 M 16 loads only (8 in)");
    run_sheet<32, 16, 16, 1, 8, 4>(c, "sheet LT 32 M 16 stores (1 in 8 out)");
    run_sheet<64, 16, 16, 1, 8, 4>(c, "sheet LT 64 M 16 stores (1 in 8 out)");
    run_sheet<32, 32, 8, 8, 8, 2>(c, "sheet LT 32 M 32 (256 thr, 2/CU) 8 in 8 out");
    return 0;
}
