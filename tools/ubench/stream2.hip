// What does the memory system of an MI355X deliver for a sweep's bytes (8 fields in, 8 fields out, 1.07 GB at 256^3 fp32),
// as a function of: the number of concurrent streams the same bytes are split into (16 = today's SoA layers, 4 = the four
// fields of a layer interleaved per cell), nontemporal hints, bytes in flight per thread, grid size, field padding.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream2.hip -o tools/ubench/stream2 && tools/ubench/stream2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// NS streams in, NS streams out, each of n4 float4; thread handles UN consecutive float4-strides per stream per iteration
template <int NS, int UN, int NT>      // NT bit 0: nontemporal loads, bit 1: nontemporal stores
__global__ void __launch_bounds__(256) k_copy(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n4, size_t stride4)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += step * UN) {
        f4 v[NS][UN];
#pragma unroll
        for (int u = 0; u < UN; u++)
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const size_t j = i + u * step;
                if (j < n4) v[s][u] = (NT & 1) ? __builtin_nontemporal_load(&in[s * stride4 + j]) : in[s * stride4 + j];
            }
#pragma unroll
        for (int u = 0; u < UN; u++)
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const size_t j = i + u * step;
                if (j < n4) { if (NT & 2) __builtin_nontemporal_store(v[s][u], &out[s * stride4 + j]); else out[s * stride4 + j] = v[s][u]; }
            }
    }
}

template <int NS, int UN, int NT>
__global__ void __launch_bounds__(256) k_read(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n4, size_t stride4)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += step * UN) {
#pragma unroll
        for (int u = 0; u < UN; u++)
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const size_t j = i + u * step;
                if (j < n4) acc += (NT & 1) ? __builtin_nontemporal_load(&in[s * stride4 + j]) : in[s * stride4 + j];
            }
    }
    if (acc.x == 1234.5f) out[0] = acc;
}

template <int NS, int UN, int NT>
__global__ void __launch_bounds__(256) k_write(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n4, size_t stride4)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    const f4 v = {1, 2, 3, (float)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += step * UN) {
#pragma unroll
        for (int u = 0; u < UN; u++)
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const size_t j = i + u * step;
                if (j < n4) { if (NT & 2) __builtin_nontemporal_store(v, &out[s * stride4 + j]); else out[s * stride4 + j] = v; }
            }
    }
}

static hipStream_t st;
template <class F> static float time_ms(F fn, int reps = 10)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    fn(); fn(); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; i++) fn();
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

static f4 *in_, *out_;
static const size_t ncell = 256ull * 256 * 256;

template <int NS, int UN, int NT>
static void run(int blocks, size_t pad, int what)
{
    // total bytes fixed: 8 fields in + 8 out of ncell floats; split into NS streams each way
    const size_t n4 = ncell * 8 / NS / 4;
    const size_t stride4 = n4 + pad / 4;
    float ms;
    const char *nm = what == 0 ? "copy " : what == 1 ? "read " : "write";
    if (what == 0) ms = time_ms([&] { hipLaunchKernelGGL((k_copy<NS, UN, NT>), dim3(blocks), dim3(256), 0, st, in_, out_, n4, stride4); });
    else if (what == 1) ms = time_ms([&] { hipLaunchKernelGGL((k_read<NS, UN, NT>), dim3(blocks), dim3(256), 0, st, in_, out_, n4, stride4); });
    else ms = time_ms([&] { hipLaunchKernelGGL((k_write<NS, UN, NT>), dim3(blocks), dim3(256), 0, st, in_, out_, n4, stride4); });
    const double bytes = (double)ncell * 4 * 8 * (what == 0 ? 2 : 1);
    printf("%s streams %2d+%2d unroll %d nt %d blocks %5d pad %5zu: %.4f ms  %.2f TB/s\n", nm, what == 2 ? 0 : NS, what == 1 ? 0 : NS, UN, NT, blocks, pad, ms, bytes / ms / 1e9);
}

int main()
{
    const size_t bytes = ncell * 4 * 8 + (64 << 20);
    CK(hipMalloc(&in_, bytes)); CK(hipMalloc(&out_, bytes));
    CK(hipMemset(in_, 0, bytes)); CK(hipMemset(out_, 0, bytes));
    CK(hipStreamCreate(&st));
    for (size_t pad : {(size_t)0, (size_t)576}) {
        run<8, 1, 0>(4096, pad, 0); run<4, 1, 0>(4096, pad, 0); run<2, 1, 0>(4096, pad, 0); run<1, 1, 0>(4096, pad, 0);
    }
    const size_t pad = 576;
    for (int blocks : {1024, 2048, 8192, 16384}) { run<8, 1, 0>(blocks, pad, 0); run<2, 1, 0>(blocks, pad, 0); }
    run<8, 1, 1>(4096, pad, 0); run<8, 1, 2>(4096, pad, 0); run<8, 1, 3>(4096, pad, 0);
    run<2, 1, 1>(4096, pad, 0); run<2, 1, 2>(4096, pad, 0); run<2, 1, 3>(4096, pad, 0);
    run<2, 2, 0>(2048, pad, 0); run<2, 4, 0>(2048, pad, 0); run<2, 4, 3>(2048, pad, 0); run<1, 4, 0>(2048, pad, 0); run<1, 8, 0>(2048, pad, 0); run<1, 8, 3>(1024, pad, 0);
    run<8, 2, 0>(2048, pad, 0);
    run<8, 1, 0>(4096, pad, 1); run<2, 1, 0>(4096, pad, 1); run<2, 4, 0>(2048, pad, 1); run<2, 4, 1>(2048, pad, 1);
    run<8, 1, 0>(4096, pad, 2); run<2, 1, 0>(4096, pad, 2); run<2, 4, 0>(2048, pad, 2); run<2, 4, 2>(2048, pad, 2);
    return 0;
}
