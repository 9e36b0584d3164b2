// Micro-benchmark: issue cost of fp32 division variants with independent operands (measurement aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define NB 16
__device__ __forceinline__ float div_fast(float x, float y)
{
    // the non-scaled core of the IEEE expansion: rcp, one Newton step, two corrections
    float r = __builtin_amdgcn_rcpf(y);
    const float e = __builtin_fmaf(-y, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = x * r;
    float rem = __builtin_fmaf(-y, q, x);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-y, q, x);
    return __builtin_fmaf(rem, r, q);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 div_fast2(f32x2 x, float y)
{
    float r = __builtin_amdgcn_rcpf(y);
    const float e = __builtin_fmaf(-y, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    const f32x2 ny = {-y, -y}, rr = {r, r};
    f32x2 q = x * rr;
    f32x2 rem = __builtin_elementwise_fma(ny, q, x);
    q = __builtin_elementwise_fma(rem, rr, q);
    rem = __builtin_elementwise_fma(ny, q, x);
    return __builtin_elementwise_fma(rem, rr, q);
}
template <int MODE>
__global__ void __launch_bounds__(512, 2) k(const float *in, float *out, unsigned long long *cyc, float y0)
{
    extern __shared__ float lds[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float x[NB];
#pragma unroll
    for (int t = 0; t < NB; t++) x[t] = in[t * 512 + threadIdx.x];
    float y = y0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; it++) {
#pragma unroll
        for (int t = 0; t < NB; t++) {
            if (MODE == 0) x[t] = x[t] / y;
            if (MODE == 1) x[t] = div_fast(x[t], y);
            if (MODE == 2) x[t] = x[t] * y;
            if (MODE == 3 && (t & 1) == 0) { f32x2 v = {x[t], x[t + 1]}; v = div_fast2(v, y); x[t] = v.x; x[t + 1] = v.y; }
        }
        asm volatile("" : "+v"(y));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int t = 0; t < NB; t++) s += x[t];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

template <int MODE>
static void run(const char *name, float *in, float *out, unsigned long long *cyc)
{
    const int blocks = 256;
    const size_t lds = 130 * 1024;
    (void)hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), lds, 0, in, out, cyc, 1.0000001f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    printf("%-28s 8 waves/CU (2 per SIMD): %7.1f ticks per operation per wave\n", name, sum / h.size() / (64.0 * NB));
}

int main()
{
    float *in, *out; unsigned long long *cyc;
    (void)hipMalloc(&in, NB * 512 * 4); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<float> h(NB * 512);
    for (size_t i = 0; i < h.size(); i++) h[i] = 1.0f + 0.001f * (float)(i % 977);
    (void)hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("IEEE division (compiler)", in, out, cyc);
    run<1>("rcp + 7 fma (no scaling)", in, out, cyc);
    run<2>("multiply", in, out, cyc);
    run<3>("packed: rcp + 5 pk ops per 2", in, out, cyc);
    return 0;
}
