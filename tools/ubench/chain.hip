// Micro-benchmark: cost of the Thomas forward chain per cell on one wave (measurement aid, not product code).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o chain chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CH 32
__device__ __forceinline__ bool good(float v) { return ((__builtin_bit_cast(unsigned, v) << 1) - 2u) >= (2u * (27u << 23) - 2u); }   // 0 or |v| >= 2^-100
template <int MODE, int ACTIVE>
__global__ void __launch_bounds__(512, 2) k(const float *in, float *out, unsigned long long *cyc, float vis, float bb)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float q[CH], d[CH];
#pragma unroll
    for (int t = 0; t < CH; t++) { q[t] = in[(t * 512 + threadIdx.x)]; d[t] = in[(CH + t) * 512 + threadIdx.x]; }
    if (MODE == 2) {
#pragma unroll
        for (int t = 0; t < CH; t++) lds[(w * CH + t) * 64 + lane] = d[t];
    }
    __syncthreads();
    float cp = 0.f, dp = 0.f;
    bool ok = true;
    unsigned long long t0 = 0, t1 = 0;
    if (w < ACTIVE) {
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int t = 0; t < CH; t++) {
            const float a = -q[t] - vis, c = q[t] - vis;
            const float dd = MODE == 2 ? lds[(w * CH + t) * 64 + lane] : d[t];
            const float den = bb - a * cp;
            const float num = dd - dp * a;
            if (MODE == 3) {            // shared refined reciprocal + two corrections per quotient (IEEE result when nothing needs scaling)
                float r = __builtin_amdgcn_rcpf(den);
                const float e = __builtin_fmaf(-den, r, 1.0f);
                r = __builtin_fmaf(e, r, r);
                float qc = c * r, qd = num * r;
                float rc = __builtin_fmaf(-den, qc, c), rd = __builtin_fmaf(-den, qd, num);
                qc = __builtin_fmaf(rc, r, qc); qd = __builtin_fmaf(rd, r, qd);
                rc = __builtin_fmaf(-den, qc, c); rd = __builtin_fmaf(-den, qd, num);
                cp = __builtin_fmaf(rc, r, qc); dp = __builtin_fmaf(rd, r, qd);
                ok = ok & good(den) & good(c) & good(num);
            } else if (MODE == 1) {            // reciprocal-multiply (NOT exact; lower bound of the chain latency)
                const float r = __builtin_amdgcn_rcpf(den);
                cp = c * r; dp = num * r;
            } else {
                cp = c / den; dp = num / den;
            }
            if (MODE == 2) lds[(w * CH + t) * 64 + lane] = dp; else d[t] = dp;
            q[t] = cp;
            if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    __syncthreads();
    float s = ok ? 0.f : 1.f;
#pragma unroll
    for (int t = 0; t < CH; t++) s += q[t] + (MODE == 2 ? lds[(w * CH + t) * 64 + lane] : d[t]);
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

template <int MODE, int ACTIVE>
static void run(const char *name, float *in, float *out, unsigned long long *cyc, int blocks)
{
    const size_t lds = 130 * 1024;
    hipFuncSetAttribute((const void *)k<MODE, ACTIVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<MODE, ACTIVE>), dim3(blocks), dim3(512), lds, 0, in, out, cyc, 0.3f, 31.f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < blocks; b++) for (int w = 0; w < ACTIVE; w++) { sum += (double)h[b * 8 + w]; n++; }
    printf("%-34s active waves %d: %8.1f ticks per %d-cell chain = %6.1f ticks/cell\n", name, ACTIVE, sum / n, CH, sum / n / CH);
}

int main()
{
    const int blocks = 256;
    float *in, *out; unsigned long long *cyc;
    hipMalloc(&in, 2 * CH * 512 * 4); hipMalloc(&out, (size_t)blocks * 16 * 512 * 4); hipMalloc(&cyc, (size_t)blocks * 16 * 8 * 8);   // sized for the largest launch below
    std::vector<float> h(2 * CH * 512);
    for (size_t i = 0; i < h.size(); i++) h[i] = 0.01f * (float)((i * 7919) % 97) + 0.1f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    // clock calibration: ticks of s_memtime per microsecond
    run<0, 1>("IEEE div, registers", in, out, cyc, blocks);
    run<0, 4>("IEEE div, registers", in, out, cyc, blocks);
    run<0, 8>("IEEE div, registers", in, out, cyc, blocks);
    run<1, 1>("rcp*mul (inexact), registers", in, out, cyc, blocks);
    run<1, 8>("rcp*mul (inexact), registers", in, out, cyc, blocks);
    run<3, 1>("fast exact + guard, registers", in, out, cyc, blocks);
    run<3, 8>("fast exact + guard, registers", in, out, cyc, blocks);
    run<2, 1>("IEEE div, d in LDS", in, out, cyc, blocks);
    run<2, 4>("IEEE div, d in LDS", in, out, cyc, blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<0, 8>), dim3(blocks * 16), dim3(512), 130 * 1024, 0, in, out, cyc, 0.3f, 31.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("k<0,8> x 4096 blocks: %.3f ms per launch\n", ms / 20);
    return 0;
}
