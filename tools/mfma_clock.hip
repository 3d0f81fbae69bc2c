// Microbenchmark: f32 MFMA issue rate and shader clock for short and long kernels (tuning aid).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_clock.hip -o tools/mfma_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, long long* stamps)
{
    f32x16 acc = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int r = 0; r < 16; r++) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

int main()
{
    float* out;
    long long* st;
    hipMalloc(&out, 1024 * 256 * 4);
    hipMalloc(&st, 1024 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int grid : {256, 512}) {
        for (int iters : {16, 64, 256, 4096, 65536}) {
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                int n = iters >= 4096 ? 1 : 20;
                for (int k = 0; k < n; k++) hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, iters, st);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                std::vector<long long> h(grid * 2);
                hipMemcpy(h.data(), st, grid * 16, hipMemcpyDeviceToHost);
                double flops = (double)grid * 4 * iters * 16 * 4096.0 * n;
                double clk = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
                if (rep == 2)
                    printf("grid %4d iters %6d: %.3f ms / %d launches, %.1f TF/s, wave cycles %lld, in-kernel clock %.2f GHz, cycles/mfma %.1f\n",
                           grid, iters, ms, n, flops / (ms * 1e-3) / 1e12, h[0], clk, (double)h[0] / (iters * 16.0));
            }
        }
    }
    return 0;
}
