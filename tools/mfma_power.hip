// Microbenchmark: sustained shader clock and f32 MFMA rate against (a) the number of busy CUs and (b) the MFMA shape.
// Question behind it: layers with 200 tiles leave 56 of 256 CUs idle -- would filling them give 28 % more, or is the
// chip power-limited under fp32 MFMA load so that the busy CUs slow down?  One workgroup of 4 waves (one per SIMD) per CU.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o tools/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, long long* stamps)
{
    f32x16 acc = {0};
    f32x4 acc4 = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        // operands that change like real data (constant operands toggle nothing: 2.40 GHz at any CU count)
        a = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, a) * 1664525u + 1013904223u) & 0x3fffffffu | 0x3f000000u);
        b = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, b) * 22695477u + 1u) & 0x3fffffffu | 0x3f000000u);
        if (SHAPE == 32) {
#pragma unroll
            for (int u = 0; u < 16; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);   // 16 x 4096 flop
        } else {
#pragma unroll
            for (int u = 0; u < 32; u++) acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4, 0, 0, 0);  // 32 x 2048 flop
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = acc4[0] + acc4[1] + acc4[2] + acc4[3];
    for (int r = 0; r < 16; r++) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) stamps[blockIdx.x * 2] = t1 - t0, stamps[blockIdx.x * 2 + 1] = r1 - r0;
}

int main()
{
    float* out;
    long long* st;
    hipMalloc(&out, 1024 * 256 * 4);
    hipMalloc(&st, 1024 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int iters = 32768;  // ~15 ms per launch
    for (int shape : {32, 16}) {
        for (int grid : {64, 128, 200, 256}) {
            float best = 1e9;
            std::vector<long long> h(grid * 2);
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                if (shape == 32) hipLaunchKernelGGL(mfma_loop<32>, dim3(grid), dim3(256), 0, 0, out, iters, st);
                else hipLaunchKernelGGL(mfma_loop<16>, dim3(grid), dim3(256), 0, 0, out, iters, st);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
                hipMemcpy(h.data(), st, grid * 16, hipMemcpyDeviceToHost);
            }
            const double flops = (double)grid * 4 * iters * 16 * 4096.0;
            const double clk = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
            printf("mfma %dx%d: %3d busy CUs: %7.3f ms, %6.1f TF/s, %.3f TF/s per CU, in-kernel clock %.2f GHz\n", shape, shape, grid, best,
                   flops / (best * 1e-3) / 1e12, flops / (best * 1e-3) / 1e12 / grid, clk);
        }
    }
    return 0;
}
