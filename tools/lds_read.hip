// Microbenchmark: cycles per ds_read_b128 for the fragment-read patterns of the conv kernel (tuning aid).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 256) smem[i] = i;
    __syncthreads();
    int off[4];
    for (int q = 0; q < 4; q++) {
        if (MODE == 0) off[q] = lane * 4 + q * 256;                                                    // linear
        if (MODE == 1) off[q] = (lane & 31) * 32 + (((2 * q + (lane >> 5)) ^ ((lane >> 1) & 7)) * 4);  // conv kernel (swizzled 128-B rows)
        if (MODE == 2) off[q] = (lane & 31) * 32 + ((2 * q + (lane >> 5)) * 4);                        // 128-B rows, no swizzle
        if (MODE == 3) off[q] = (lane & 31) * 36 + 4 * (lane >> 5) + 8 * q;                            // 36-float pitch
    }
    const float* base = smem + (wave & 1) * 2048;
    f32x4 s = {0, 0, 0, 0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int q = 0; q < 4; q++) s += *(const f32x4*)(base + off[q]);
        asm volatile("" ::: "memory");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * 256] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int iters = 4096;
    const char* names[4] = {"linear lane*16", "swizzled 128-B rows (conv kernel)", "128-B rows unswizzled", "36-float pitch"};
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 40960, 0, out, cyc, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 40960, 0, out, cyc, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 40960, 0, out, cyc, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 40960, 0, out, cyc, iters);
            hipDeviceSynchronize();
        }
        long long h[256];
        hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        printf("%-36s %.1f cycles per ds_read_b128 per wave (4 waves/CU reading)\n", names[mode], (double)h[0] / (iters * 4.0));
    }
    return 0;
}
