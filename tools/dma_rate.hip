// Microbenchmark: per-CU global->LDS (LDS-DMA) and global->VGPR fill rate for cache-hot data (tuning aid).
// One workgroup per CU; W loader waves; each wave-instruction moves 1 KiB (64 lanes x 16 B: 8 rows x 128 B).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GLDS16(gp, lp) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp), (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)

template <int MODE>  // 0 = LDS-DMA, 1 = VGPR loads
__global__ __launch_bounds__(512) void k(const float* src, float* out, long long* cyc, int iters, int span_kb)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // each workgroup walks its own `span_kb` window (hot after the first pass), 1 KiB per wave-instruction
    const float* base = src + (size_t)blockIdx.x * span_kb * 256;
    f32x4 s = {0, 0, 0, 0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        const int piece = (i * nw + wave) % span_kb;  // 1 KiB piece index inside the window
        const float* p = base + piece * 256 + lane * 4;
        if (MODE == 0) {
            GLDS16(p, smem + ((i & 7) * nw + wave) * 256);
            if ((i & 3) == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else {
            s += *(const f32x4*)p;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) s = *(const f32x4*)(smem + lane * 4);
    out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    float *src, *out;
    long long* cyc;
    hipMalloc(&src, (size_t)256 * 1024 * 1024);
    hipMemset(src, 0, (size_t)256 * 1024 * 1024);
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int iters = 2048;
    for (int span_kb : {16, 256, 1024}) {          // per-workgroup window: L1-ish, L2, beyond L2 (256 WGs x 1 MiB = 256 MiB)
        for (int waves : {1, 4, 8}) {
            for (int mode = 0; mode < 2; mode++) {
                for (int rep = 0; rep < 2; rep++) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(waves * 64), 65536, 0, src, out, cyc, iters, span_kb);
                    else hipLaunchKernelGGL(k<1>, dim3(256), dim3(waves * 64), 65536, 0, src, out, cyc, iters, span_kb);
                    hipDeviceSynchronize();
                }
                long long h[256];
                hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
                double avg = 0;
                for (int i = 0; i < 256; i++) avg += h[i];
                avg /= 256;
                double bytes = (double)iters * waves * 1024;
                printf("window %4d KiB/WG  %d loader waves  %-8s %.1f B/clk/CU  (%.0f GB/s per CU at 2.4 GHz)\n", span_kb, waves,
                       mode == 0 ? "LDS-DMA" : "VGPR", bytes / avg, bytes / avg * 2.4);
            }
        }
    }
    return 0;
}
