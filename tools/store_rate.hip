// store_rate.hip -- what does the WIDTH of a write-through (sc1) store cost when every wave instruction still covers whole
// 128-byte lines?  The conv epilogue stores one fp32 per lane (64 lanes = two full lines per instruction).  Variants:
//   0: dword  sc1, lane i -> base + 4 i           (the epilogue's pattern)
//   1: dwordx4 sc1, lane i -> base + 16 i          (8 full lines per instruction)
//   2: dword  plain        3: dwordx4 plain
//   4: dwordx4 sc1 SCATTERED: lane i -> row i of a 1 KB-stride matrix (what a lane-per-pixel epilogue does)
// Each workgroup (256 threads) writes a contiguous 64 KB block, the grid covers `MB` megabytes, repeated `reps` times from a graph.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/store_rate tools/store_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* out, float v)
{
    float* base = out + (size_t)blockIdx.x * 16384;  // 64 KB per workgroup
    const int tid = threadIdx.x;
    if (MODE == 0 || MODE == 2) {
#pragma unroll 16
        for (int i = 0; i < 64; i++) {
            float* p = base + i * 256 + tid;
            if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *p = v;
        }
    } else if (MODE == 1 || MODE == 3) {
#pragma unroll 16
        for (int i = 0; i < 16; i++) {
            f32x4* p = (f32x4*)(base + i * 1024) + tid;
            const f32x4 w = {v, v, v, v};
            if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
            else *p = w;
        }
    } else {
        // 64 rows x 256 floats per workgroup; lane = row (per wave: 64 rows), 16-byte unit u of the row per instruction
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll 16
        for (int i = 0; i < 16; i++) {
            f32x4* p = (f32x4*)(base + lane * 256 + wave * 64 + i * 4);
            const f32x4 w = {v, v, v, v};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
        }
    }
}

int main(int argc, char** argv)
{
    const int MB = argc > 1 ? atoi(argv[1]) : 26;
    const int reps = 50;
    const size_t bytes = (size_t)MB << 20;
    float* buf;
    CK(hipMalloc(&buf, bytes));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = (int)(bytes / 65536);
    const char* names[5] = {"dword sc1 (coalesced)", "dwordx4 sc1 (coalesced)", "dword plain", "dwordx4 plain", "dwordx4 sc1, lane = row (scattered)"};
    printf("%d MB per pass, %d workgroups of 256 threads\n", MB, grid);
    for (int mode = 0; mode < 5; mode++) {
        for (int w = 0; w < 2; w++) {
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; r++) {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(256), 0, st, buf, 1.f); break;
                    case 1: hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(256), 0, st, buf, 1.f); break;
                    case 2: hipLaunchKernelGGL(store_kernel<2>, dim3(grid), dim3(256), 0, st, buf, 1.f); break;
                    case 3: hipLaunchKernelGGL(store_kernel<3>, dim3(grid), dim3(256), 0, st, buf, 1.f); break;
                    default: hipLaunchKernelGGL(store_kernel<4>, dim3(grid), dim3(256), 0, st, buf, 1.f); break;
                }
            }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (w == 1) printf("%-40s %8.2f us per pass  %7.2f TB/s\n", names[mode], ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
        }
    }
    return 0;
}
