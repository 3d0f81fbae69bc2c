// Microbenchmark: the consumer loop of the conv kernel in isolation (tuning aid).
// Per "chunk": 16 dependent v_mfma_f32_32x32x2_f32, operands from registers loaded by 8 ds_read_b128 one chunk earlier.
// Variants: 0 = no LDS reads, 1 = reads interleaved one per MFMA, 2 = reads in one burst, 3 = variant 1 + s_barrier per chunk,
//           4 = variant 3 with 4 extra idle waves in the barrier (8 waves per workgroup, like the conv kernel)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define GLDS16(gp, lp) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp), (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(768) void k(float* out, long long* cyc, int chunks, const float* src)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    for (int i = threadIdx.x; i < 20480; i += blockDim.x) smem[i] = (i & 255) * 1e-3f;
    __syncthreads();
    if (threadIdx.x >= 256) {  // partner waves: V == 4 only take part in the barriers, V >= 5 also run the LDS-DMA ring
        if (V == 9) __builtin_amdgcn_s_setprio(3);
        if (V == 11 || V == 12) {
            // classic path: global_load_dwordx4 -> VGPR (3 chunks in flight) -> ds_write_b128 into the stage of chunk t+1
            const float* win = src + (size_t)blockIdx.x * (V == 12 ? 262144 : 4096) + lane * 4;
            const int wmask = (V == 12 ? 63 : 0);
            const int pwv = (threadIdx.x >> 6) - 4;
            f32x4 r0[4], r1[4], r2[4], r3[4];
            auto ld = [&](f32x4(&r)[4], int c) {
                const float* g = win + (size_t)(c & wmask) * 4096 + pwv * 256;
#pragma unroll
                for (int i = 0; i < 4; i++) r[i] = __builtin_nontemporal_load((const f32x4*)(g + i * 1024));
            };
            auto st = [&](f32x4(&r)[4], int c) {
                float* sb = smem + (c % 5) * 4096 + pwv * 256 + lane * 4;
#pragma unroll
                for (int i = 0; i < 4; i++) *(f32x4*)(sb + i * 1024) = r[i];
            };
            ld(r0, 0), ld(r1, 1), ld(r2, 2), ld(r3, 3);
            for (int t = 0; t < chunks; t += 4) {
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); st(r1, t + 1); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier(); ld(r0, t + 4);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); st(r2, t + 2); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier(); ld(r1, t + 5);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); st(r3, t + 3); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier(); ld(r2, t + 6);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); st(r0, t + 4); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier(); ld(r3, t + 7);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (V >= 5) {
            // 5-stage ring of 16 KiB chunks, 4 x 1 KiB LDS-DMA per wave per chunk, hot source window per workgroup
            const float* win = src + (size_t)blockIdx.x * (V == 6 ? 262144 : 4096) + lane * 4;   // V==6: 1 MiB window (beyond L1)
            const int wmask = (V == 6 ? 63 : 0);
            auto issue = [&](int stage, int c) {
                const int pwv = (threadIdx.x >> 6) - 4, npw = (blockDim.x >> 6) - 4;  // producer wave index / count
                float* sb = smem + stage * 4096 + pwv * 256;
                const float* g = win + (size_t)(c & wmask) * 4096 + pwv * 256;
#pragma unroll
                for (int i = 0; i < (V == 8 || V == 10 ? 2 : 4); i++) GLDS16(g + i * (256 * npw), sb + i * (256 * npw));
            };
            for (int p = 0; p < 4; p++) issue(p, p);
            int stage = 0;
            long long pw = 0, pi = 0;
            for (int t = 0; t < chunks; t++) {
                long long q0 = V == 7 ? __builtin_amdgcn_s_memtime() : 0;
                if (V == 8 || V == 10) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // chunk t+1 landed (2 younger chunks of 4 instr in flight)
                long long q1 = V == 7 ? __builtin_amdgcn_s_memtime() : 0;
                __builtin_amdgcn_s_barrier();
                long long q2 = V == 7 ? __builtin_amdgcn_s_memtime() : 0;
                issue(stage == 0 ? 4 : stage - 1, t + 4);
                long long q3 = V == 7 ? __builtin_amdgcn_s_memtime() : 0;
                pw += q1 - q0, pi += q3 - q2;
                if (V == 7 && t == chunks - 1 && threadIdx.x == 256) cyc[1024 + blockIdx.x] = pw, cyc[1536 + blockIdx.x] = pi, cyc[2048 + blockIdx.x] = 0;
                stage = stage == 4 ? 0 : stage + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            for (int t = 0; t < chunks; t++) __builtin_amdgcn_s_barrier();
        }
        return;
    }
    int fo[4];
    for (int q = 0; q < 4; q++) fo[q] = (lane & 31) * 32 + (((2 * q + (lane >> 5)) ^ ((lane >> 1) & 7)) * 4);
    const float* Ab = smem + (wave >> 1) * 1024;
    const float* Bb = smem + 4096 + (wave & 1) * 1024;
    f32x4 a0[4], b0[4], a1[4], b1[4];
    for (int q = 0; q < 4; q++) a0[q] = a1[q] = *(const f32x4*)(Ab + fo[q]), b0[q] = b1[q] = *(const f32x4*)(Bb + fo[q]);
    f32x16 acc = {0};
    long long bwait = 0;
    long long rt0 = __builtin_amdgcn_s_memrealtime();
    long long t0 = __builtin_amdgcn_s_memtime();
    auto step = [&](f32x4(&ca)[4], f32x4(&cb)[4], f32x4(&na)[4], f32x4(&nb)[4], int t) {
        const float* A2 = V >= 5 ? smem + ((t + 1) % 5) * 4096 + (wave >> 1) * 1024 : Ab + (t & 1) * 2048;
        if (V == 2) {
#pragma unroll
            for (int q = 0; q < 4; q++) na[q] = *(const f32x4*)(A2 + fo[q]), nb[q] = *(const f32x4*)(Bb + fo[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[q][e], cb[q][e], acc, 0, 0, 0);
                const int r = q * 4 + e;
                if ((V == 1 || (V >= 3 && V != 7)) && r < 8) {
                    if (r & 1) nb[r >> 1] = *(const f32x4*)(Bb + fo[r >> 1]);
                    else na[r >> 1] = *(const f32x4*)(A2 + fo[r >> 1]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if (V >= 3 && r == 0) {
                    if (V == 7) {
                        long long b0 = __builtin_amdgcn_s_memtime();
                        __builtin_amdgcn_s_barrier();
                        long long b1 = __builtin_amdgcn_s_memtime();
                        bwait += b1 - b0;
                    } else __builtin_amdgcn_s_barrier();
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = 0; t < chunks; t += 2) {
        step(a0, b0, a1, b1, t);
        step(a1, b1, a0, b0, t + 1);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    long long rt1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int r = 0; r < 16; r++) s += acc[r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0, cyc[512 + blockIdx.x] = bwait, cyc[2560 + blockIdx.x] = rt1 - rt0;
}

int main()
{
    float* out;
    long long* cyc;
    hipMalloc(&out, 512 * 256 * 4);
    hipMalloc(&cyc, 4096 * 8);
    const int chunks = 512;
    const char* names[13] = {"MFMA only", "reads interleaved 1/MFMA", "reads in one burst", "interleaved + s_barrier/chunk (4 waves)",
                            "interleaved + s_barrier/chunk (8 waves, 4 idle)", "+ 4 producer waves, LDS-DMA ring, L1-hot source",
                            "+ 4 producer waves, LDS-DMA ring, 1 MiB window", "producers (hot) but consumers read nothing",
                            "producers (hot) move half the bytes (2 DMA/wave/chunk)", "4 producers at s_setprio 3 (hot, full bytes)",
                            "8 producer waves x 2 DMA (hot, full bytes)",
                            "4 producers, global_load -> VGPR -> ds_write (hot)", "4 producers, global_load -> VGPR -> ds_write (1 MiB window)"};
    float* src;
    hipMalloc(&src, (size_t)512 * 262144 * 4);
    hipMemset(src, 0, (size_t)512 * 262144 * 4);
    hipFuncSetAttribute((const void*)k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipFuncSetAttribute((const void*)k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipFuncSetAttribute((const void*)k<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipFuncSetAttribute((const void*)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipFuncSetAttribute((const void*)k<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipFuncSetAttribute((const void*)k<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipFuncSetAttribute((const void*)k<11>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipFuncSetAttribute((const void*)k<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    for (int v = 0; v < 5; v++) {}
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    for (int wgs : {256, 512}) {
        for (int v = 0; v < 13; v++) {
            int threads = v == 10 ? 768 : (v >= 4 ? 512 : 256);
            for (int rep = 0; rep < 2; rep++) {
                if (v == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(threads), 81920, 0, out, cyc, chunks, src);
                if (v == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(threads), 81920, 0, out, cyc, chunks, src);
                if (v == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(threads), 81920, 0, out, cyc, chunks, src);
                if (v == 3) hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(threads), 81920, 0, out, cyc, chunks, src);
                if (v == 4) hipLaunchKernelGGL(k<4>, dim3(wgs), dim3(threads), 81920, 0, out, cyc, chunks, src);
                if (v == 5) hipLaunchKernelGGL(k<5>, dim3(wgs), dim3(threads), 81920, 0, out, cyc, chunks, src);
                if (v == 6) hipLaunchKernelGGL(k<6>, dim3(wgs), dim3(threads), 81920, 0, out, cyc, chunks, src);
                if (v == 7) hipLaunchKernelGGL(k<7>, dim3(wgs), dim3(threads), 81920, 0, out, cyc, chunks, src);
                if (v == 8) hipLaunchKernelGGL(k<8>, dim3(wgs), dim3(threads), 81920, 0, out, cyc, chunks, src);
                if (v == 9) hipLaunchKernelGGL(k<9>, dim3(wgs), dim3(threads), 81920, 0, out, cyc, chunks, src);
                if (v == 11) hipLaunchKernelGGL(k<11>, dim3(wgs), dim3(threads), 81920, 0, out, cyc, chunks, src);
                if (v == 12) hipLaunchKernelGGL(k<12>, dim3(wgs), dim3(threads), 81920, 0, out, cyc, chunks, src);
                if (v == 10) hipLaunchKernelGGL(k<10>, dim3(wgs), dim3(threads), 81920, 0, out, cyc, chunks, src);
                hipDeviceSynchronize();
            }
            long long h[512];
            hipMemcpy(h, cyc, wgs * 8, hipMemcpyDeviceToHost);
            double avg = 0;
            for (int i = 0; i < wgs; i++) avg += h[i];
            avg /= wgs;
            long long hr[512];
            hipMemcpy(hr, cyc + 2560, wgs * 8, hipMemcpyDeviceToHost);
            double ravg = 0;
            for (int i = 0; i < wgs; i++) ravg += hr[i];
            ravg /= wgs;
            printf("%3d WGs  %-48s %.0f cycles per chunk (16 MFMA = 1024 ideal)  %.3f us/chunk, s_memtime runs at %.0f MHz\n", wgs, names[v], avg / chunks, ravg / chunks / 100.0, avg / ravg * 100.0);
            if (v == 7) {
                long long g[2048];
                hipMemcpy(g, cyc, sizeof g, hipMemcpyDeviceToHost);
                printf("         consumer barrier wait %.0f cyc/chunk; producer: landing wait %.0f, issue of 4 DMA %.0f cyc/chunk\n",
                       (double)g[512] / chunks, (double)g[1024] / chunks, (double)g[1536] / chunks);
            }
        }
    }
    return 0;
}
