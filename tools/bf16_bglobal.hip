// Microbenchmark for ONE question (DESIGN 4.1b vs section 8): in bf16 the conv loop is bound by the bytes that go
// INTO the LDS, so does it pay to keep only the ACTIVATIONS (A) in the LDS-DMA ring and read the WEIGHTS (B) straight from global
// memory into registers, with the four consumer waves arranged 4 x 1 over M (tile 128 x 32: one B block per workgroup, shared by
// nobody, no B bytes through the LDS at all) -- for the long-K layers of the 23 x 23 stage (M = 1587, N = 256, K = 1024 / 2304)?
// The product runs these layers as 64 x 32 tiles with 2 in-workgroup K groups (200 workgroups, A AND B through the ring: 24 KiB per
// 128-element K step); tools/layer_table.py (LT_BF16=1) gives its times on the same box.  This file times the variant on the same
// shapes as a plain GEMM C[m][n] = sum_k A[m][k] B[n][k] (1x1 conv; a 3x3 layer moves the same bytes per K step), bf16 in, fp32
// accumulate, bf16 out, L2-warm operands like the product's steady state.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bf16_bglobal tools/bf16_bglobal.hip && tools/bf16_bglobal
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* ldsp;
#define GLDS16(gp, lp) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp), (ldsp)(lp), 16, 0, 0)
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int NS = 5;                 // ring stages of 128 rows x 128 B = 16 KiB
constexpr int STAGE = 128 * 32;       // floats

struct P {
    const __bf16* a;   // [Mpad][K]
    const __bf16* b;   // [N][K]
    __bf16* c;         // [Mpad][N]
    unsigned long long* stamps;  // [2 * grid]: start / end of every workgroup (100 MHz)
    int K, N;
};

// 128 x 32 tile: consumer wave w owns rows 32 w .. 32 w + 31 (one 32x32 accumulator); A chunks (128 rows x 64 bf16) by LDS-DMA from 4
// producer waves, XOR-swizzled like the product's ring; B fragments (32 columns x 64 bf16 per chunk = 4 x 16 B per lane) by
// global_load_dwordx4, two chunks ahead.
__global__ __launch_bounds__(512, 4) void gemm_bglobal(const P p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;
    const int tiles_n = p.N / 32, tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int G = p.K / 64;  // chunks
    if (threadIdx.x == 0) p.stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    if (producer) {
        // piece j (0..15) of a chunk = rows 8 j .. 8 j + 7; wave w issues pieces w, w + 4, w + 8, w + 12
        const __bf16* src[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = (wave + 4 * i) * 8 + (lane >> 3);
            const int unit = (lane & 7) ^ ((row >> 1) & 7);
            src[i] = p.a + (size_t)(tm * 128 + row) * p.K + unit * 8;
        }
        int nis = 0;
        auto issue = [&]() __attribute__((always_inline)) {
            float* sb = smem + (nis % NS) * STAGE + wave * 256;
#pragma unroll
            for (int i = 0; i < 4; i++) GLDS16(src[i] + (size_t)nis * 64, sb + i * 4 * 256);
            nis++;
        };
        for (int k = 0; k < NS - 1 && k < G; k++) issue();
        for (int g = 0; g < G; g++) {
            // chunk g must have landed before barrier g; up to min(nis - g - 1, ...) younger chunks may stay in flight
            const int young = nis - g - 1;
            if (young >= 3) wait_vm<12>();
            else if (young == 2) wait_vm<8>();
            else if (young == 1) wait_vm<4>();
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();  // chunk g visible; the consumers are past chunk g - 1
            if (nis < G) issue();          // into the stage of chunk g - 1
        }
        __builtin_amdgcn_s_barrier();
        return;
    }
    // consumers
    const int col = lane & 31, hh = lane >> 5;
    int fo[4];
#pragma unroll
    for (int q = 0; q < 4; q++) fo[q] = (wave * 32 + col) * 32 + (((2 * q + hh) ^ ((col >> 1) & 7)) * 4);
    typedef __attribute__((address_space(1))) const f32x4 cgf4;
    const __bf16* bp = p.b + (size_t)(tn * 32 + col) * p.K + hh * 8;
    auto ldb = [&](f32x4(&B)[4], int g) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; q++) B[q] = *(cgf4*)(bp + (size_t)g * 64 + q * 16);
    };
    f32x4 B0[4], B1[4], B2[4], A[4];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    ldb(B0, 0);
    if (G > 1) ldb(B1, 1);
    auto step = [&](int g, f32x4(&Bc)[4], f32x4(&Bn)[4]) __attribute__((always_inline)) {
        __builtin_amdgcn_s_barrier();  // chunk g in LDS
        const float* Ab = smem + (g % NS) * STAGE;
#pragma unroll
        for (int q = 0; q < 4; q++) A[q] = *(const f32x4*)(Ab + fo[q]);
        if (g + 2 < G) ldb(Bn, g + 2);
#pragma unroll
        for (int q = 0; q < 4; q++)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[q]), __builtin_bit_cast(bf16x8, Bc[q]), acc, 0, 0, 0);
    };
    int g = 0;
    for (; g + 2 < G; g += 3) step(g, B0, B2), step(g + 1, B1, B0), step(g + 2, B2, B1);
    if (g < G) step(g, B0, B2), g++;
    if (g < G) step(g, B1, B0), g++;
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int m = tm * 128 + wave * 32 + 4 * hh + (r & 3) + 8 * (r >> 2);
        p.c[(size_t)m * p.N + tn * 32 + col] = (__bf16)acc[r];
    }
    if (threadIdx.x == 0) p.stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

int main()
{
    const int M = 1587, Mpad = 13 * 128, N = 256;
    printf("variant: 128x32 tiles, A by LDS-DMA ring (16 KiB per 64-element K chunk), B global -> registers (4 x 4 KiB per chunk through the "
           "vector-memory path, no LDS); %d workgroups (product: 64x32x2, 200 workgroups, 12 KiB per 64-element chunk per workgroup)\n",
           (Mpad / 128) * (N / 32));
    for (int K : {1024, 2304}) {
        __bf16 *a, *b, *c;
        unsigned long long* st;
        const int grid = (Mpad / 128) * (N / 32);
        CK(hipMalloc(&a, (size_t)Mpad * K * 2)); CK(hipMalloc(&b, (size_t)N * K * 2)); CK(hipMalloc(&c, (size_t)Mpad * N * 2));
        CK(hipMalloc(&st, grid * 16));
        std::vector<unsigned short> ha((size_t)Mpad * K), hb((size_t)N * K);
        for (size_t i = 0; i < ha.size(); i++) ha[i] = (unsigned short)(0x3c00 + (((unsigned)i * 2654435761u) >> 26));  // 0.0078 .. 0.0117: small positive bf16 values
        for (size_t i = 0; i < hb.size(); i++) hb[i] = (unsigned short)(0x3a00 + ((((unsigned)i * 40503u) >> 10) & 63));
        CK(hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
        const size_t lds = (size_t)NS * STAGE * 4;
        CK(hipFuncSetAttribute((const void*)gemm_bglobal, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        P p{a, b, c, st, K, N};
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(gemm_bglobal, dim3(grid), dim3(512), lds, 0, p);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 200;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(gemm_bglobal, dim3(grid), dim3(512), lds, 0, p);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> hs(2 * grid);
        CK(hipMemcpy(hs.data(), st, grid * 16, hipMemcpyDeviceToHost));
        unsigned long long s0 = ~0ull, s1 = 0;
        for (int i = 0; i < grid; i++) s0 = std::min(s0, hs[2 * i]), s1 = std::max(s1, hs[2 * i + 1]);
        // spot check of one output against the host (bf16 inputs, double accumulation)
        std::vector<unsigned short> hc((size_t)Mpad * N);
        CK(hipMemcpy(hc.data(), c, hc.size() * 2, hipMemcpyDeviceToHost));
        auto f = [](unsigned short v) { unsigned u = (unsigned)v << 16; float x; memcpy(&x, &u, 4); return (double)x; };
        double worst = 0;
        for (int t = 0; t < 64; t++) {
            const int m = (t * 97) % M, n = (t * 41) % N;
            double ref = 0;
            for (int k = 0; k < K; k++) ref += f(ha[(size_t)m * K + k]) * f(hb[(size_t)n * K + k]);
            worst = std::max(worst, fabs(f(hc[(size_t)m * N + n]) - ref) / fabs(ref));
        }
        const double flops = 2.0 * M * N * K;
        printf("K = %4d: slot (launch to launch, events) %.2f us, first-to-last wave %.2f us, %.1f TFLOP/s on the slot; A+B bytes through the "
               "vector-memory path per workgroup %.0f KiB; max rel err of 64 samples %.3g\n",
               K, ms * 1e3 / reps, (s1 - s0) * 0.01, flops / (ms * 1e-3 / reps) / 1e12, (K / 64) * 32.0, worst);
        (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(st);
    }
    return 0;
}
