// Microbenchmark: do fp32 VALU FMAs (v_pk_fma_f32: 64 FLOP/clk/SIMD, the same peak as v_mfma_f32_32x32x2_f32) run BESIDE the fp32 matrix
// instruction on one SIMD, or do the two share their multipliers?  (Question behind it, DESIGN.md section 8: the conv stack sits at half of
// the fp32 matrix instruction's peak and that instruction is the ceiling; an exact-fp32 k-ordered fma chain on the vector ALU is
// bit-identical to the matrix instruction's result, so work moved there would not change a bit -- if the chip can do both at once.)
// One workgroup of 8 waves per CU (two per SIMD), registers only: waves 0-3 issue dependent MFMA chains, waves 4-7 independent packed FMAs.
// Modes: MFMA waves alone, VALU waves alone, both together; and ONE wave per SIMD interleaving VPM packed FMAs behind every MFMA.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_coexec.hip -o tools/mfma_valu_coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float scramble(float a, unsigned m, unsigned c)
{
    return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, a) * m + c) & 0x3fffffffu | 0x3f000000u);
}

// 16 independent accumulator pairs: 16 v_pk_fma_f32 per call (no dependency between them; each pair depends on its own previous value)
#define PKFMA16(A, B)                                                  \
    _Pragma("unroll") for (int u = 0; u < 16; u++)                     \
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[u]) : "v"(A), "v"(B));

// mode bit 0: waves 0-3 run MFMA chains; bit 1: waves 4-7 run packed FMAs.  VPM > 0: waves 0-3 ALSO issue VPM packed FMAs behind every MFMA.
template <int VPM>
__global__ __launch_bounds__(512) void coexec(float* out, int iters, int mode, long long* stamps)
{
    const int wave = threadIdx.x >> 6;
    const bool mf = wave < 4;
    if ((mf && !(mode & 1)) || (!mf && !(mode & 2))) return;
    f32x16 acc = {0};
    f32x2 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = f32x2{0.f, 0.f};
    float a = threadIdx.x * 1e-3f + 0.5f, b = 1.0f + blockIdx.x * 1e-6f;
    f32x2 pa = {a, a * 0.5f}, pb = {b, b * 0.25f};
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (mf) {
        for (int i = 0; i < iters; i++) {
            a = scramble(a, 1664525u, 1013904223u), b = scramble(b, 22695477u, 1u);
            pa[0] = a, pb[1] = b;
#pragma unroll
            for (int u = 0; u < 16; u++) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);  // 4096 flop, 64 cycles
                if constexpr (VPM > 0) {
#pragma unroll
                    for (int k = 0; k < VPM; k++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[(u * VPM + k) & 15]) : "v"(pa), "v"(pb));
                }
            }
        }
    } else {
        for (int i = 0; i < iters; i++) {
            a = scramble(a, 1664525u, 1013904223u), b = scramble(b, 22695477u, 1u);
            pa[0] = a, pb[1] = b;
            // 16 x 16 packed FMAs = 256 per iteration = 65536 flop per wave: the flop count of the MFMA wave's 16 MFMAs
#pragma unroll
            for (int rep = 0; rep < 16; rep++) { PKFMA16(pa, pb) }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int r = 0; r < 16; r++) s += acc[r] + v[r][0] + v[r][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * 8 + wave) * 2] = t1 - t0, stamps[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0;
}

template <int VPM>
static void run(const char* what, int mode, int iters, float* out, long long* st)
{
    const int grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float best = 1e9;
    std::vector<long long> h(grid * 16);
    for (int rep = 0; rep < 4; rep++) {
        hipMemset(st, 0, grid * 16 * 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL(coexec<VPM>, dim3(grid), dim3(512), 0, 0, out, iters, mode, st);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        hipMemcpy(h.data(), st, grid * 16 * 8, hipMemcpyDeviceToHost);
    }
    // per-wave durations (device clock) of wave 0 (MFMA) and wave 4 (VALU) of workgroup 0
    const double mfma_flop = (mode & 1) ? (double)grid * 4 * iters * 16 * 4096.0 : 0.0;
    const double valu_flop = ((mode & 2) ? (double)grid * 4 * iters * 256 * 256.0 : 0.0) + ((mode & 1) ? (double)grid * 4 * iters * 16 * VPM * 256.0 : 0.0);
    const double t_m = h[1] / 100e6, t_v = h[4 * 2 + 1] / 100e6;
    const double clk = (mode & 1) ? (double)h[0] / t_m / 1e9 : (double)h[8] / t_v / 1e9;
    printf("%-64s %7.3f ms | MFMA waves %7.3f ms %6.1f TF/s | VALU %7.3f ms %6.1f TF/s | together %6.1f TF/s | clock %.2f GHz\n", what, best,
           t_m * 1e3, t_m > 0 ? mfma_flop / t_m / 1e12 : 0.0, t_v * 1e3, t_v > 0 && (mode & 2) ? (double)grid * 4 * iters * 256 * 256.0 / t_v / 1e12 : 0.0,
           (mfma_flop + valu_flop) / (best * 1e-3) / 1e12, clk);
}

int main()
{
    float* out;
    long long* st;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&st, 256 * 16 * 8);
    const int iters = 16384;
    run<0>("MFMA waves alone (4 per CU, one per SIMD)", 1, iters, out, st);
    run<0>("VALU waves alone (4 per CU: 256 v_pk_fma_f32 per 16-MFMA-equivalent)", 2, iters, out, st);
    run<0>("both: an MFMA wave and a VALU wave on every SIMD", 3, iters, out, st);
    run<2>("one wave per SIMD: 2 packed FMAs behind every MFMA", 1, iters, out, st);
    run<4>("one wave per SIMD: 4 packed FMAs behind every MFMA", 1, iters, out, st);
    run<8>("one wave per SIMD: 8 packed FMAs behind every MFMA", 1, iters, out, st);
    run<14>("one wave per SIMD: 14 packed FMAs behind every MFMA", 1, iters, out, st);
    run<4>("MFMA + 4 packed FMAs per MFMA, and a VALU wave beside it", 3, iters, out, st);
    return 0;
}
