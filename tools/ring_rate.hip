// Microbenchmark: the conv kernel's LDS ring in isolation (tuning aid) -- how fast can loader waves fill 16 KiB stages
// (64 A rows + 64 B rows x 128 B, XOR-swizzled source units) while consumer waves run the fp32 MFMA chain?
// Variables: the LDS-DMA instruction form (global_load_lds with 64-bit lane addresses / buffer_load ... offen lds with a
// 32-bit lane offset and a scalar chunk offset), number of loader waves, consumers on/off, workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ring_rate tools/ring_rate.hip && tools/ring_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GLDS16(gp, lp) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp), (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)
typedef __attribute__((address_space(3))) void* ldsp;

#ifndef RING_NS
#define RING_NS 5
#endif
constexpr int NS = RING_NS;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct P {
    const float* a;      // A windows: [window][64 rows][kwin*32 floats]
    const float* b;      // B window (shared): [64 rows][kwin*32 floats]
    float* out;
    long long* res;      // per WG: {cycles, realtime ticks}
    int chunks, kwin, share, mfma, reads;
    int lfirst, cons_waves, m16, lprio, cprio, two_acc;  // first active loader wave; consumer waves that compute; 16x16x4 MFMAs; priorities
};

// MODE 0: global_load_lds (64-bit addresses); 1: buffer_load offen lds (voffset + scalar chunk offset); 2: MODE 0 without swizzle
// LW: loader waves (1, 2, 4); each chunk = 16 wave-instructions of 1 KiB, LW waves issue 16/LW each
template <int MODE, int LW, int NM, int M16, int RD, int SCHED, int PCS, int TWO>
__global__ __launch_bounds__(SCHED == 7 ? 768 : 512) void ring(const P p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CT = SCHED == 7 ? 512 : 256;  // consumer threads (SCHED 7: EIGHT consumer waves, two per SIMD, each on half of a chunk's K)
    const int tid = threadIdx.x >= CT ? threadIdx.x - CT : threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = __builtin_amdgcn_readfirstlane(threadIdx.x) >= CT;
    const int G = p.chunks;
    constexpr int PER = PCS / LW;  // DMA instructions per loader wave per chunk
    constexpr int STAGE = PCS * 256;  // floats per stage
    long long t0 = 0, r0 = 0;
    if (producer) {
        const int lw = wave - p.lfirst;
        if (lw < 0 || lw >= LW) {  // idle loader waves still take part in the barriers
            __builtin_amdgcn_s_barrier();
            for (int g = 0; g < (SCHED == 4 ? G / 2 : G); g++) __builtin_amdgcn_s_barrier();
            return;
        }
        if (p.lprio) __builtin_amdgcn_s_setprio(3);
        const int srow = lane >> 3;
        const float* aw = p.a + (size_t)(blockIdx.x / p.share) * 64 * p.kwin * 32;
        // piece j (0..15) of a chunk: rows 8j..8j+7 of the 128-row stage (0-63 A, 64-127 B); wave w takes pieces w, w+LW, ...
        const float* src[PER];
        unsigned voff[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int piece = lw + i * LW, row = piece * 8 + srow;
            const int unit = MODE == 2 ? (lane & 7) : ((lane & 7) ^ ((row >> 1) & 7));
            const float* base = row < 64 ? aw + (size_t)row * p.kwin * 32 : p.b + (size_t)((row - 64) & 63) * p.kwin * 32 + (row >= 128 ? 16 * p.kwin * 32 : 0);
            src[i] = base + unit * 4;
            voff[i] = (unsigned)((const char*)src[i] - (const char*)p.a);
        }
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-resource type does not exist in the host pass
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, 0x7fffffff, 0x00020000);
#endif
        int cc = 0;
        auto issue = [&](int stage) __attribute__((always_inline)) {
            float* sb = smem + stage * STAGE + lw * 256;
#pragma unroll
            for (int i = 0; i < PER; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
                if (MODE == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (ldsp)(sb + i * LW * 256), 16, voff[i], cc * 128, 0, 0);
                else
#endif
                    GLDS16(src[i] + cc * 32, sb + i * LW * 256);
            }
            if (++cc == p.kwin) cc = 0;
        };
        auto wait_landed = [&](int young) __attribute__((always_inline)) {
            switch (young) {
                case 1: wait_vm<PER>(); break;
                case 2: wait_vm<2 * PER>(); break;
                case 3: wait_vm<(3 * PER > 63 ? 63 : 3 * PER)>(); break;
                case 4: wait_vm<(4 * PER > 63 ? 63 : 4 * PER)>(); break;
                case 5: wait_vm<(5 * PER > 63 ? 63 : 5 * PER)>(); break;
                case 0: wait_vm<0>(); break;
                default: if (young > 5) wait_vm<(5 * PER > 63 ? 63 : 5 * PER)>(); else wait_vm<0>(); break;
            }
        };
        t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        if constexpr (MODE == 3) {
            // register staging: global_load_dwordx4 (3 chunks in flight) -> ds_write_b128 at the swizzled slot of the next stage
            f32x4 R0[PER], R1[PER], R2[PER];
            int ldsoff[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int piece = lw + i * LW, row = piece * 8 + srow;
                ldsoff[i] = row * 32 + (((lane & 7) ^ ((row >> 1) & 7)) * 4);
                src[i] = (row < 64 ? aw + (size_t)row * p.kwin * 32 : p.b + (size_t)(row - 64) * p.kwin * 32) + (lane & 7) * 4;
            }
            auto ld = [&](f32x4(&R)[PER]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < PER; i++) R[i] = *(const f32x4*)(src[i] + cc * 32);
                if (++cc == p.kwin) cc = 0;
            };
            auto st = [&](f32x4(&R)[PER], int stage) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < PER; i++) *(f32x4*)(smem + stage * STAGE + ldsoff[i]) = R[i];
            };
            ld(R0), ld(R1), ld(R2);
            wait_vm<2 * PER>();
            st(R0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int stage = 0;
            auto body = [&](f32x4(&Rnew)[PER], f32x4(&Rnext)[PER]) __attribute__((always_inline)) {
                const int nstage = stage + 1 == NS ? 0 : stage + 1;
                ld(Rnew);
                wait_vm<2 * PER>();
                st(Rnext, nstage);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                stage = nstage;
            };
            for (int g = 0; g < G; g += 3) body(R0, R1), body(R1, R2), body(R2, R0);  // G is a multiple of 6
            wait_vm<0>();
            if (threadIdx.x == CT) {
                p.res[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
                p.res[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
            }
            return;
        }
        if constexpr (SCHED == 4) {
            // TWO chunks per barrier (round 5 probe): barrier P opens chunks 2P, 2P + 1; the consumers prefetch chunk 2P + 2's fragments behind
            // chunk 2P + 1's MFMAs, so chunks <= 2P + 2 must have landed at barrier P; behind it the stages of chunks 2P - 2, 2P - 1 are refilled
            // with chunks 2P - 2 + NS, 2P - 1 + NS.  G even.
            int issued = 0;
            auto put = [&]() __attribute__((always_inline)) { issue(issued % NS); issued++; };
            for (int q = 0; q < NS - 2 && q < G; q++) put();  // chunks 0 .. NS - 3 before barrier 0 (stages NS - 2, NS - 1 play "chunks -2, -1")
            // landed(c): at most (issued - 1 - c) chunks younger than c may be in flight
            auto need = [&](int c) __attribute__((always_inline)) {
                const int young = issued - 1 - (c < G - 1 ? c : G - 1);
                wait_landed(young < 0 ? 0 : young);
            };
            need(2);
            __builtin_amdgcn_s_barrier();  // barrier 0
            for (int P2 = 0; P2 < G / 2; P2++) {
                // behind barrier P: refill
                if (issued < G) put();
                if (issued < G) put();
                need(2 * (P2 + 1) + 2);
                __builtin_amdgcn_s_barrier();  // barrier P + 1
            }
            if (threadIdx.x == CT) {
                p.res[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
                p.res[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < NS - 1; q++)
            if (q < G) issue(q);
        wait_landed(G - 1 < NS - 2 ? G - 1 : NS - 2);
        __builtin_amdgcn_s_barrier();
        int stage = 0;
        for (int g = 0; g < G; g++) {
            const int young = G - 2 - g;
            wait_landed(young < NS - 3 ? young : NS - 3);
            __builtin_amdgcn_s_barrier();
            if (g + NS - 1 < G) issue(stage == 0 ? NS - 1 : stage - 1);
            stage = stage + 1 == NS ? 0 : stage + 1;
        }
        if (threadIdx.x == CT) {
            p.res[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
            p.res[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
        }
        return;
    }
    // consumers
    if (SCHED != 7 && wave >= p.cons_waves) {
        __builtin_amdgcn_s_barrier();
        for (int g = 0; g < (SCHED == 4 ? G / 2 : G); g++) __builtin_amdgcn_s_barrier();
        return;
    }
    if (p.cprio) __builtin_amdgcn_s_setprio(3);
    f32x4 c4[4];
#pragma unroll
    for (int i = 0; i < 4; i++) c4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sub = SCHED == 7 ? wave >> 2 : 0;  // SCHED 7: K half of the chunk (fragments 2 sub, 2 sub + 1)
    const int wm = (wave & 3) >> 1, wn = wave & 1;
    int fo[4];
#pragma unroll
    for (int q = 0; q < 4; q++) fo[q] = (lane & 31) * 32 + (((2 * q + (lane >> 5)) ^ ((lane >> 1) & 7)) * 4);
    struct Frag { f32x4 a[4], b[4]; };
    Frag F0, F1;
    f32x16 acc, accb;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f, accb[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; q++) F0.a[q] = F0.b[q] = F1.a[q] = F1.b[q] = f32x4{1.f, 2.f, 3.f, 4.f};
    int stage = 0;
#define MMA(x, y, e)                                                                                     \
    do {                                                                                                 \
        if (M16) {                                                                                       \
            c4[(e & 1) * 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c4[(e & 1) * 2], 0, 0, 0);         \
            c4[(e & 1) * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, c4[(e & 1) * 2 + 1], 0, 0, 0); \
        } else if (TWO == 2)                                                                             \
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y));            \
        else if (TWO && ((e) & 1))                                                               \
            accb = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, accb, 0, 0, 0);                            \
        else                                                                                             \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);                              \
    } while (0)
    // SCHED 6 (round 5 probe): the B fragments come straight from global memory / L2 -- three chunks ahead, in registers -- and only the 4 A
    // fragments are read from LDS: half the ds_read_b128 per step, 4 global_load_dwordx4 instead
    f32x4 Bg[3][4];
    typedef __attribute__((address_space(1))) const f32x4 cgf4;
    cgf4* bsrc = (cgf4*)(p.b + (size_t)(wn * 32 + (lane & 31)) * p.kwin * 32 + (lane >> 5) * 4);
    int bcc = 0;
    auto ldB = [&](f32x4(&R)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; q++) R[q] = bsrc[bcc * 8 + 2 * q];
        if (++bcc == p.kwin) bcc = 0;
    };
    if constexpr (SCHED == 6) ldB(Bg[0]), ldB(Bg[1]), ldB(Bg[2]);
    auto step6 = [&](Frag& cur, Frag& nxt, f32x4(&Bcur)[4]) __attribute__((always_inline)) {
        const int nstage = stage + 1 == NS ? 0 : stage + 1;
        __builtin_amdgcn_s_barrier();
        const float* Ab = smem + nstage * STAGE + (wm * 32) * 32;
        f32x4 Bn[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = q * 4 + e;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[q][e], Bcur[q][e], acc, 0, 0, 0);
                if ((i & 3) == 1) {
                    nxt.a[i >> 2] = *(const f32x4*)(Ab + fo[i >> 2]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                } else if ((i & 3) == 3) {
                    Bn[i >> 2] = bsrc[bcc * 8 + 2 * (i >> 2)];
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one VMEM read
                }
            }
        if (++bcc == p.kwin) bcc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) Bcur[q] = Bn[q];  // (the slot just consumed takes the chunk three ahead)
        __builtin_amdgcn_sched_barrier(0);
        stage = nstage;
    };
    auto step7 = [&](Frag& cur, Frag& nxt) __attribute__((always_inline)) {
        // eight consumer waves: this one takes fragments q = 2 sub, 2 sub + 1 of its block (16 of the chunk's 32 k): 8 MFMAs, 4 reads
        const int nstage = stage + 1 == NS ? 0 : stage + 1;
        __builtin_amdgcn_s_barrier();
        const float* Ab = smem + nstage * STAGE + (wm * 32) * 32;
        const float* Bb = smem + nstage * STAGE + (64 + wn * 32) * 32;
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[q][e], cur.b[q][e], acc, 0, 0, 0);
                if (e & 1) {
                    const int r = q * 2 + (e >> 1);  // 0..3
                    if (r & 1) nxt.b[r >> 1] = *(const f32x4*)(Bb + (sub ? fo[2 + (r >> 1)] : fo[r >> 1]));
                    else nxt.a[r >> 1] = *(const f32x4*)(Ab + (sub ? fo[2 + (r >> 1)] : fo[r >> 1]));
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        stage = nstage;
    };
    auto step = [&](Frag& cur, Frag& nxt) __attribute__((always_inline)) {
        const int nstage = stage + 1 == NS ? 0 : stage + 1;
        if constexpr (SCHED >= 1) {
            // barrier first, then the 16 MFMAs of this chunk with the 8 reads of the next one behind every second MFMA (SCHED 1)
            // or behind MFMAs 0-7 (SCHED 2) or 4-11 (SCHED 3); SCHED 4: as 1, the barrier is the caller's (every second chunk)
            if constexpr (SCHED != 4) __builtin_amdgcn_s_barrier();
            const float* Ab = smem + nstage * STAGE + (wm * 32) * 32;
            const float* Bb = smem + nstage * STAGE + (64 + wn * 32) * 32;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int i = q * 4 + e;
                    MMA(cur.a[q][e], cur.b[q][e], e);
                    const int r = (SCHED == 1 || SCHED == 4) ? (i & 1 ? i >> 1 : -1) : SCHED == 2 ? (i < 8 ? i : -1) : (i >= 4 && i < 12 ? i - 4 : -1);
                    if (r >= 0) {
                        if (r & 1) nxt.b[r >> 1] = *(const f32x4*)(Bb + fo[r >> 1]);
                        else nxt.a[r >> 1] = *(const f32x4*)(Ab + fo[r >> 1]);
                        __builtin_amdgcn_sched_group_barrier(0x008, (SCHED == 1 || SCHED == 4) ? 2 : 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            stage = nstage;
            return;
        }
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (q * 4 + e < NM) MMA(cur.a[q][e], cur.b[q][e], e);
        __builtin_amdgcn_s_barrier();
        const float* Ab = smem + nstage * STAGE + (wm * 32) * 32;
        const float* Bb = smem + nstage * STAGE + (64 + wn * 32) * 32;
#pragma unroll
        for (int q = 2; q < 4; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (q * 4 + e < NM) MMA(cur.a[q][e], cur.b[q][e], e);
                const int r = (q - 2) * 4 + e;
                if (RD) {
                    if (r & 1) nxt.b[r >> 1] = *(const f32x4*)(Bb + fo[r >> 1]);
                    else nxt.a[r >> 1] = *(const f32x4*)(Ab + fo[r >> 1]);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        stage = nstage;
    };
    __builtin_amdgcn_s_barrier();
    if constexpr (SCHED == 7) {
        for (int g = 0; g < G; g += 2) {
            step7(F0, F1);
            step7(F1, F0);
        }
    } else
    if constexpr (SCHED == 6) {
        for (int g = 0; g < G; g += 6) {  // G is a multiple of 6
            step6(F0, F1, Bg[0]), step6(F1, F0, Bg[1]), step6(F0, F1, Bg[2]);
            step6(F1, F0, Bg[0]), step6(F0, F1, Bg[1]), step6(F1, F0, Bg[2]);
        }
        for (int k = 0; k < 3; k++)
            for (int q = 0; q < 4; q++) acc[0] += Bg[k][q][0];
    } else
    if constexpr (SCHED == 4) {
        for (int g = 0; g < G; g += 2) {  // barrier 0 (above) opened chunks 0, 1 (and 2 for the prefetch)
            step(F0, F1);
            step(F1, F0);
            __builtin_amdgcn_s_barrier();
        }
    } else
    for (int g = 0; g < G; g += 2) {
        step(F0, F1);
        if (g + 1 < G) step(F1, F0);
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[r] + accb[r] + c4[r & 3][r >> 2];
    p.out[(size_t)blockIdx.x * 256 + (tid & 255)] = s;
}

template <int MODE, int LW, int NM = 16, int M16 = 0, int RD = 1, int SCHED = 0, int PCS = 16, int TWO = 0>
void run(const char* name, P p, int grid)
{
    hipFuncSetAttribute((const void*)ring<MODE, LW, NM, M16, RD, SCHED, PCS, TWO>, hipFuncAttributeMaxDynamicSharedMemorySize, NS * PCS * 1024);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((ring<MODE, LW, NM, M16, RD, SCHED, PCS, TWO>), dim3(grid), dim3(SCHED == 7 ? 768 : 512), NS * PCS * 1024, 0, p);
        hipDeviceSynchronize();
    }
    std::vector<long long> h(grid * 2);
    hipMemcpy(h.data(), p.res, grid * 16, hipMemcpyDeviceToHost);
    std::vector<double> cy, rt;
    for (int i = 0; i < grid; i++) cy.push_back((double)h[2 * i]), rt.push_back((double)h[2 * i + 1]);
    std::sort(cy.begin(), cy.end()), std::sort(rt.begin(), rt.end());
    const double c = cy[grid / 2], t = rt[grid / 2] * 10e-9;  // 100 MHz ticks
    const double bytes = (double)p.chunks * PCS * 1024, wgpc = grid / 256.0;
    printf("%-34s grid %4d lw %d mfma %d reads %d share %d | %7.0f cyc/chunk/WG  %5.1f B/clk/CU  %5.1f GB/s/CU  clk %.2f GHz  mfma-bound %4.0f%%\n", name,
           grid, LW, NM, RD, p.share, c / p.chunks, bytes / c * wgpc, bytes / t * wgpc / 1e9, c / t / 1e9, 64.0 * (NM ? NM : 16) * wgpc / (c / p.chunks) * 100);
}

int main()
{
    P p;
    const int kwin = 8;
    float *a, *b, *out;
    long long* res;
    hipMalloc(&a, (size_t)512 * 64 * kwin * 32 * 4 + (size_t)64 * kwin * 32 * 4);
    b = a + (size_t)512 * 64 * kwin * 32;  // inside the buffer resource of `a`
    std::vector<float> init((size_t)513 * 64 * kwin * 32);
    for (size_t i = 0; i < init.size(); i++) init[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(a, init.data(), init.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 512 * 256 * 4);
    hipMalloc(&res, 512 * 16);
    p.a = a, p.b = b, p.out = out, p.res = res, p.chunks = 510, p.kwin = kwin;
    p.share = 4, p.reads = 1, p.mfma = 16;
    p.two_acc = 0;
    auto cfg = [&](int lfirst, int cons, int lprio, int cprio) { p.lfirst = lfirst, p.cons_waves = cons, p.m16 = 0, p.lprio = lprio, p.cprio = cprio; };
    for (int grid : {200, 256}) {
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 1>("16 KiB stages (64x64 tile)", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 1, 24>("24 KiB stages (64x32 tile x 2 K groups)", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 0, 0, 1, 1, 24>("24 KiB stages, no MFMA", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 1, 32>("32 KiB stages (32x32 x 4 K groups / 64x64 x 2)", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 0>("16 KiB, barrier behind MFMA 8, reads behind 8-15", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 0, 0>("16 KiB, the same without the fragment reads", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 1, 16, 2>("16 KiB stages, accumulator in AGPRs (asm)", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 1, 24, 2>("24 KiB stages, accumulator in AGPRs (asm)", p, grid);
        cfg(0, 8, 1, 0); run<1, 4, 16, 0, 1, 7>("16 KiB stages, EIGHT consumer waves (8 MFMAs + 4 reads each)", p, grid);
        cfg(0, 8, 1, 0); run<1, 4, 16, 0, 1, 7, 24>("24 KiB stages, EIGHT consumer waves", p, grid);
        cfg(0, 8, 1, 0); run<1, 4, 16, 0, 1, 7, 20>("20 KiB stages, EIGHT consumer waves", p, grid);
        cfg(0, 4, 0, 0); run<1, 4, 16, 0, 1, 1>("16 KiB stages, no priorities", p, grid);
        cfg(0, 4, 0, 1); run<1, 4, 16, 0, 1, 1>("16 KiB stages, consumers at priority 3", p, grid);
        cfg(0, 4, 0, 1); run<1, 4, 16, 0, 1, 1, 24>("24 KiB stages, consumers at priority 3", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 6>("16 KiB stages, B fragments from L2 (4 LDS + 4 global reads)", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 6, 24>("24 KiB stages, B fragments from L2", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 1, 16, 1>("16 KiB stages, two accumulator chains", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 1, 24, 1>("24 KiB stages, two accumulator chains", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 4, 24, 1>("24 KiB, two chains, TWO chunks per barrier", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 4>("16 KiB stages, TWO chunks per barrier", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 4, 24>("24 KiB stages, TWO chunks per barrier", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 4, 20>("20 KiB stages, TWO chunks per barrier", p, grid);
        cfg(0, 4, 1, 0); run<1, 4, 16, 0, 1, 1, 20>("20 KiB stages (32x128 tile)", p, grid);
    }
    return 0;
}
