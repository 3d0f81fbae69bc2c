// Probe (round 6): is the ONE tile family tools/tile_enum.py says beats the plan in use -- 112 x 64 tiles of 16x16
// accumulators (v_mfma_f32_16x16x4_f32), 240 tiles = ONE round over the 256 CUs for the N = 1 024 launches of the 23x23 stage -- really
// faster than 400 tiles of 64 x 64 (two per CU on 144 CUs), once everything a tile costs is paid: cold start, ring fill, the step
// efficiency of ONE workgroup per CU, the 16x16 C layout's half-line (64-byte) stores and shortcut reads?
//
// A stand-alone, numerically CORRECT fp32 GEMM + epilogue of exactly the product's plain 1x1 launches (res4*_branch2c: M = 1 587,
// N = 1 024, K = 256; res5a_branch2c_new: K = 512):   out[m][n] = relu(sum_k A[m][k] W[n][k] + bias[n] + resid[m][n])
// written the way the product's conv_stream_kernel is (wave-specialised workgroup of 4 consumer + 4 producer waves, buffer-addressed
// LDS-DMA into a ring of stages, XOR-swizzled 16-byte units, counted vmcnt waits, one raw barrier per 32-float K chunk, write-through
// epilogue stores), in two shapes that share every line except the consumer loop and the epilogue:
//   T64  : 64 x 64 tiles, four 32x32 accumulators (v_mfma_f32_32x32x2_f32), ring 5 x 16 KiB, two workgroups per CU   -- the plan in use
//   T112 : 112 x 64 tiles, 4 waves x seven 16x16 accumulators (wave w: column block w, all seven row blocks), ring 5 x 22 KiB, one per CU
// Both are timed as the product times a launch: N back-to-back dependent launches on one stream between two events.  T64 calibrates the
// harness against the product's own figure for the layer (tools/layer_table.py in the same gpurun call).
//   hipcc --offload-arch=gfx950 -O3 -o tools/tile112_probe tools/tile112_probe.hip && tools/tile112_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) const float cgfloat;

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                             \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

struct GArgs {
    const float *A, *W, *bias, *resid;
    float* out;
    int M, N, K, tiles_m, tiles_n;
};

template <int N>
__device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory");
}
__device__ __forceinline__ void put_f32(gfloat* p, float v) { __hip_atomic_store((float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t srd_t;
__device__ __forceinline__ srd_t make_srd(const void* base) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ void bload_lds(srd_t r, float* lds, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, (int)soff, 0, 0);
}
#else
typedef int srd_t;
__device__ __forceinline__ srd_t make_srd(const void*) { return 0; }
__device__ __forceinline__ void bload_lds(srd_t, float*, unsigned, unsigned) {}
#endif

// BM = 64: 32x32x2 accumulators; BM = 112 (or any multiple of 16 up to 112): 16x16x4 accumulators
// EPI = false: timing variant without the epilogue's shortcut reads and stores (the accumulators stay live through a store that never
// happens): what the K loop + cold start cost alone
template <int BM, int NS, bool EPI>
__global__ __launch_bounds__(512, BM == 64 ? 4 : 1) void gemm_tile_kernel(const GArgs a)
{
    constexpr bool M16 = BM != 64;
    // stage image as the product's: A in 32-row blocks (the last one partly unused when BM is not a multiple of 32), then the 64 B rows;
    // producer wave w lands rows 8 w .. 8 w + 7 of EVERY 32-row block: ARB + 2 instructions per wave and chunk, all compile-time structure
    constexpr int BN = 64, ARB = (BM + 31) / 32, BROW0 = ARB * 32, ROWS = BROW0 + BN;
    constexpr int PER = ARB + 2;
    constexpr int STAGE = ROWS * 32;                                 // floats
    constexpr int RB = BM / 16;                                      // 16-row blocks (M16)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;
    // tile: N fastest, XCD-aware (workgroup ids are dealt round-robin over 8 XCDs: an XCD takes a contiguous eighth of the sequence)
    int item;
    {
        const int items = a.tiles_m * a.tiles_n, id = blockIdx.x, xcd = id & 7, l = id >> 3;
        const int qd = items >> 3, rm = items & 7;
        item = __builtin_amdgcn_readfirstlane((xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + l);
    }
    const int tile_m = item / a.tiles_n, tile_n = item - tile_m * a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int G = a.K >> 5;

    if (producer) {
        __builtin_amdgcn_s_setprio(3);
        const srd_t srdA = make_srd(a.A), srdB = make_srd(a.W);
        unsigned a_vo[ARB], b_vo[2];
        const int srow = tid >> 3;                               // row within a 32-row block (wave w: rows 8 w .. 8 w + 7)
        const int unit = (tid & 7) ^ ((tid >> 4) & 7);           // SOURCE unit for LDS slot (row 32 i + srow, unit tid & 7)
#pragma unroll
        for (int i = 0; i < ARB; i++) {
            const int row = 32 * i + srow;
            a_vo[i] = (row < BM && m0 + row < a.M) ? (unsigned)(((m0 + row) * a.K + unit * 4) * 4) : 0x80000000u;
        }
#pragma unroll
        for (int i = 0; i < 2; i++) b_vo[i] = (unsigned)(((n0 + 32 * i + srow) * a.K + unit * 4) * 4);
        int nis = 0, istage = 0;
        unsigned so = 0;
        auto put = [&]() __attribute__((always_inline)) {
            float* sb = smem + __builtin_amdgcn_readfirstlane(istage) * STAGE + wave * 256;   // the hardware adds lane * 16 B
            const unsigned u = (unsigned)__builtin_amdgcn_readfirstlane((int)so);
#pragma unroll
            for (int i = 0; i < ARB; i++) bload_lds(srdA, sb + i * 1024, a_vo[i], u);
#pragma unroll
            for (int i = 0; i < 2; i++) bload_lds(srdB, sb + BROW0 * 32 + i * 1024, b_vo[i], u);
            so += 128, istage = istage + 1 == NS ? 0 : istage + 1, nis++;
        };
        auto wait_landed = [&](int young) __attribute__((always_inline)) {
            switch (young) {
                case 1: wait_vm<PER>(); break;
                case 2: wait_vm<2 * PER>(); break;
                case 3: wait_vm<3 * PER>(); break;
                case 4: wait_vm<4 * PER>(); break;
                default: wait_vm<0>(); break;
            }
        };
        put();
        if (G > 1) put();
        wait_landed(nis - 1);
        __builtin_amdgcn_s_barrier();  // chunk 0 visible
        if (G > 2) put();
        for (int g = 0; g < G; g++) {
            const int young = nis - g - 2;
            wait_landed(young > 0 ? (young < NS - 3 ? young : NS - 3) : 0);
            __builtin_amdgcn_s_barrier();
            const int target = G < g + NS ? G : g + NS;
            while (nis < target) put();
        }
        return;
    }

    // ---- consumers ------------------------------------------------------------------------------------------------------------
    cgfloat* bias = (cgfloat*)a.bias;
    cgfloat* resid = (cgfloat*)a.resid;
    gfloat* out = (gfloat*)a.out;
    if constexpr (!M16) {
        const int wm = wave >> 1, wn = wave & 1;
        int fo[4];
#pragma unroll
        for (int q = 0; q < 4; q++) fo[q] = (lane & 31) * 32 + (((2 * q + (lane >> 5)) ^ ((lane >> 1) & 7)) * 4);
        struct Frag { f32x4 a[4], b[4]; } F0, F1;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        auto rd = [&](Frag& F, int stage) __attribute__((always_inline)) {
            const float* Ab = smem + stage * STAGE + (wm * 32) * 32;
            const float* Bb = smem + stage * STAGE + (BROW0 + wn * 32) * 32;
#pragma unroll
            for (int q = 0; q < 4; q++) F.a[q] = *(const f32x4*)(Ab + fo[q]), F.b[q] = *(const f32x4*)(Bb + fo[q]);
        };
        // D[m][n]: n = lane & 31, m = 8 (i >> 2) + 4 (lane >> 5) + (i & 3).  Bias and shortcut of the tile are requested NOW (as the product
        // does), so the epilogue never waits for them
        const int n = n0 + wn * 32 + (lane & 31);
        const float bv = bias[n];
        float rs[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int m = m0 + wm * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
            rs[i] = (EPI && m < a.M) ? resid[(unsigned)(m * a.N + n)] : 0.f;
        }
        int stage = 0;
        __builtin_amdgcn_s_barrier();  // chunk 0 visible
        rd(F0, 0);
        auto step = [&](Frag& cur, Frag& nxt, bool more) __attribute__((always_inline)) {
            const int nstage = stage + 1 == NS ? 0 : stage + 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const float* Ab = smem + nstage * STAGE + (wm * 32) * 32;
            const float* Bb = smem + nstage * STAGE + (BROW0 + wn * 32) * 32;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int i = q * 4 + e;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[q][e], cur.b[q][e], acc, 0, 0, 0);
                    if ((i & 1) && more) {
                        const int r = i >> 1;
                        if (r & 1) nxt.b[r >> 1] = *(const f32x4*)(Bb + fo[r >> 1]);
                        else nxt.a[r >> 1] = *(const f32x4*)(Ab + fo[r >> 1]);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            stage = nstage;
        };
        for (int g = 0; g < G; g += 2) {
            step(F0, F1, g + 1 < G);
            if (g + 1 < G) step(F1, F0, g + 2 < G);
        }
        // epilogue from registers: bias and shortcut were requested before the K loop
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int m = m0 + wm * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
            float o = acc[i] + bv + rs[i];
            o = __builtin_fmaxf(o, 0.f);
            if (m < a.M && (EPI || __builtin_bit_cast(unsigned, o) == 0x7fc12345u)) put_f32(out + (unsigned)(m * a.N + n), o);
        }
    } else {
        // wave w: column block w (16 columns), row blocks 0 .. RB-1.  Fragment of a 16-row block for half-chunk j: lane (r = lane & 15,
        // kq = lane >> 4) reads logical unit kq + 4 j of row r: floats 4 (kq + 4 j) .. + 3; MFMA e of that half uses float e, i.e. the four
        // lanes of a row supply k = 4 (kq + 4 j) + e: every k of the chunk exactly once, A and B alike.
        int foA[RB][2], foB[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int u = (lane >> 4) + 4 * j;
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                const int row = rb * 16 + (lane & 15);
                foA[rb][j] = row * 32 + ((u ^ ((row >> 1) & 7)) * 4);
            }
            const int rowB = BROW0 + wave * 16 + (lane & 15);
            foB[j] = rowB * 32 + ((u ^ ((rowB >> 1) & 7)) * 4);
        }
        struct Frag { f32x4 a[RB][2], b[2]; } F0, F1;
        f32x4 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; rb++) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        // D[m][n]: n = lane & 15, m = 4 (lane >> 4) + i; bias and shortcut requested before the K loop
        const int n = n0 + wave * 16 + (lane & 15);
        const float bv = bias[n];
        float rs[RB][4];
#pragma unroll
        for (int rb = 0; rb < RB; rb++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int m = m0 + rb * 16 + 4 * (lane >> 4) + i;
                rs[rb][i] = (EPI && m < a.M) ? resid[(unsigned)(m * a.N + n)] : 0.f;
            }
        int stage = 0;
        __builtin_amdgcn_s_barrier();  // chunk 0 visible
        {
            const float* S = smem;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                F0.b[j] = *(const f32x4*)(S + foB[j]);
#pragma unroll
                for (int rb = 0; rb < RB; rb++) F0.a[rb][j] = *(const f32x4*)(S + foA[rb][j]);
            }
        }
        auto step = [&](Frag& cur, Frag& nxt, bool more) __attribute__((always_inline)) {
            const int nstage = stage + 1 == NS ? 0 : stage + 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const float* S = smem + nstage * STAGE;
            // 8 RB MFMAs (32 cycles each), the 2 RB + 2 fragment reads of the next chunk behind every third one (B of a half first: every
            // MFMA of the half needs it).  Everything is indexed by the unrolled loops' constants: the fragments stay in registers.
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int rb = 0; rb < RB; rb++) {
                        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[rb][j][e], cur.b[j][e], acc[rb], 0, 0, 0);
                        const int i = (j * 4 + e) * RB + rb, r = (i - 1) / 3;
                        if (i % 3 == 1 && r < 2 * RB + 2) {
                            if (more) {
                                const int jj = r / (RB + 1), k = r % (RB + 1);
                                if (k == 0) nxt.b[jj] = *(const f32x4*)(S + foB[jj]);
                                else nxt.a[k - 1][jj] = *(const f32x4*)(S + foA[k - 1][jj]);
                            }
                            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                    }
            static_assert((8 * RB - 2) / 3 + 1 >= 2 * RB + 2, "enough read slots behind the MFMAs");
            __builtin_amdgcn_sched_barrier(0);
            stage = nstage;
        };
        for (int g = 0; g < G; g += 2) {
            step(F0, F1, g + 1 < G);
            if (g + 1 < G) step(F1, F0, g + 2 < G);
        }
        // epilogue from registers (16 lanes write 64 contiguous bytes of a row)
#pragma unroll
        for (int rb = 0; rb < RB; rb++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int m = m0 + rb * 16 + 4 * (lane >> 4) + i;
                float o = acc[rb][i] + bv + rs[rb][i];
                o = __builtin_fmaxf(o, 0.f);
                if (m < a.M && (EPI || __builtin_bit_cast(unsigned, o) == 0x7fc12345u)) put_f32(out + (unsigned)(m * a.N + n), o);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Shape P (round 6, second question): 112 x 16 tiles, FOUR in-workgroup K groups -- for the launches of the 23x23 stage whose N is small
// and whose K is long (res4*_branch2a: N = 256, K = 1 024; res4*_branch2b: N = 256, K = 2 304), which run as 200 tiles of 64 x 32 x 2 on
// 256 CUs (0.78 of the chip) today.  15 x 16 = 240 tiles, one round, 1 792 outputs per CU instead of 2 048.  Their epilogue moves 1.6 MB
// (no shortcut), so the 16x16 layout's half-line stores -- what killed 112 x 64 on the N = 1 024 launches -- are small change here.
// Every consumer wave holds seven 16x16 accumulators (all seven row blocks of the one column block) and takes a QUARTER of every
// chunk's K: wave w reads the 16-byte units 2 w, 2 w + 1 of each row (ds_read_b64: lane (r, kq) gets k = 8 w + 2 kq + {0, 1}), 14 MFMAs and
// 8 reads per chunk; after the K loop the four partial sums meet in LDS.  out[m][n] = relu(sum_k A[m][k] W[n][k] + bias[n]).
template <int NS, bool EPI>
__global__ __launch_bounds__(512, 1) void gemm_p_kernel(const GArgs a)
{
    constexpr int BM = 112, RB = 7, ARB = 4, BROW0 = ARB * 32, ROWS = BROW0 + 32, PER = ARB + 1, STAGE = ROWS * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;
    int item;
    {
        const int items = a.tiles_m * a.tiles_n, id = blockIdx.x, xcd = id & 7, l = id >> 3;
        const int qd = items >> 3, rm = items & 7;
        item = __builtin_amdgcn_readfirstlane((xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + l);
    }
    const int tile_m = item / a.tiles_n, tile_n = item - tile_m * a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * 16;
    const int G = a.K >> 5;
    if (producer) {
        __builtin_amdgcn_s_setprio(3);
        const srd_t srdA = make_srd(a.A), srdB = make_srd(a.W);
        unsigned a_vo[ARB], b_vo;
        const int srow = tid >> 3;
        const int unit = (tid & 7) ^ ((tid >> 4) & 7);
#pragma unroll
        for (int i = 0; i < ARB; i++) {
            const int row = 32 * i + srow;
            a_vo[i] = (row < BM && m0 + row < a.M) ? (unsigned)(((m0 + row) * a.K + unit * 4) * 4) : 0x80000000u;
        }
        b_vo = srow < 16 ? (unsigned)(((n0 + srow) * a.K + unit * 4) * 4) : 0x80000000u;   // rows 16 .. 31 of the B block: nothing fetched
        int nis = 0, istage = 0;
        unsigned so = 0;
        auto put = [&]() __attribute__((always_inline)) {
            float* sb = smem + __builtin_amdgcn_readfirstlane(istage) * STAGE + wave * 256;
            const unsigned u = (unsigned)__builtin_amdgcn_readfirstlane((int)so);
#pragma unroll
            for (int i = 0; i < ARB; i++) bload_lds(srdA, sb + i * 1024, a_vo[i], u);
            bload_lds(srdB, sb + BROW0 * 32, b_vo, u);
            so += 128, istage = istage + 1 == NS ? 0 : istage + 1, nis++;
        };
        auto wait_landed = [&](int young) __attribute__((always_inline)) {
            switch (young) {
                case 1: wait_vm<PER>(); break;
                case 2: wait_vm<2 * PER>(); break;
                case 3: wait_vm<3 * PER>(); break;
                case 4: wait_vm<4 * PER>(); break;
                default: wait_vm<0>(); break;
            }
        };
        put();
        if (G > 1) put();
        wait_landed(nis - 1);
        __builtin_amdgcn_s_barrier();
        if (G > 2) put();
        for (int g = 0; g < G; g++) {
            const int young = nis - g - 2;
            wait_landed(young > 0 ? (young < NS - 3 ? young : NS - 3) : 0);
            __builtin_amdgcn_s_barrier();
            const int target = G < g + NS ? G : g + NS;
            while (nis < target) put();
        }
        // the producers stay for the two barriers of the consumers' reduction (a wave that has ended no longer counts, but being there is simpler to reason about)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        return;
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    cgfloat* bias = (cgfloat*)a.bias;
    gfloat* out = (gfloat*)a.out;
    int foA[RB], foB;
    {
        const int kq = lane >> 4, u = 2 * wave + (kq >> 1), half = kq & 1;
#pragma unroll
        for (int rb = 0; rb < RB; rb++) {
            const int row = rb * 16 + (lane & 15);
            foA[rb] = row * 32 + ((u ^ ((row >> 1) & 7)) * 4) + half * 2;
        }
        const int rowB = BROW0 + (lane & 15);
        foB = rowB * 32 + ((u ^ ((rowB >> 1) & 7)) * 4) + half * 2;
    }
    struct Frag { f32x2 a[RB], b; } F0, F1;
    f32x4 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; rb++) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int n = n0 + (lane & 15);
    const float bv = bias[n];
    int stage = 0;
    __builtin_amdgcn_s_barrier();  // chunk 0 visible
    F0.b = *(const f32x2*)(smem + foB);
#pragma unroll
    for (int rb = 0; rb < RB; rb++) F0.a[rb] = *(const f32x2*)(smem + foA[rb]);
    auto step = [&](Frag& cur, Frag& nxt, bool more) __attribute__((always_inline)) {
        const int nstage = stage + 1 == NS ? 0 : stage + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const float* S = smem + nstage * STAGE;
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[rb][e], cur.b[e], acc[rb], 0, 0, 0);
                const int i = e * RB + rb;   // 0 .. 13: the 8 reads of the next chunk behind MFMAs 0, 1, 3, 4, 6, 7, 9, 10
                if (i % 3 != 2 && (i - i / 3) < RB + 1) {
                    const int r = i - i / 3;
                    if (more) {
                        if (r == 0) nxt.b = *(const f32x2*)(S + foB);
                        else nxt.a[r - 1] = *(const f32x2*)(S + foA[r - 1]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        stage = nstage;
    };
    for (int g = 0; g < G; g += 2) {
        step(F0, F1, g + 1 < G);
        if (g + 1 < G) step(F1, F0, g + 2 < G);
    }
    // the four K quarters meet in LDS (the ring is idle: every chunk has been consumed).  Wave w finishes row blocks w and w + 4 (w < 3) /
    // block 3 (w = 3): every wave stores the partials of the blocks it does NOT finish, lane-linear 16-byte slots, then adds the three others'
    float* part = smem;   // [writer wave][block][lane] x 16 B = 4 x 7 x 1 KiB
    __builtin_amdgcn_s_barrier();   // (all consumers are past their last fragment read)
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
        if ((rb & 3) != wave) *(f32x4*)(part + ((wave * RB + rb) * 64 + lane) * 4) = acc[rb];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int rb = 0; rb < RB; rb++) {
        if ((rb & 3) != wave) continue;   // (wave-uniform)
        f32x4 sum = acc[rb];
        // fixed order: K quarters 0, 1, 2, 3 (the own one in its place), so the result does not depend on which wave finishes the block
        f32x4 q[4];
#pragma unroll
        for (int w = 0; w < 4; w++) q[w] = w == wave ? sum : *(const f32x4*)(part + ((w * RB + rb) * 64 + lane) * 4);
        sum = ((q[0] + q[1]) + q[2]) + q[3];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int m = m0 + rb * 16 + 4 * (lane >> 4) + i;
            float o = __builtin_fmaxf(sum[i] + bv, 0.f);
            if (m < a.M && (EPI || __builtin_bit_cast(unsigned, o) == 0x7fc12345u)) put_f32(out + (unsigned)(m * a.N + n), o);
        }
    }
}

template <int NS, bool EPI = true>
static double run_p(const char* name, GArgs a, int reps, std::vector<float>& hout, const std::vector<float>& ref, const std::vector<int>& samples)
{
    const size_t lds = (size_t)NS * (128 + 32) * 128;
    auto k = gemm_p_kernel<NS, EPI>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void*)k));
    a.tiles_m = (a.M + 111) / 112, a.tiles_n = a.N / 16;
    const int grid = a.tiles_m * a.tiles_n;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipMemsetAsync(a.out, 0, (size_t)a.M * a.N * 4, st));
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a);
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    CK(hipMemcpy(hout.data(), a.out, (size_t)a.M * a.N * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    if (EPI)
        for (size_t s = 0; s < samples.size(); s++) maxerr = std::max(maxerr, (double)fabsf(hout[samples[s]] - ref[s])), maxref = std::max(maxref, (double)fabsf(ref[s]));
    std::vector<double> us;
    for (int rep = 0; rep < 7; rep++) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us.push_back(ms * 1e3 / reps);
    }
    std::sort(us.begin(), us.end());
    const double flops = 2.0 * a.M * a.N * a.K;
    printf("  %-5s 112 x 16 x 4 K groups: %4d workgroups (%.2f per CU), %3d VGPRs, %6zu B LDS: %6.2f us per launch (median of 7 x %d back-to-back; min %.2f max %.2f), "
           "%5.1f TFLOP/s, max |err| %.2e of max |ref| %.2f\n",
           name, grid, grid / 256.0, fa.numRegs, lds, us[3], reps, us[0], us[6], flops / (us[3] * 1e-6) / 1e12, maxerr, maxref);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    CK(hipStreamDestroy(st));
    return us[3];
}

template <int BM, int NS, bool EPI = true>
static double run(const char* name, GArgs a, int reps, std::vector<float>& hout, const std::vector<float>& ref, const std::vector<int>& samples)
{
    constexpr int ROWS = (BM + 31) / 32 * 32 + 64;
    const size_t lds = (size_t)NS * ROWS * 128;
    auto k = gemm_tile_kernel<BM, NS, EPI>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void*)k));
    a.tiles_m = (a.M + BM - 1) / BM, a.tiles_n = a.N / 64;
    const int grid = a.tiles_m * a.tiles_n;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipMemsetAsync(a.out, 0, (size_t)a.M * a.N * 4, st));
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a);
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    CK(hipMemcpy(hout.data(), a.out, (size_t)a.M * a.N * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    if (EPI)
        for (size_t s = 0; s < samples.size(); s++) maxerr = std::max(maxerr, (double)fabsf(hout[samples[s]] - ref[s])), maxref = std::max(maxref, (double)fabsf(ref[s]));
    std::vector<double> us;
    for (int rep = 0; rep < 7; rep++) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us.push_back(ms * 1e3 / reps);
    }
    std::sort(us.begin(), us.end());
    const double flops = 2.0 * a.M * a.N * a.K;
    printf("  %-5s %3d x 64 tiles: %4d workgroups (%.2f per CU), %3d VGPRs, %6zu B LDS: %6.2f us per launch (median of 7 x %d back-to-back; min %.2f max %.2f), "
           "%5.1f TFLOP/s, max |err| %.2e of max |ref| %.2f\n",
           name, BM, grid, grid / 256.0, fa.numRegs, lds, us[3], reps, us[0], us[6], flops / (us[3] * 1e-6) / 1e12, maxerr, maxref);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    CK(hipStreamDestroy(st));
    return us[3];
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    const int M = 1587, N = 1024;
    for (int K : {256, 512}) {
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N), hr((size_t)M * N), hout((size_t)M * N);
        unsigned s = 12345u + K;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : hA) v = rnd();
        for (auto& v : hW) v = rnd() * 0.1f;
        for (auto& v : hb) v = rnd();
        for (auto& v : hr) v = rnd();
        std::vector<int> samples;
        std::vector<float> ref;
        for (int i = 0; i < 4000; i++) {
            s = s * 1664525u + 1013904223u;
            const int m = i < 64 ? M - 1 - i : (int)((s >> 8) % M);   // the last rows (the partial tile) always
            s = s * 1664525u + 1013904223u;
            const int n = (int)((s >> 8) % N);
            double acc = 0;
            for (int k = 0; k < K; k++) acc += (double)hA[(size_t)m * K + k] * hW[(size_t)n * K + k];
            acc += hb[n] + hr[(size_t)m * N + n];
            samples.push_back(m * N + n), ref.push_back((float)std::max(acc, 0.0));
        }
        GArgs a{};
        float *dA, *dW, *db, *dr, *dout;
        CK(hipMalloc(&dA, hA.size() * 4 + 65536));
        CK(hipMalloc(&dW, hW.size() * 4));
        CK(hipMalloc(&db, N * 4));
        CK(hipMalloc(&dr, hr.size() * 4));
        CK(hipMalloc(&dout, hr.size() * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
        a.A = dA, a.W = dW, a.bias = db, a.resid = dr, a.out = dout, a.M = M, a.N = N, a.K = K;
        printf("M = %d, N = %d, K = %d (%s): ideal at 157.3 TFLOP/s %.2f us\n", M, N, K, K == 256 ? "res4*_branch2c" : "res5a_branch2c_new", 2.0 * M * N * K / 157.3e12 * 1e6);
        double t64 = 0, t112 = 0;
        for (int round = 0; round < 2; round++) {   // interleaved, twice
            t64 = run<64, 5>("T64", a, reps, hout, ref, samples);
            t112 = run<112, 5>("T112", a, reps, hout, ref, samples);
            run<96, 5>("T96", a, reps, hout, ref, samples);
            run<80, 5>("T80", a, reps, hout, ref, samples);
        }
        printf("  112 x 64 against 64 x 64: %+.2f us per launch\n", t112 - t64);
        printf("  without the epilogue (no shortcut reads, no stores; err column void):\n");
        const double k64 = run<64, 5, false>("T64-", a, reps, hout, ref, samples);
        const double k112 = run<112, 5, false>("T112-", a, reps, hout, ref, samples);
        printf("  K loop + cold start: 64 x 64 %.2f us, 112 x 64 %.2f us (%+.2f); epilogue: 64 x 64 %.2f us, 112 x 64 %.2f us (%+.2f)\n", k64, k112, k112 - k64,
               t64 - k64, t112 - k112, (t112 - k112) - (t64 - k64));
        CK(hipFree(dA));
        CK(hipFree(dW));
        CK(hipFree(db));
        CK(hipFree(dr));
        CK(hipFree(dout));
    }
    // ---- second question: the small-N, long-K 1x1 launches (res4*_branch2a: M = 1 587, N = 256, K = 1 024; no shortcut) as 112 x 16 x 4 ----
    for (int K : {1024, 2304}) {   // 2 304 = res4*_branch2b's K (3x3 x 256) as a plain GEMM: its K loop without the tap gather
        const int N2 = 256;
        std::vector<float> hA((size_t)M * K), hW((size_t)N2 * K), hb(N2), hr((size_t)M * N2, 0.f), hout((size_t)M * N2);
        unsigned s = 777u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : hA) v = rnd();
        for (auto& v : hW) v = rnd() * 0.1f;
        for (auto& v : hb) v = rnd();
        std::vector<int> samples;
        std::vector<float> ref;
        for (int i = 0; i < 4000; i++) {
            s = s * 1664525u + 1013904223u;
            const int m = i < 64 ? M - 1 - i : (int)((s >> 8) % M);
            s = s * 1664525u + 1013904223u;
            const int n = (int)((s >> 8) % N2);
            double acc = 0;
            for (int k = 0; k < K; k++) acc += (double)hA[(size_t)m * K + k] * hW[(size_t)n * K + k];
            acc += hb[n];
            samples.push_back(m * N2 + n), ref.push_back((float)std::max(acc, 0.0));
        }
        GArgs a{};
        float *dA, *dW, *db, *dr, *dout;
        CK(hipMalloc(&dA, hA.size() * 4 + 65536));
        CK(hipMalloc(&dW, hW.size() * 4));
        CK(hipMalloc(&db, N2 * 4));
        CK(hipMalloc(&dr, hr.size() * 4));
        CK(hipMalloc(&dout, hr.size() * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), N2 * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dr, 0, hr.size() * 4));
        a.A = dA, a.W = dW, a.bias = db, a.resid = dr, a.out = dout, a.M = M, a.N = N2, a.K = K;
        printf("M = %d, N = %d, K = %d (%s; the product runs it as 200 tiles of 64 x 32 x 2: %s of execution + the 2.85-us boundary per back-to-back launch): ideal %.2f us\n",
               M, N2, K, K == 1024 ? "res4*_branch2a" : "res4*_branch2b's K loop", K == 1024 ? "9.5 us" : "19.3 us", 2.0 * M * N2 * K / 157.3e12 * 1e6);
        for (int round = 0; round < 2; round++) {
            run<64, 5>("T64", a, reps, hout, ref, samples);    // (100 tiles of 64 x 64: the harness's yardstick, not a plan anybody would use)
            run_p<5>("P", a, reps, hout, ref, samples);
        }
        run_p<5, false>("P-", a, reps, hout, ref, samples);
        CK(hipFree(dA));
        CK(hipFree(dW));
        CK(hipFree(db));
        CK(hipFree(dr));
        CK(hipFree(dout));
    }
    return 0;
}
