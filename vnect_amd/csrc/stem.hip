// stem.hip -- the network's stem as ONE launch on spatial tiles: conv1 (7x7, stride 2, SAME, 3 -> 64, bias, ReLU) and the 3x3 /
// stride-2 max-pool behind it (/root/reference/src/vnect_model.py:27-29), optionally with gen_input_batch in front
// (src/estimator.py:70-81: the input patch is then computed from the uint8 frame, pyramid.h, and the (S,368,368) batch never exists).
//
// Replaces, bit for bit, conv_stream_kernel<..., SPAN> (fp32) / <..., BF> (bf16) on conv1 + maxpool_kernel (+ pyramid_kernel): conv1's
// 26 MB output is never written and re-read, two (three) launches and their boundaries go.
//
// Tile = up to 5 pooled rows x 23 pooled columns of one image (92 = 4 x 23; the host cuts an image's 92 rows into groups of 4 and
// 5 so that 3 images give 252 tiles: one per CU, one round).  A workgroup (8 waves, one per CU)
//   1. lands the tile's input patch in LDS: conv rows 2 r0 .. 2 r1 and conv columns 46 c .. 46 c + 46 (one more than 2 x the pooled
//      extent: the pooling windows overlap by one) need input rows 4 r0 - 2 .. and 99 input columns, NHWC4 like the batch tensor,
//      zeros outside the image (the conv's SAME padding);
//   2. runs the conv as an implicit GEMM on the matrix cores, M = the tile's conv pixels in row-major order (32 per block), N = 2 x 32
//      channels, K exactly as conv1's stand-alone kernel walks it -- fp32: 7 filter rows x 4 pixel pairs x 3 channels of
//      v_mfma_f32_32x32x2_f32 (the fourth NHWC4 channel has zero weights and is skipped there too); bf16: 4 row pairs x 4
//      v_mfma_f32_32x32x16_bf16 -- so every output sees the same products in the same order and is bit-identical.  The A fragments
//      are read straight out of the patch (a lane's row is a conv pixel, its fragment the 16 bytes at input pixel (2 y + ky,
//      2 x + 2 q + h)); the B fragments (the 57 KB of weights) never touch LDS: a wave owns one 32-channel half for the whole tile
//      and keeps its 84 (64) fragment registers loaded once from L2;
//   3. pools without staging the conv tile (132 KB in fp32): every relu(acc + bias) goes straight from the accumulator registers
//      into the up to four pooling windows that contain it with ds_max_f32 on a 29 KB pooled tile (zero-initialised: ReLU outputs are
//      >= 0; max is exact and order-free, so this is bit-identical to maxpool_kernel; TF's SAME pool pads 0 before / 1 after for
//      184 -> 92 and padding never wins);
//   4. writes the pooled tile with 16-byte write-through stores.
#include "kernels.h"
#include "pyramid.h"

#include <type_traits>

namespace vnect {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int STEM_CW = 2 * STEM_TW + 1;             // conv columns per tile (47)
constexpr int STEM_PW = 100;                         // patch row stride in pixels (99 used; even, so a bf16 pixel pair is 16-byte aligned)
constexpr int STEM_PH = 2 * (2 * STEM_MAXH + 1) + 6; // patch rows at most: 27, + 1 for the bf16 form's zero-weight eighth filter row
constexpr int STEM_THREADS = 512;

template <bool BF>
constexpr size_t stem_lds() { return (size_t)STEM_PH * STEM_PW * 4 * (BF ? 2 : 4) + (size_t)STEM_MAXH * STEM_TW * 64 * 4; }

// FRAME: the patch is computed from the uint8 frame (pyramid.h) instead of being copied from the batch tensor
template <bool BF, bool FRAME, bool PROF>
__global__ __launch_bounds__(STEM_THREADS, 2) void stem_kernel(const StemArgs a)
{
    typedef typename std::conditional<BF, __bf16, float>::type T;
    typedef T tx4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    T* patch = (T*)smem;                                                     // [PH][PW][4]
    float* pooled = smem + STEM_PH * STEM_PW * 4 * sizeof(T) / sizeof(float);  // [h][23][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (PROF && tid == 0 && blockIdx.x == 0) a.prof[0] = (unsigned long long)__builtin_amdgcn_s_memrealtime();

    // ---- tile geometry (uniform) ---------------------------------------------------------------------------------------
    const int tile = blockIdx.x, c = tile & 3, gi = tile >> 2;
    const int sI = gi / a.groups, g = gi - sI * a.groups;
    const int r0 = a.row0[g], r1 = a.row0[g + 1], h = r1 - r0;     // pooled rows [r0, r1) of image sI
    const int cy0 = 2 * r0, cx0 = 2 * STEM_TW * c;                  // first conv row / column of the tile
    const int nrow = 2 * h + 1;                                    // conv rows 2 r0 .. 2 r1
    const int npix = nrow * STEM_CW, nblk = (npix + 31) >> 5;
    // patch rows: input rows 2 cy0 - 2 .. 2 (cy0 + nrow - 1) + 4; the bf16 form walks K in row PAIRS, so its (zero-weight) eighth
    // filter row reads one row more -- filled with the real input like every other, so that even the signs of the zero products
    // match the stand-alone kernel's.  Same for the zero-weight eighth pixel of a row (patch column 99).
    const int prow = 2 * nrow + (BF ? 6 : 5);
    const int iy0 = 2 * cy0 - 2, ix0 = 2 * cx0 - 2;

    // ---- this wave's weights: channel half wn, all of K, in the MFMA's B-fragment layout, requested before anything else ---------
    const int wn = wave & 1, mg = wave >> 1;
    const int hh = lane >> 5, nrowB = wn * 32 + (lane & 31);
    f32x3 Bf[BF ? 1 : 7][4];         // fp32: [ky][q] = weights of pixel 2 q + h, channels 0..2
    f32x4 Bb[BF ? 4 : 1][4];         // bf16: [t][q] = 8 bf16 of unit 2 q + h
    {
        typedef __attribute__((address_space(1))) const f32x4 cgf4;
        typedef __attribute__((address_space(1))) const f32x3 cgf3;
        if constexpr (BF) {
            const __bf16* wp = (const __bf16*)a.w + nrowB * 256;
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int q = 0; q < 4; q++) Bb[t][q] = *(cgf4*)(wp + t * 64 + (2 * q + hh) * 8);
        } else {
            const float* wp = a.w + nrowB * 224;
#pragma unroll
            for (int ky = 0; ky < 7; ky++)
#pragma unroll
                for (int q = 0; q < 4; q++) Bf[ky][q] = *(cgf3*)(wp + ky * 32 + (2 * q + hh) * 4);
        }
    }
    const float bias = a.bias[nrowB];

    // ---- 1. the input patch --------------------------------------------------------------------------------------------
    for (int i = tid; i < h * STEM_TW * 16; i += STEM_THREADS) ((f32x4*)pooled)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (FRAME) {
        const int s = sI + a.scale_base;
        for (int i = tid; i < prow * STEM_PW; i += STEM_THREADS) {
            const int pr = i / STEM_PW, pc = i - pr * STEM_PW;
            const int y = iy0 + pr, x = ix0 + pc;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)y < (unsigned)BOX && (unsigned)x < (unsigned)BOX) {
                int v[3];
                pyramid_pixel(a.fp, a.dyn, a.tabs, s, y, x, v);
                o = f32x4{a.tabs->lut[v[0]], a.tabs->lut[v[1]], a.tabs->lut[v[2]], 0.f};
            }
            ((tx4*)patch)[i] = __builtin_convertvector(o, tx4);
        }
    } else {
        const tx4* img = (const tx4*)a.batch + (long long)sI * BOX * BOX;
        constexpr int PER = (STEM_PH * STEM_PW + STEM_THREADS - 1) / STEM_THREADS;  // 6: every load of the patch in flight at once
        tx4 v[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = tid + k * STEM_THREADS;
            const int pr = i / STEM_PW, pc = i - pr * STEM_PW;
            const int y = iy0 + pr, x = ix0 + pc;
            v[k] = tx4{(T)0.f, (T)0.f, (T)0.f, (T)0.f};
            if (i < prow * STEM_PW && (unsigned)y < (unsigned)BOX && (unsigned)x < (unsigned)BOX) v[k] = img[y * BOX + x];
        }
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = tid + k * STEM_THREADS;
            if (i < prow * STEM_PW) ((tx4*)patch)[i] = v[k];
        }
    }
    __syncthreads();

    // ---- 2. + 3. conv blocks mg, mg + 4, ... of channel half wn; pooling straight from the accumulators -------------------------
    const int col = lane & 31, n = wn * 32 + col;
    for (int b = mg; b < nblk; b += 4) {
        // A fragments: this lane's row is conv pixel p = 32 b + (lane & 31) of the tile (rows past the tile read pixel 0: finite values,
        // results unused)
        int p = 32 * b + col;
        if (p >= npix) p = 0;
        const int cy = p / STEM_CW, cx = p - cy * STEM_CW;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        // K steps (chunk, q) in the stand-alone kernel's order; the fragment of step k + 1 is requested before the MFMAs of step k
        // (left to itself the compiler reads, waits, multiplies: an LDS round trip in front of every MFMA group)
        if constexpr (BF) {
            // unit u = 2 q + h of row pair t: input row 2 cy + 2 t + (u >> 2), pixels 2 cx + 2 (u & 3), + 1 (8 bytes each)
            const __bf16* ab = (const __bf16*)patch + ((2 * cy) * STEM_PW + 2 * cx) * 4;
            auto rd = [&](int k) __attribute__((always_inline)) {
                const int t = k >> 2, u = 2 * (k & 3) + hh;
                return *(const f32x4*)(ab + ((2 * t + (u >> 2)) * STEM_PW + 2 * (u & 3)) * 4);
            };
            f32x4 cur = rd(0), nxt = cur;
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // (the read of step 0)
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k + 1 < 16) nxt = rd(k + 1);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur), __builtin_bit_cast(bf16x8, Bb[k >> 2][k & 3]), acc, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one DS read ...
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // ... then this step's MFMA
                cur = nxt;
            }
        } else {
            const float* ab = (const float*)patch + ((2 * cy) * STEM_PW + 2 * cx + hh) * 4;
            auto rd = [&](int k) __attribute__((always_inline)) { return *(const f32x3*)(ab + ((k >> 2) * STEM_PW + 2 * (k & 3)) * 4); };
            f32x3 cur = rd(0), nxt = cur;
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // (the read of step 0)
#pragma unroll
            for (int k = 0; k < 28; k++) {
                if (k + 1 < 28) nxt = rd(k + 1);
#pragma unroll
                for (int e = 0; e < 3; e++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[e], Bf[k >> 2][k & 3][e], acc, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one DS read ...
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);  // ... then this step's three MFMAs
                cur = nxt;
            }
        }
        // C/D map: column = lane & 31 (channel n), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): the 16 conv pixels of this lane
        const int pbase = 32 * b + 4 * hh;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int pp = pbase + (r & 3) + 8 * (r >> 2);
            const int y = pp / STEM_CW, x = pp - y * STEM_CW;  // tile-relative conv pixel (uniform over the 32 lanes of a half)
            float v = __builtin_fmaxf(acc[r] + bias, 0.f);
            if constexpr (BF) v = (float)(__bf16)v;           // what the stand-alone conv1 stores (round to nearest even), then pools
            // inside the tile and inside the 184 x 184 conv output?  (conv row / column 184 is the pool's padding)
            if (pp < npix && cy0 + y < 2 * 92 && cx0 + x < 2 * 92) {
                // pooling windows that contain conv row y: pooled row y >> 1 (if the tile owns it) and, for even y >= 2, the one above
                const int pa = y >> 1, pb = x >> 1;
                const bool a0 = pa < h, a1 = !(y & 1) && y >= 2, b0 = pb < STEM_TW, b1 = !(x & 1) && x >= 2;
                float* cell = pooled + (pa * STEM_TW + pb) * 64 + n;
                if (a0 && b0) __hip_atomic_fetch_max(cell, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (a0 && b1) __hip_atomic_fetch_max(cell - 64, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (a1 && b0) __hip_atomic_fetch_max(cell - STEM_TW * 64, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (a1 && b1) __hip_atomic_fetch_max(cell - STEM_TW * 64 - 64, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();

    // ---- 4. the pooled tile: rows r0 .. r1 - 1, columns 23 c .. 23 c + 22 of image sI, 64 channels (16 units of 4) ---------------
    for (int i = tid; i < h * STEM_TW * 16; i += STEM_THREADS) {
        const int cell = i >> 4, u = i & 15;
        const int pr = cell / STEM_TW, pc = cell - pr * STEM_TW;
        const f32x4 v = ((const f32x4*)pooled)[i];
        T* dst = (T*)a.out + (((long long)sI * 92 + r0 + pr) * 92 + STEM_TW * c + pc) * 64 + u * 4;
        store_wt((tx4*)dst, __builtin_convertvector(v, tx4));
    }
    if (PROF && tid == 0 && blockIdx.x < PROF_WGS) a.prof_end[blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
}

hipError_t stem_setup()
{
    hipError_t e;
#define STEM_ATTR(BF, FR, PR)                                                                                                        \
    if ((e = hipFuncSetAttribute((const void*)stem_kernel<BF, FR, PR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stem_lds<BF>())) != \
        hipSuccess)                                                                                                                  \
        return e;
    STEM_ATTR(false, false, false) STEM_ATTR(false, false, true) STEM_ATTR(false, true, false) STEM_ATTR(false, true, true)
    STEM_ATTR(true, false, false) STEM_ATTR(true, false, true) STEM_ATTR(true, true, false) STEM_ATTR(true, true, true)
#undef STEM_ATTR
    return hipSuccess;
}

hipError_t launch_stem(const StemArgs& a, hipStream_t st)
{
    // host-side shape checks: the kernel's indexing assumes exactly these
    if (a.S < 1 || a.groups < 1 || a.groups > STEM_MAXGROUPS || a.row0[0] != 0 || a.row0[a.groups] != 92) return hipErrorInvalidValue;
    for (int g = 0; g < a.groups; g++) {
        const int hgt = (int)a.row0[g + 1] - (int)a.row0[g];
        if (hgt < 1 || hgt > STEM_MAXH) return hipErrorInvalidValue;
    }
    if (!a.w || !a.bias || !a.out || (a.from_frame ? (!a.fp || !a.tabs || !a.dyn.frame) : !a.batch)) return hipErrorInvalidValue;
    const dim3 grid(a.S * a.groups * 4), block(STEM_THREADS);
    const bool prof = a.prof != nullptr;
#define STEM_GO(BF, FR, PR) hipLaunchKernelGGL((stem_kernel<BF, FR, PR>), grid, block, stem_lds<BF>(), st, a)
    if (a.bf16) {
        if (a.from_frame) { if (prof) STEM_GO(true, true, true); else STEM_GO(true, true, false); }
        else { if (prof) STEM_GO(true, false, true); else STEM_GO(true, false, false); }
    } else {
        if (a.from_frame) { if (prof) STEM_GO(false, true, true); else STEM_GO(false, true, false); }
        else { if (prof) STEM_GO(false, false, true); else STEM_GO(false, false, false); }
    }
#undef STEM_GO
    return hipGetLastError();
}

}  // namespace vnect
