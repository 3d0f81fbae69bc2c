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
//   4. writes the pooled tile with 16-byte write-through stores -- or (PAIR, round 3) does not write it at all: the two 1x1 layers that
//      read pool1, res2a_branch2a (64 -> 64, ReLU) and res2a_branch1 (64 -> 256) (vnect_model.py:32-35; one launch of their own so far),
//      run here as a second GEMM on the pooled tile in LDS -- [92 or 115 pixels] x [64] x [64 x 320], K in the stand-alone launch's
//      order, so bit-identical to it -- and their outputs are what the launch stores.  One launch and pool1's round trip fewer.
#include "kernels.h"
#include "pyramid.h"

#include <cstdlib>
#include <type_traits>

namespace vnect {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int STEM_CW = 2 * STEM_TW + 1;             // conv columns per tile (47)
constexpr int STEM_LW = 48;                          // row stride of the tile's conv pixels in M (47 + one dummy): a lane's groups of four
                                                     // consecutive M rows then start at a column that is a multiple of 4 and never cross a row
constexpr int STEM_PH = 2 * (2 * STEM_MAXH + 1) + 6; // patch rows at most: 27, + 1 for the bf16 form's zero-weight eighth filter row
constexpr int STEM_THREADS = 512;

// FRAME form: scratch behind the pooled tile -- the `/255 - 0.4` table, one descriptor per patch row / column, and the rectangle of the
// frame the tile's patch is made from, as raw bytes (rows of STEM_REG_PITCH bytes)
struct StemRow {   // patch row pr = canvas row iy0 + pr
    short kind;    // 0: outside the 368 x 368 canvas (the conv's zero padding); 1: black canvas outside the scaled image;
                   // 2: a row of the square itself (q0); 3: blend of square rows q0, q1 with weights b0, b1 (OpenCV fixed point)
    short q0, q1, b0, b1;
    short fill_[3];
};
struct StemCol {   // patch column pc = canvas column ix0 + pc; kind 3: taps q, q + 1 with weights a0, a1; 4: the single tap q (right border)
    short kind, q, a0, a1;
};
constexpr int STEM_SCR_LUT = 0, STEM_SCR_ROWS = 1024, STEM_SCR_COLS = STEM_SCR_ROWS + 16 * 32, STEM_SCR_MIS = STEM_SCR_COLS + 8 * 104,
              STEM_SCR_REG = STEM_SCR_MIS + 4 * STEM_REG_ROWS, STEM_SCR_BYTES = STEM_SCR_REG + STEM_REG_ROWS * STEM_REG_PITCH;
static_assert(STEM_SCR_REG % 16 == 0 && STEM_PH <= 32 && STEM_PW <= 104, "scratch layout");

template <bool BF, bool FRAME = false>
constexpr size_t stem_lds()
{
    // patch, pooled tile, (FRAME) scratch, and 512 bytes for the PAIR form's table of output-pixel offsets
    // (bf16: at least the eight waves' output tiles of the PAIR form's write-out, 8 x 32 rows x 336 B)
    const size_t base = (size_t)STEM_PH * STEM_PW * 4 * (BF ? 2 : 4) + (size_t)STEM_MAXH * STEM_TW * 64 * 4 + (FRAME ? STEM_SCR_BYTES : 0);
    return (BF && base < (size_t)8 * 32 * 336 ? (size_t)8 * 32 * 336 : base) + 512;
}

// FRAME: the patch is computed from the uint8 frame instead of being copied from the batch tensor (the host takes this form only for
// frames whose squarify step is a copy, and only if every tile's rectangle fits the scratch: plan::stem_frame_fits)
// scalar write-through stores (conv.hip's epilogue stores: the bytes leave the L2 while the kernel runs)
__device__ __forceinline__ void stem_put(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stem_put(__bf16* p, float v)
{
    const __bf16 b = (__bf16)v;  // round to nearest even
    __hip_atomic_store((unsigned short*)p, __builtin_bit_cast(unsigned short, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// end stamps of the profiling twin: every workgroup's end (the host takes the maximum), and workgroup 0's span in both clocks
__device__ __forceinline__ void stem_prof_end(const StemArgs& a)
{
    if (blockIdx.x == 0) {
        a.prof[25] = (unsigned long long)__builtin_amdgcn_s_memtime();
        a.prof[26] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
    }
    a.prof_end[blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
}

template <bool BF, bool FRAME, bool PROF, bool PAIR = false>
__global__ __launch_bounds__(STEM_THREADS, 2) void stem_kernel(const StemArgs a)
{
    typedef typename std::conditional<BF, __bf16, float>::type T;
    typedef T tx4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    T* patch = (T*)smem;                                                     // [PH][PW][4]
    float* pooled = smem + STEM_PH * STEM_PW * 4 * sizeof(T) / sizeof(float);  // [h][23][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (PROF && tid == 0 && blockIdx.x == 0) {
        a.prof[0] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
        a.prof[24] = (unsigned long long)__builtin_amdgcn_s_memtime();  // shader cycles (kernels.h: PROF_SLOTS)
    }

    // ---- tile geometry (uniform) ---------------------------------------------------------------------------------------
    const int tile = blockIdx.x, c = tile & 3, gi = tile >> 2;
    const int sI = gi / a.groups, g = gi - sI * a.groups;
    const int r0 = a.row0[g], r1 = a.row0[g + 1], h = r1 - r0;     // pooled rows [r0, r1) of image sI
    const int cy0 = 2 * r0, cx0 = 2 * STEM_TW * c;                  // first conv row / column of the tile
    const int nrow = 2 * h + 1;                                    // conv rows 2 r0 .. 2 r1
    const int npix = nrow * STEM_LW, nblk = (npix + 31) >> 5;  // M = conv pixels in rows of 48 (column 47 is a dummy)
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

    // PAIR: element offset (into the 64-channel output tensor) of the pixel behind each of the tile's up to 128 GEMM rows, one entry
    // per thread now instead of 16 divisions per lane later.  Rows past the tile point at the tensors' slack pixels (64 behind the
    // last real one, rt_plan.cpp: never read), so that the stores need no per-row branch.
    unsigned* const pixtab = (unsigned*)((char*)smem + stem_lds<BF, FRAME>() - 512);
    if constexpr (PAIR) {
        if (tid < 128) {
            const int py = tid / STEM_TW, px = tid - py * STEM_TW;
            const int pixel = tid < h * STEM_TW ? ((sI * 92 + r0 + py) * 92 + STEM_TW * c + px) : a.S * 92 * 92 + (tid & 31);
            pixtab[tid] = (unsigned)pixel * 64u;
        }
    }

    // ---- 1. the input patch --------------------------------------------------------------------------------------------
    for (int i = tid; i < h * STEM_TW * 16; i += STEM_THREADS) ((f32x4*)pooled)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.dbg & 4) {
        for (int i = tid; i < prow * STEM_PW; i += STEM_THREADS) ((tx4*)patch)[i] = tx4{(T)0.f, (T)0.f, (T)0.f, (T)0.f};
    } else if constexpr (FRAME) {
        // gen_input_batch for this tile (estimator.py:70-81; the arithmetic of pyramid.h, which pyramid_kernel evaluates pixel by pixel
        // with ~16 dependent byte gathers each): here the frame bytes the whole patch needs are landed ONCE, as aligned dwords, and
        // every table entry the patch needs is in LDS before the first pixel is computed -- three short phases, no global load in the
        // per-pixel code.  Only for frames whose squarify step is a copy (frame long side == 368: square(y, x) = frame[y - offy][x -
        // offx] or black).
        char* scr = (char*)(pooled + STEM_MAXH * STEM_TW * 64);
        float* lut = (float*)(scr + STEM_SCR_LUT);
        StemRow* rows = (StemRow*)(scr + STEM_SCR_ROWS);
        StemCol* cols = (StemCol*)(scr + STEM_SCR_COLS);
        int* rowmis = (int*)(scr + STEM_SCR_MIS);
        unsigned char* reg = (unsigned char*)(scr + STEM_SCR_REG);
        const int s = sI + a.scale_base;
        const ResizeTab& t = a.tabs->t[s];
        const int scaled = a.tabs->scaled[s];
        const bool resize = scaled && !t.copy;       // (a scale whose size rounds to 368 is a plain copy of the square)
        const int off = scaled ? a.tabs->pad[s] : 0;  // origin and size of the scaled image on the canvas
        const int dh = scaled ? t.dh : BOX, dw = scaled ? t.dw : BOX;
        const int offy = a.fp->offy, offx = a.fp->offx, FH = a.fp->sq.dh, FW = a.fp->sq.dw;  // the frame inside the square
        // -- A + B in ONE memory round trip and one barrier (round 5; they used to be two of each: the rectangle was computed from the
        //    row / column descriptors in LDS, i.e. behind the barrier that published them):
        //    (A) tables -> LDS: the `/255 - 0.4` table, one descriptor per patch row / column;
        //    (B) the rectangle of the square the patch reads, [qy0, qy1] x [qx0, qx1] (from the first / last patch row and column that lie
        //    inside the scaled image), as frame bytes.  Its four corners are four UNIFORM table entries, fetched here directly (the same
        //    entries the descriptors hold: plan::stem_frame_fits does this arithmetic on the host).  LDS row r holds square row qy0 + r;
        //    pixel qx sits at byte 3 (qx - qx0) + rowmis[r], the offset chosen so that LDS dwords and global dwords are aligned to each
        //    other; bytes outside the frame stay 0.
        const int ylo = iy0 > off ? iy0 : off, yhi = (iy0 + prow - 1 < off + dh - 1 ? iy0 + prow - 1 : off + dh - 1);
        const int xlo = ix0 > off ? ix0 : off, xhi = (ix0 + STEM_PW - 1 < off + dw - 1 ? ix0 + STEM_PW - 1 : off + dw - 1);
        int qy0 = 0, qx0 = 0, RH = 0, RW = 0;
        if (ylo <= yhi && xlo <= xhi) {  // (uniform)
            qy0 = resize ? t.sy0[ylo - off] : ylo - off, qx0 = resize ? t.sx[xlo - off] : xlo - off;
            const int qy1 = resize ? t.sy1[yhi - off] : yhi - off;
            int qx1 = resize ? t.sx[xhi - off] + 1 : xhi - off;
            if (qx1 > BOX - 1) qx1 = BOX - 1;
            RH = qy1 - qy0 + 1, RW = qx1 - qx0 + 1;  // <= STEM_REG_ROWS, 3 RW + 8 <= STEM_REG_PITCH: plan::stem_frame_fits
        }
        qy0 = __builtin_amdgcn_readfirstlane(qy0), qx0 = __builtin_amdgcn_readfirstlane(qx0);
        RH = __builtin_amdgcn_readfirstlane(RH), RW = __builtin_amdgcn_readfirstlane(RW);
        const float lutv = tid < 256 ? a.tabs->lut[tid] : 0.f;
        StemRow rdesc = {0, 0, 0, 0, 0, {0, 0, 0}};
        if (tid < prow) {
            const int y = iy0 + tid, dy = y - off;
            if ((unsigned)y < (unsigned)BOX) {
                if ((unsigned)dy >= (unsigned)dh) rdesc.kind = 1;
                else if (!resize) rdesc.kind = 2, rdesc.q0 = rdesc.q1 = (short)dy;
                else rdesc.kind = 3, rdesc.q0 = t.sy0[dy], rdesc.q1 = t.sy1[dy], rdesc.b0 = t.b0[dy], rdesc.b1 = t.b1[dy];
            }
        }
        StemCol cdesc = {0, 0, 0, 0};
        if (tid >= 128 && tid < 128 + STEM_PW) {
            const int x = ix0 + tid - 128, dx = x - off;
            if ((unsigned)x < (unsigned)BOX) {
                if ((unsigned)dx >= (unsigned)dw) cdesc.kind = 1;
                else if (!resize) cdesc.kind = 2, cdesc.q = (short)dx;
                else cdesc.kind = dx < t.xmax ? 3 : 4, cdesc.q = t.sx[dx], cdesc.a0 = t.a0[dx], cdesc.a1 = t.a1[dx];
            }
        }
        {
            // The rectangle's dwords, row-major with RPW = the dwords a row really needs (3 RW + 6 bytes and up to 3 of alignment), dealt
            // over the 512 threads: slot k of this thread is dword i = tid + 512 k.  Only the first `trips` slots exist (uniform), 4-5 for
            // a tile of the full-size image, ~12 at scale 0.6 -- the worst case the scratch admits (64 rows x 160 dwords) is 20, and every
            // slot costs ~30 instructions of address arithmetic whether or not it holds a byte (they all ran until round 5).
            constexpr int PER = STEM_REG_ROWS * (STEM_REG_PITCH / 4) / STEM_THREADS;  // 20 slots at most
            const int RPW = __builtin_amdgcn_readfirstlane((3 * RW + 6 + 3 + 3) >> 2);  // <= STEM_REG_PITCH / 4
            const int total = RH * RPW, trips = __builtin_amdgcn_readfirstlane((total + STEM_THREADS - 1) / STEM_THREADS);
            // i / RPW in float: i + 0.5 is at least 0.5 / 160 away from a multiple of RPW, the reciprocal's and the product's rounding
            // errors are five orders below that (i < 10 240)
            const float rpw_inv = __builtin_amdgcn_rcpf((float)RPW);
            const int fx0 = qx0 - offx;                             // frame column of square column qx0 (may be negative)
            const int fxL = fx0 > 0 ? fx0 : 0, fxR = (fx0 + RW - 1 < FW - 1 ? fx0 + RW - 1 : FW - 1);
            const int lead = fxL - fx0, cnt = fxR - fxL + 1;        // clipped-away pixels on the left; pixels inside the frame
            const unsigned long long fbase = (unsigned long long)a.dyn.frame;
            unsigned v[PER], msk[PER];
#pragma unroll
            for (int k = 0; k < PER; k++) {
                if (k < trips) {  // (uniform)
                    const int i = tid + k * STEM_THREADS, r = (int)(((float)i + 0.5f) * rpw_inv), j = i - r * RPW;
                    const int fy = qy0 + r - offy;
                    const bool rv = i < total && (unsigned)fy < (unsigned)FH && cnt > 0;
                    const unsigned long long G = fbase + (unsigned long long)(rv ? fy : 0) * (unsigned long long)a.dyn.row_stride + 3ull * fxL;
                    const int m = (int)((G - 3ull * lead) & 3ull);
                    const int lo = 3 * lead + m, hi = lo + 3 * cnt;       // the row's valid bytes in LDS
                    int first = lo - 4 * j, last = hi - 4 * j;
                    first = first < 0 ? 0 : first, last = last > 4 ? 4 : last;
                    const bool live = rv && last > first;
                    msk[k] = live ? (0xFFFFFFFFu >> (8 * (4 - last))) & (0xFFFFFFFFu << (8 * first)) : 0u;
                    const unsigned long long addr = live ? G - (unsigned long long)lo + 4ull * j : fbase;  // aligned either way
                    v[k] = *(__attribute__((address_space(1))) const unsigned*)addr;  // global_load_dword (a generic pointer would be a FLAT load)
                    if (j == 0 && i < total) rowmis[r] = m;
                }
            }
            if (tid < 256) lut[tid] = lutv;
            if (tid < prow) rows[tid] = rdesc;
            if (tid >= 128 && tid < 128 + STEM_PW) cols[tid - 128] = cdesc;
#pragma unroll
            for (int k = 0; k < PER; k++) {
                if (k < trips) {
                    const int i = tid + k * STEM_THREADS, r = (int)(((float)i + 0.5f) * rpw_inv), j = i - r * RPW;
                    if (i < total) ((unsigned*)reg)[r * (STEM_REG_PITCH / 4) + j] = v[k] & msk[k];
                }
            }
        }
        __syncthreads();
        // -- C: the patch, pixel by pixel, from LDS only
        for (int i = tid; i < prow * STEM_PW; i += STEM_THREADS) {
            const int pr = i / STEM_PW, pc = i - pr * STEM_PW;
            const StemRow r = rows[pr];
            const StemCol q = cols[pc];
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (r.kind && q.kind) {
                int val[3] = {0, 0, 0};
                if (r.kind >= 2 && q.kind >= 2) {
                    const unsigned char* R0 = reg + (r.q0 - qy0) * STEM_REG_PITCH + rowmis[r.q0 - qy0] + 3 * (q.q - qx0);
                    if (r.kind == 2) {
                        val[0] = R0[0], val[1] = R0[1], val[2] = R0[2];
                    } else {
                        const unsigned char* R1 = reg + (r.q1 - qy0) * STEM_REG_PITCH + rowmis[r.q1 - qy0] + 3 * (q.q - qx0);
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            int h0, h1;
                            if (q.kind == 3) h0 = R0[ch] * q.a0 + R0[3 + ch] * q.a1, h1 = R1[ch] * q.a0 + R1[3 + ch] * q.a1;
                            else h0 = R0[ch] * 2048, h1 = R1[ch] * 2048;
                            val[ch] = (((r.b0 * (h0 >> 4)) >> 16) + ((r.b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                        }
                    }
                }
                o = f32x4{lut[val[0]], lut[val[1]], lut[val[2]], 0.f};
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
    for (int b = mg; b < ((a.dbg & 1) ? 0 : nblk); b += 4) {
        // A fragments: this lane's row is conv pixel p = 32 b + (lane & 31) of the tile (rows past the tile read pixel 0: finite values,
        // results unused)
        int p = 32 * b + col;
        if (p >= npix) p = 0;
        const int cy = p / STEM_LW;
        int cx = p - cy * STEM_LW;
        if (cx >= STEM_CW) cx = 0;  // the dummy column reads pixel 0 of its row
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
        // C/D map: column = lane & 31 (channel n), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): this lane holds four groups of four
        // consecutive M rows, group g at M row 32 b + 4 hh + 8 g = conv pixels (y, x0 .. x0 + 3) with x0 a multiple of 4.  Of the
        // pooling windows (3 wide, stride 2) that contain them, column window x0 / 2 is complete in the group (x0, x0 + 1, x0 + 2),
        // window x0 / 2 + 1 gets (x0 + 2, x0 + 3) and window x0 / 2 - 1 gets x0; row-wise conv row y lies in pooled row y >> 1 and, if
        // even and >= 2, in the one above: at most 6 ds_max_f32 per four values.  Pixels outside the tile / the 184 x 184 conv output
        // contribute 0, the pooled tile's initial value (ReLU outputs are >= 0).
        int pp = 32 * b + 4 * hh;
        int y = pp / STEM_LW, x0 = pp - y * STEM_LW;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v[e] = __builtin_fmaxf(acc[4 * g4 + e] + bias, 0.f);
                if constexpr (BF) v[e] = (float)(__bf16)v[e];  // what the stand-alone conv1 stores (round to nearest even), then pools
                if (x0 + e >= STEM_CW || cx0 + x0 + e >= 2 * 92) v[e] = 0.f;  // the dummy column; conv column 184 (the pool's padding)
            }
            if (y < nrow && cy0 + y < 2 * 92 && !(a.dbg & 2)) {
                const float m012 = __builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), v[2]), m23 = __builtin_fmaxf(v[2], v[3]);
                const int pa = y >> 1, pb = x0 >> 1;
                const bool up = !(y & 1) && y >= 2;  // also the last row of the pooled row above
                float* cell = pooled + (pa * STEM_TW + pb) * 64 + n;
                if (pa < h) {
                    if (pb < STEM_TW) __hip_atomic_fetch_max(cell, m012, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (pb + 1 < STEM_TW) __hip_atomic_fetch_max(cell + 64, m23, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (pb >= 1) __hip_atomic_fetch_max(cell - 64, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (up) {
                    float* above = cell - STEM_TW * 64;
                    if (pb < STEM_TW) __hip_atomic_fetch_max(above, m012, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (pb + 1 < STEM_TW) __hip_atomic_fetch_max(above + 64, m23, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (pb >= 1) __hip_atomic_fetch_max(above - 64, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            x0 += 8;  // the lane's next group: 8 M rows on
            if (x0 >= STEM_LW) x0 -= STEM_LW, y++;
        }
    }
    __syncthreads();

    if constexpr (PAIR) {
        // ---- 4'. the pair GEMM: rows = the tile's pooled pixels (row-major, 32 per block: 3 or 4 blocks), columns = 64 + 256 channels
        // (10 blocks), K = 64.  Wave (rb = wave & 3, half = wave >> 2) takes row block rb and column blocks 5 half .. 5 half + 4: its A
        // fragments (read once; the tile's rows are 256 B apart, so these reads conflict -- 8 of them per wave) serve all five.
        typedef __attribute__((address_space(1))) const f32x4 cgf4;
        constexpr int NQ = BF ? 4 : 8;
        const int npx = h * STEM_TW, rb = wave & 3, cb0 = 5 * (wave >> 2);
        const bool active = rb * 32 < npx && !(a.dbg & 16);  // (uniform; dbg 16: no pair GEMM at all)
        int ia = rb * 32 + col;  // (a wave without a row block reads pixel 0 like the rows past the tile: unused)
        if (ia >= npx) ia = 0;  // rows past the tile read pixel 0: finite values, results unused
        f32x4 Af[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if constexpr (BF) {  // the pooled values are bf16 numbers held as fp32 (rounded before the max): the conversion is exact
                const f32x4 lo = *(const f32x4*)(pooled + ia * 64 + 16 * q + 8 * hh), hi = *(const f32x4*)(pooled + ia * 64 + 16 * q + 8 * hh + 4);
                const bf16x8 v = {(__bf16)lo[0], (__bf16)lo[1], (__bf16)lo[2], (__bf16)lo[3], (__bf16)hi[0], (__bf16)hi[1], (__bf16)hi[2], (__bf16)hi[3]};
                Af[q] = __builtin_bit_cast(f32x4, v);
            } else
                Af[q] = *(const f32x4*)(pooled + ia * 64 + 8 * q + 4 * hh);
        }
        // output pixel offsets of the lane's 16 C/D rows (row = (r & 3) + 8 (r >> 2) + 4 hh): four 16-byte reads of the table
        unsigned offa[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 t4 = *(const u32x4*)(pixtab + rb * 32 + 4 * hh + 8 * g4);
#pragma unroll
            for (int e = 0; e < 4; e++) offa[4 * g4 + e] = t4[e];
        }
        // bf16: the output leaves through LDS tiles laid over the patch / pooled / scratch areas: every wave must have read its A fragments
        // and table entries (above) before any wave writes there
        if constexpr (BF) __syncthreads();
        if (active) {
            cgf4* bp = (cgf4*)a.pair_w + lane;
            // bf16: this wave's output tile in LDS, over the patch / pooled / scratch areas -- every wave has its A fragments in registers,
            // but waves still READING theirs must be past that point before anyone writes here: the barrier below
            constexpr int WT_LD = 168;  // row stride in elements: 336 B, 16-byte aligned, rows 4 banks apart
            __bf16* const wtile = (__bf16*)smem + wave * (32 * WT_LD);
            f32x4 Bc[NQ], Bn[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) Bc[q] = bp[(cb0 * NQ + q) * 64], Bn[q] = Bc[q];
            // A LOOP, not five unrolled copies: this code runs once per wave on a cold instruction cache (~35 cycles per instruction
            // fetched, conv.hip), and the first version -- 1 500 straight-line instructions with a branch around every store -- made the
            // launch 23 us longer than the stand-alone pair launch it replaced.
#pragma unroll 1
            for (int b = 0; b < 5; b++) {
                const int cb = cb0 + b, n2 = cb * 32 + col;
                if (b + 1 < 5) {
#pragma unroll
                    for (int q = 0; q < NQ; q++) Bn[q] = bp[((cb + 1) * NQ + q) * 64];
                }
                const float bias2 = a.pair_bias[n2];
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    if constexpr (BF) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Af[q]), __builtin_bit_cast(bf16x8, Bc[q]), acc, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Af[q][e], Bc[q][e], acc, 0, 0, 0);
                    }
                }
                // columns 0..63 = res2a_branch2a (ReLU), 64..319 = res2a_branch1 (none): uniform per block
                const bool first = cb < 2;
                T* ob = first ? (T*)a.pair_out_a + n2 : (T*)a.pair_out_b + (n2 - 64);
                if constexpr (BF) {
                    // bf16: the block goes to this wave's LDS tile [32 rows][160 columns + pad] and leaves in whole rows below (round 5: as
                    // 16 two-byte stores per lane and block -- 64-byte half lines -- the 16 MB of this GEMM's output took 8.6 us of a
                    // 26-us launch, tools/stem_breakdown.sh; lane pairs exchanging values by DPP for 4-byte stores did not change that)
                    __bf16* const tl = wtile + (cb - cb0) * 32 + col;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const float v = acc[r] + bias2;
                        tl[((r & 3) + 8 * (r >> 2) + 4 * hh) * WT_LD] = (__bf16)(first ? __builtin_fmaxf(v, 0.f) : v);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const float v = acc[r] + bias2;
                        stem_put(ob + (first ? offa[r] : offa[r] * 4u), first ? __builtin_fmaxf(v, 0.f) : v);
                    }
                }
#pragma unroll
                for (int q = 0; q < NQ; q++) Bc[q] = Bn[q];
            }
            if constexpr (BF) {
                // the wave's 32 x 160 tile, 16 bytes per lane and store: row = tile row -> its pixel (pixtab; rows past the tile go to the
                // tensors' slack pixels), 20 units per row -- wave half 0: 8 units of res2a_branch2a's 64 channels, then channels 0..95 of
                // res2a_branch1; half 1: channels 96..255 -- whole 128-, 192- and 320-byte runs
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's own tile: no barrier)
                const int hf = wave >> 2;
#pragma unroll 2
                for (int k = 0; k < 10; k++) {
                    const int i = lane + 64 * k, row = i / 20, u = i - row * 20;
                    const f32x4 val = *(const f32x4*)(wtile + row * WT_LD + 8 * u);
                    const unsigned pe = pixtab[rb * 32 + row];  // element offset of the pixel in the 64-channel tensor
                    __bf16* dst = (hf == 0 && u < 8) ? (__bf16*)a.pair_out_a + pe + 8 * u
                                                     : (__bf16*)a.pair_out_b + pe * 4u + (hf == 0 ? 8 * (u - 8) : 96 + 8 * u);
                    store_wt((f32x4*)dst, val);
                }
            }
        }
        if (PROF && tid == 0 && blockIdx.x < PROF_WGS) stem_prof_end(a);
        return;
    }
    // ---- 4. the pooled tile: rows r0 .. r1 - 1, columns 23 c .. 23 c + 22 of image sI, 64 channels (16 units of 4) ---------------
    for (int i = tid; i < h * STEM_TW * 16; i += STEM_THREADS) {
        const int cell = i >> 4, u = i & 15;
        const int pr = cell / STEM_TW, pc = cell - pr * STEM_TW;
        const f32x4 v = ((const f32x4*)pooled)[i];
        T* dst = (T*)a.out + (((long long)sI * 92 + r0 + pr) * 92 + STEM_TW * c + pc) * 64 + u * 4;
        store_wt((tx4*)dst, __builtin_convertvector(v, tx4));
    }
    if (PROF && tid == 0 && blockIdx.x < PROF_WGS) stem_prof_end(a);
}

hipError_t stem_setup()
{
    hipError_t e;
#define STEM_ATTR(BF, FR, PR)                                                                                                        \
    if ((e = hipFuncSetAttribute((const void*)stem_kernel<BF, FR, PR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stem_lds<BF, FR>())) != \
        hipSuccess)                                                                                                                  \
        return e;                                                                                                                    \
    if ((e = hipFuncSetAttribute((const void*)stem_kernel<BF, FR, PR, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stem_lds<BF, FR>())) != \
        hipSuccess)                                                                                                                  \
        return e;
    STEM_ATTR(false, false, false) STEM_ATTR(false, false, true) STEM_ATTR(false, true, false) STEM_ATTR(false, true, true)
    STEM_ATTR(true, false, false) STEM_ATTR(true, false, true) STEM_ATTR(true, true, false) STEM_ATTR(true, true, true)
#undef STEM_ATTR
    return hipSuccess;
}

hipError_t launch_stem(const StemArgs& a_in, hipStream_t st)
{
    StemArgs a = a_in;
    static const int dbg = getenv("VNECT_STEM_DBG") ? atoi(getenv("VNECT_STEM_DBG")) : 0;
    a.dbg = dbg;
    // host-side shape checks: the kernel's indexing assumes exactly these
    if (a.S < 1 || a.groups < 1 || a.groups > STEM_MAXGROUPS || a.row0[0] != 0 || a.row0[a.groups] != 92) return hipErrorInvalidValue;
    for (int g = 0; g < a.groups; g++) {
        const int hgt = (int)a.row0[g + 1] - (int)a.row0[g];
        if (hgt < 1 || hgt > STEM_MAXH) return hipErrorInvalidValue;
    }
    if (!a.w || !a.bias || !a.out || (a.from_frame ? (!a.fp || !a.tabs || !a.dyn.frame) : !a.batch)) return hipErrorInvalidValue;
    if (a.pair_w && (!a.pair_bias || !a.pair_out_a || !a.pair_out_b)) return hipErrorInvalidValue;
    const dim3 grid(a.S * a.groups * 4), block(STEM_THREADS);
    const bool prof = a.prof != nullptr;
#define STEM_GO(BF, FR, PR)                                                                                              \
    do {                                                                                                                 \
        if (a.pair_w) hipLaunchKernelGGL((stem_kernel<BF, FR, PR, true>), grid, block, (stem_lds<BF, FR>()), st, a);       \
        else hipLaunchKernelGGL((stem_kernel<BF, FR, PR>), grid, block, (stem_lds<BF, FR>()), st, a);                    \
    } while (0)
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
