// conv.hip -- fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32),
// plus the small layout / pooling / bone-length kernels of the VNect graph.
//
// Replaces the TF1 ops executed by sess.run at /root/reference/src/estimator.py:100-104 for the graph
// of src/vnect_model.py:25-217: Conv2D(+BiasAdd+Add+Relu), Conv2DBackpropInput (as 4 sub-pixel phases),
// FusedBatchNorm (folded into the epilogue), MaxPool, and the bone-length Mul/Add/Sqrt/Concat.
//
// Tiling: a 512-thread workgroup = 4 consumer waves (one per SIMD, one 32x32 MFMA accumulator each) + 4 producer
// waves.  K runs in 32-float chunks: a chunk is one filter tap and 32 consecutive input channels, i.e. one 128-byte
// run per NHWC input pixel, so global reads are whole cache lines.  A (gathered activation rows) and B (pre-packed
// weights, [N][K]) chunks go global -> LDS by buffer-addressed LDS-DMA into a ring of stages (details at the kernel).
#include "kernels.h"

#ifndef X3_DBG
#define X3_DBG 0  // tuning builds only (make VARIANT=_b EXTRA=-DX3_DBG=n): timing probes of the split-product loop, wrong results
#endif
#ifndef VNECT_AB
#define VNECT_AB 0  // tuning builds only: store forms of the epilogue (1 plain, 2 streaming, 3 write-through + streaming)
#endif
#ifndef BF16_NOSTORE
#define BF16_NOSTORE 0  // tuning builds only (wrong results): no bf16 epilogue stores
#endif

#include <cstdlib>
#include <type_traits>
#include <vector>

namespace vnect {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------
// The LDS image of a chunk is lane-linear per LDS-DMA instruction (8 rows x 128 B).  Bank conflicts are removed by
// fetching, for LDS slot (row r, 16-B unit u'), the SOURCE unit u = u' ^ ((r >> 1) & 7) and applying the same XOR on
// the fragment reads: a ds_read_b128 lane group (16 distinct rows, one unit) then covers all 16 slots of the 256-B
// bank row.  Chunks are issued NS-1 ahead with counted vmcnt waits; one raw s_barrier per chunk.
template <int N>
__device__ __forceinline__ void wait_vm()
{
    // (the counter has six bits; a larger count can only be asked for by a branch no ring of that shape ever takes, and the stricter wait is safe)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory");
}

// x / d for 0 <= x, x * d < 2^32, with mg = ceil(2^32 / d) (d == 1: mg wraps to 0, handled); one v_mul_hi instead of
// the ~35-instruction integer division sequence, which sat on every workgroup's critical path before its first load
// (d == 1: mg is 2^32 wrapped to 0, the product's high half is 0 and the mask adds x back -- no branch: the `d == 1 ? x : ...` form
// compiled to a compare-and-branch around every division of the cold start)
__device__ __forceinline__ int fdiv(int x, unsigned mg, int d) { return (int)(__umulhi((unsigned)x, mg) + ((unsigned)x & (0u - (unsigned)(d == 1)))); }

// Pointers that went through an SGPR pin (inline asm) lose their address space; accesses through them would be FLAT
// (counted on vmcnt AND lgkmcnt, so the compiler waits for each store before the next: measured 2.5-5 us per epilogue).
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) __bf16 gbf16;
typedef __attribute__((address_space(1))) const float cgfloat;
typedef __attribute__((address_space(1))) const __bf16 cgbf16;

// Epilogue stores are write-through (`sc1`): the bytes leave the XCD's L2 while the kernel still runs instead of in the
// write-back sweep at the kernel boundary, which every dependent launch behind this one waits for (A/B in one gpurun call:
// 1010 -> 1023 frames/s fp32).  Same bytes, same order of arithmetic: results are unchanged.
// (-DVNECT_AB=1, tools/ab.sh: plain stores again, to repeat the comparison.)
#ifndef F32_NOSTORE
#define F32_NOSTORE 0
#endif
__device__ __forceinline__ void put_f32(gfloat* p, float v)
{
#if F32_NOSTORE  // timing probe (make VARIANT=_nst EXTRA=-DF32_NOSTORE=1; wrong results): what the fp32 epilogue stores cost.  The value
    if (__builtin_bit_cast(unsigned, v) != 0x7fc12345u) return;  // stays live (a bare return lets the compiler drop the K loop with it)
#endif
#if VNECT_AB == 2  // probe: streaming (`nt`) stores
    __builtin_nontemporal_store(v, (float*)p);
#elif VNECT_AB == 3  // probe: write-through AND streaming
    asm volatile("global_store_dword %0, %1, off sc1 nt" ::"v"((float*)p), "v"(v) : "memory");
#elif VNECT_AB
    *p = v;
#else
    __hip_atomic_store((float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dword ... sc1
#endif
}
__device__ __forceinline__ void put_bf16(gbf16* p, float v)
{
#if BF16_NOSTORE  // timing probe (make VARIANT=_ns EXTRA=-DBF16_NOSTORE=1; wrong results): what the 2-byte stores of the bf16 epilogues cost
    return;           // -- up to 31 of 260 us of kernel time per frame (profiles/r05_bf16_store_cost.txt)
#endif
    const __bf16 b = (__bf16)v;  // round to nearest even
#if VNECT_AB
    *p = b;
#else
    __hip_atomic_store((unsigned short*)p, __builtin_bit_cast(unsigned short, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// Buffer-addressed LDS-DMA (`buffer_load_dwordx4 v_off, s[srd], s_off offen lds`): address = base + v_off + s_off, 16 bytes per
// lane to M0-base + lane * 16; a lane whose v_off is >= num_records fetches nothing and lands zeros.  The descriptor
// type only exists in the device pass.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t srd_t;
__device__ __forceinline__ srd_t make_srd(const void* base)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void bload_lds(srd_t r, float* lds, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, (int)soff, 0, 0);
}
#else
typedef int srd_t;
__device__ __forceinline__ srd_t make_srd(const void*) { return 0; }
__device__ __forceinline__ void bload_lds(srd_t, float*, unsigned, unsigned) {}
#endif

// ---------------------------------------------------------------------------------------------------------
// Bone-length features (vnect_model.py:198-209) inside the transposed conv's launch (FUSE = 2): the tile with output columns
// 128 .. 191 holds delta_x (128 + j), delta_y (149 + j), delta_z (170 + j), j < 21; its accumulators are in LDS ([64 rows][BONE_LS]),
// and the four producer waves (256 threads) write bone_j = sqrt((dx^2 + dy^2) + dz^2) to column 191 + j of every row's output
// pixel and zeros to the padding columns 212 .. ldc - 1, exactly what bone_kernel does as a launch of its own.
constexpr int BONE_N0 = 128, BONE_LS = 68;
__device__ __forceinline__ float bone_len(float x, float y, float z)
{
#pragma clang fp contract(off)
    return sqrtf((x * x + y * y) + z * z);  // one rounding per operation, in both the fused and the stand-alone form
}
template <bool BF>
__device__ __forceinline__ void bone_features(const ConvArgs& a, const float* t, int m0, int phase, int tid, int M, int Wo, int Ho,
                                              unsigned mg_wo, unsigned mg_ho)
{
    const int py = phase >> 1, px = phase & 1;
    const int npad = a.ldc - 212;  // padding columns behind the 212 features
    auto pixel = [&](int m) {
        const int q = fdiv(m, mg_wo, Wo), ox = m - q * Wo;
        const int sI = fdiv(q, mg_ho, Ho), oy = q - sI * Ho;
        return (sI * a.OH + oy * a.os + py) * a.OW + ox * a.os + px;
    };
    // thread -> (row = tid >> 2, quarter = tid & 3): a row's 21 + npad columns are walked by 4 threads; the row's output pixel
    // is computed once
    const int row = tid >> 2, m = m0 + row;
    if (m >= M) return;
    const unsigned base = (unsigned)(pixel(m) * a.ldc + 191);
    const float* tr = t + row * BONE_LS;
    for (int j = tid & 3; j < 21 + npad; j += 4) {
        float v = 0.f;
        if (j < 21) {
            float x = tr[j], y = tr[21 + j], z = tr[42 + j];
            if constexpr (BF) x = (float)(__bf16)x, y = (float)(__bf16)y, z = (float)(__bf16)z;  // what the stand-alone kernel reads back
            v = bone_len(x, y, z);
        }
        if constexpr (BF) put_bf16((gbf16*)a.out + base + j, v);
        else put_f32((gfloat*)a.out + base + j, v);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Second GEMM of a TAIL launch (see the kernel): out[64][tail_n] = mid[64][64] x tail_w^T + tail_bias + shortcut, ReLU, for the
// 64 rows m0 .. of this workgroup.  mid (the first layer's relu(acc + bias) tile) is in LDS at `smem`, row stride TAIL_MS.
// The output is 2 row halves x tail_n / 32 column blocks of 32 x 32 (K = 64 each, in the K order of the stand-alone 1x1
// layer).  All eight waves of the workgroup take part: a CONSUMER wave (wm, g) takes column block g, a PRODUCER wave (wm, g) the
// blocks 2 + g, 4 + g, ... -- three times as many, because a producer wave has registers to spare while the first GEMM runs: it
// requests the shortcut values of its first two blocks and the weight fragments of its first block BEFORE the K loop, so the
// shortcut read (half of this layer's bytes) overlaps the first GEMM's MFMA phase instead of following it.
template <bool BF>
constexpr int TAIL_MS = BF ? 72 : 68;  // mid row stride in elements: 144 B (bf16) / 272 B (fp32), conflict-free 16-byte reads
template <bool BF>
struct TailRegs {
    static constexpr int NQ = BF ? 4 : 8;  // MFMA groups over K = 64: 4 x (k = 16) bf16 / 8 x 4 x (k = 2) fp32
    unsigned rw[2][16];                    // shortcut values of two blocks, as loaded (fp32 bits or a zero-extended bf16)
    f32x4 Bf[NQ];                          // weight fragments of one block
};
// shortcut values of column block cb for the lane's 16 rows (mb = first row of the lane's C/D map)
template <bool BF>
__device__ __forceinline__ void tail_load_resid(const ConvArgs& a, int mb, int cb, int lane, unsigned (&rw)[16])
{
    const unsigned off = (unsigned)(mb * a.ldr + cb * 32 + (lane & 31));
    if constexpr (BF) {
        typedef __attribute__((address_space(1))) const unsigned short cgu16;
        cgu16* rp = (cgu16*)a.resid;
#pragma unroll
        for (int r = 0; r < 16; r++) rw[r] = rp[off], rp += ((r & 3) == 3 ? 5 : 1) * a.ldr;
    } else {
        typedef __attribute__((address_space(1))) const unsigned cgu32;
        cgu32* rp = (cgu32*)a.resid;
#pragma unroll
        for (int r = 0; r < 16; r++) rw[r] = rp[off], rp += ((r & 3) == 3 ? 5 : 1) * a.ldr;
    }
}
template <bool BF>
__device__ __forceinline__ void tail_load_b(const ConvArgs& a, int cb, int lane, f32x4 (&Bf)[TailRegs<BF>::NQ])
{
    // tail_w is packed in fragment order (hostplan.h: pack_tail): [block][q][lane] x 16 bytes -- one contiguous KiB per instruction
    typedef __attribute__((address_space(1))) const f32x4 cgf4;
    constexpr int NQ = TailRegs<BF>::NQ;
    cgf4* bp = (cgf4*)a.tail_w + cb * (NQ * 64) + lane;
#pragma unroll
    for (int q = 0; q < NQ; q++) Bf[q] = bp[q * 64];
}
// The 64-wide tail's chain (bf16 only, FUSE = 3 on 64x64 tiles): like the wide tail's, the block output tile (64 pixels x 256 channels)
// is kept in LDS behind the mid tile -- row stride 264 elements -- and the NEXT block's branch2a (1x1, 256 -> 64, ReLU:
// res2a -> res2b_branch2a, vnect_model.py:44-47) runs on it as a third GEMM (chain_narrow).
constexpr int NCHAIN_OS = 264;
constexpr int NCHAIN_OFF = 64 * TAIL_MS<true>;  // bf16 elements from the start of the LDS
// bf16 64-wide tail: does the launch take the staged form (its output through the LDS tile)?  Uniform, decided by the arguments alone, so
// every wave of the workgroup agrees and meets at the barrier in front of tail_write_out.
template <bool BF>
__device__ __forceinline__ bool tail_staged(const ConvArgs& a)
{
    return BF && a.resid != nullptr && a.relu_cols >= 256 && !a.out_f32 && a.Nvalid == 256 && a.tail_n == 256 && (a.ldc & 7) == 0;
}
// ... and the write-out: the 64 x 256 bf16 tile, 16 bytes per thread and store, all 512 threads (rows past M go to the tensor's slack)
__device__ __forceinline__ void tail_write_out(const ConvArgs& a, const float* smem, int m0)
{
    const __bf16* tile = (const __bf16*)smem + NCHAIN_OFF;
    __bf16* out = (__bf16*)a.out + (long long)m0 * a.ldc;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = (int)threadIdx.x + 512 * k, row = i >> 5, u = i & 31;
        store_wt((f32x4*)(out + row * a.ldc + 8 * u), *(const f32x4*)(tile + row * NCHAIN_OS + 8 * u));
    }
}
// NBLK blocks cb0, cb0 + cbs, ...; PRE: T.rw[0], T.rw[1] (blocks 0, 1) and T.Bf (block 0) were requested by the caller
template <bool BF, int NBLK, bool PRE, bool CH = false>
__device__ __forceinline__ void tail_gemm(const ConvArgs& a, const float* smem, int m0, int wm, int cb0, int cbs, int lane, TailRegs<BF>& T)
{
    static_assert(!CH || BF, "the 64-wide chain exists in bf16 only");
    constexpr int MS = TAIL_MS<BF>, NQ = TailRegs<BF>::NQ;
    constexpr int UQ = BF ? 16 : 8, UH = BF ? 8 : 4;
    const int col = lane & 31, hh = lane >> 5;
    const int mb = m0 + wm * 32 + 4 * hh;  // C/D map: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    f32x4 Af[NQ];
    f32x16 acc;
    {
        const int arow = wm * 32 + col;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if constexpr (BF) Af[q] = *(const f32x4*)((const __bf16*)smem + arow * MS + UQ * q + UH * hh);
            else Af[q] = *(const f32x4*)(smem + arow * MS + UQ * q + UH * hh);
        }
    }
    if constexpr (!PRE) {
        tail_load_b<BF>(a, cb0, lane, T.Bf);
        if (a.resid) tail_load_resid<BF>(a, mb, cb0, lane, T.rw[0]);
        if (a.resid && NBLK > 1) tail_load_resid<BF>(a, mb, cb0 + cbs, lane, T.rw[1]);
    }
    const bool t_of32 = !BF || a.out_f32;
#pragma unroll
    for (int b = 0; b < NBLK; b++) {
        const int cb = cb0 + b * cbs, n2 = cb * 32 + col;
        const float bias2 = ((cgfloat*)a.tail_bias)[n2];
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if constexpr (BF) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Af[q]), __builtin_bit_cast(bf16x8, T.Bf[q]), acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Af[q][e], T.Bf[q][e], acc, 0, 0, 0);
            }
        }
        if (b + 1 < NBLK) tail_load_b<BF>(a, cb + cbs, lane, T.Bf);  // in flight behind this block's epilogue
        const bool relu2 = cb * 32 < a.relu_cols;                      // uniform per block
        const unsigned off0 = (unsigned)(mb * a.ldc + n2);
        gfloat* op = (gfloat*)a.out;
        unsigned(&rw)[16] = T.rw[b & 1];
        if (BF && tail_staged<BF>(a)) {
            // bf16, the case the network has (a block output: shortcut + ReLU, stored as bf16, all 256 columns valid): the block goes to the
            // workgroup's OUTPUT TILE in LDS (64 rows x 256 channels behind the mid tile) and leaves from there in whole 512-byte rows
            // (tail_write_out, round 5) -- as 16 two-byte stores per lane and block, 64-byte half lines, the 13 MB of a 92x92 block output
            // cost ~3 us of a 10-us launch (profiles/r05_bf16_store_cost.txt).  The chain GEMM (CH) reads the same tile.
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float o = __builtin_fmaxf(acc[r] + bias2 + __builtin_bit_cast(float, rw[r] << 16), 0.f);
                ((__bf16*)smem)[NCHAIN_OFF + (wm * 32 + 4 * hh + (r & 3) + 8 * (r >> 2)) * NCHAIN_OS + n2] = (__bf16)o;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float o = acc[r] + bias2;
                if (a.resid) o = o + __builtin_bit_cast(float, BF ? rw[r] << 16 : rw[r]);
                if (relu2) o = __builtin_fmaxf(o, 0.f);
                const unsigned oo = off0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * a.ldc);
                if (n2 < a.Nvalid) {
                    if (t_of32) put_f32(op + oo, o);
                    else put_bf16((gbf16*)op + oo, o);
                }
            }
        }
        if (b + 2 < NBLK && a.resid) tail_load_resid<BF>(a, mb, cb + 2 * cbs, lane, T.rw[b & 1]);  // the slot just consumed
    }
}

// ---------------------------------------------------------------------------------------------------------
// The WIDE tail (round 3): the same fusion for a 3x3 layer with 128 output channels -- res3*_branch2b -> res3*_branch2c (+ shortcut,
// ReLU; vnect_model.py:62-103) and the head's res5c_branch2b -> res5c_branch2c (the final maps; vnect_model.py:211-217).  The 3x3 layer
// runs on 32 x 128 tiles (conv_stream_kernel<32, 128, 1, ...>: four consumer waves side by side over N, the same number of
// workgroups and the same MFMA work per wave as the 64 x 64 plan's 200 tiles), so a workgroup owns ALL 128 channels of its 32 pixels;
// relu(acc + bias) goes to LDS as a 32 x 128 tile and the 1x1 layer is a second GEMM with K = 128 over tail_n / 32 column blocks:
// tail wave tw (the four producer waves first -- they hold their first block's weight fragments and both blocks' shortcut values in
// registers since before the K loop -- then the four consumer waves) takes blocks tw, tw + 8.  Every output accumulates its 128
// products in the order the stand-alone layer does (chunk by chunk, unit by unit, the same MFMA instruction), from the values that
// layer would read back from HBM: bit-identical to the two launches.  These instantiations may use 256 VGPRs (their 100-KB ring admits
// one workgroup per CU anyway), so a wave keeps all of its A fragments (64 registers in fp32) across its blocks.
#ifndef WT_DBG
#define WT_DBG 0  // timing probes of the wide tail (wrong results): 1 = 1 / NQ of the MFMAs, 2 = no stores, 3 = one weight load per block
#endif
template <bool BF>
constexpr int WIDE_MS = BF ? 136 : 132;  // mid row stride in elements: 272 B (bf16) / 528 B (fp32), conflict-free 16-byte reads
// The chain GEMM's A operand: the tail's OUTPUT tile (32 pixels x 512 channels, what the block writes to HBM) kept in LDS behind the
// mid tile; row stride 516 / 520 elements (conflict-free 16-byte reads, like the mid tile's)
template <bool BF>
constexpr int CHAIN_OS = BF ? 520 : 516;
template <bool BF>
constexpr int CHAIN_OFF = 32 * WIDE_MS<BF>;  // elements from the start of the LDS
template <bool BF>
struct WideRegs {
    static constexpr int NQ = BF ? 8 : 16;  // MFMA groups over K = 128
    unsigned rw[2][16];                      // shortcut values of this wave's two blocks, as loaded
    f32x4 Bf[2][NQ];                         // weight fragments of both blocks
};
template <bool BF>
__device__ __forceinline__ void wide_load_b(const ConvArgs& a, int cb, int lane, f32x4 (&Bf)[WideRegs<BF>::NQ])
{
    // tail_w is packed in fragment order (hostplan.h: pack_tail): [block][q][lane] x 16 bytes -- one contiguous KiB per instruction
    typedef __attribute__((address_space(1))) const f32x4 cgf4;
    constexpr int NQ = WideRegs<BF>::NQ;
    cgf4* bp = (cgf4*)a.tail_w + cb * (NQ * 64) + lane;
#pragma unroll
    for (int q = 0; q < NQ; q++) Bf[q] = bp[(WT_DBG == 3 ? 0 : q) * 64];
}
// the shortcut values of a tail wave's blocks (any wave can request them before the K loop: 32 registers) ...
template <bool BF>
__device__ __forceinline__ void wide_load_resid(const ConvArgs& a, int m0, int tw, int lane, WideRegs<BF>& T)
{
    const int nb = (a.tail_n + 31) >> 5, mb = m0 + 4 * (lane >> 5);
    if (!a.resid) return;
    if (tw < nb) tail_load_resid<BF>(a, mb, tw, lane, T.rw[0]);
    if (tw + 8 < nb) tail_load_resid<BF>(a, mb, tw + 8, lane, T.rw[1]);
}
// ... and the weight fragments (a producer wave before the K loop -- it has the registers --, a consumer wave behind it)
template <bool BF>
__device__ __forceinline__ void wide_load_weights(const ConvArgs& a, int tw, int lane, WideRegs<BF>& T)
{
    const int nb = (a.tail_n + 31) >> 5;
    if (tw < nb) wide_load_b<BF>(a, tw, lane, T.Bf[0]);
    if (tw + 8 < nb) wide_load_b<BF>(a, tw + 8, lane, T.Bf[1]);
}
// Everything a wave's tail needs from memory has been requested by now (T); its two blocks' MFMA chains are independent and run
// interleaved, every A fragment (one 16-byte LDS read) feeding both.  The two tail waves of a SIMD -- one producer, one consumer
// wave -- share its matrix pipe: the producer's operands are in registers when the K loop ends, so its MFMAs cover the consumer's
// wait for its weight fragments.
template <bool BF>
constexpr int CHAIN_DEPTH = BF ? 16 : 8;  // weight groups of the chain GEMM in flight ahead of its MFMAs (registers the tail's own weights free)
// bf16 wide tail whose output is a block output (shortcut + ReLU, bf16, all 512 columns valid): staged like the 64-wide tail's (tail_staged /
// tail_write_out) -- the blocks go to the LDS output tile only (32 rows x 512 channels behind the mid tile, where the chain GEMM reads them
// anyway) and leave in whole 1-KiB rows, 16 bytes per thread and store, behind the barrier all eight tail waves meet at.
template <bool BF>
__device__ __forceinline__ bool wide_staged(const ConvArgs& a)
{
    return BF && a.resid != nullptr && a.relu_cols >= 512 && !a.out_f32 && a.Nvalid == 512 && a.tail_n == 512 && (a.ldc & 7) == 0;
}
__device__ __forceinline__ void wide_write_out(const ConvArgs& a, const float* smem, int m0)
{
    const __bf16* tile = (const __bf16*)smem + CHAIN_OFF<true>;
    __bf16* out = (__bf16*)a.out + (long long)m0 * a.ldc;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = (int)threadIdx.x + 512 * k, row = i >> 6, u = i & 63;
        store_wt((f32x4*)(out + row * a.ldc + 8 * u), *(const f32x4*)(tile + row * CHAIN_OS<true> + 8 * u));
    }
}
// CH: the launch has a chain GEMM behind this tail (chain_gemm below): every block's output also goes to the LDS output tile, and a
// wave that runs a chain block (cq != nullptr: tail waves 0..3) requests that block's first weight groups BEFORE its stores -- vmcnt
// counts loads and stores alike on this chip, so loads issued behind the 32 write-through stores could only be waited for together
// with them.
template <bool BF, bool CH>
__device__ __forceinline__ void tail_wide(const ConvArgs& a, const float* smem, int m0, int tw, int lane, WideRegs<BF>& T, f32x4* cq = nullptr)
{
    constexpr int MS = WIDE_MS<BF>, NQ = WideRegs<BF>::NQ;
    constexpr int UQ = BF ? 16 : 8, UH = BF ? 8 : 4;
    const int nb = (a.tail_n + 31) >> 5;
    if (tw >= nb) return;
    const bool two = tw + 8 < nb;  // (wave-uniform)
    const int col = lane & 31, hh = lane >> 5;
    const int mb = m0 + 4 * hh;  // C/D map: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const float bias0 = ((cgfloat*)a.tail_bias)[tw * 32 + col], bias1 = two ? ((cgfloat*)a.tail_bias)[(tw + 8) * 32 + col] : 0.f;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) acc0[r] = 0.f, acc1[r] = 0.f;
    auto afrag = [&](int q) __attribute__((always_inline)) {
        if constexpr (BF) return *(const f32x4*)((const __bf16*)smem + col * MS + UQ * q + UH * hh);
        else return *(const f32x4*)(smem + col * MS + UQ * q + UH * hh);
    };
    auto chain = [&](auto TWO) __attribute__((always_inline)) {
        f32x4 An = afrag(0);
#pragma unroll
        for (int q = 0; q < (WT_DBG == 1 ? 1 : NQ); q++) {
            const f32x4 Af = An;
            if (q + 1 < NQ) An = afrag(q + 1);
            if constexpr (BF) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Af), __builtin_bit_cast(bf16x8, T.Bf[0][q]), acc0, 0, 0, 0);
                if constexpr (decltype(TWO)::value)
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Af), __builtin_bit_cast(bf16x8, T.Bf[1][q]), acc1, 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Af[e], T.Bf[0][q][e], acc0, 0, 0, 0);
                    if constexpr (decltype(TWO)::value) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Af[e], T.Bf[1][q][e], acc1, 0, 0, 0);
                }
            }
        }
    };
    if (two) chain(std::true_type{});
    else chain(std::false_type{});
    if constexpr (CH) {
        if (cq) {
            typedef __attribute__((address_space(1))) const f32x4 cgf4;
            cgf4* bp = (cgf4*)a.chain_w + tw * ((512 / (BF ? 16 : 8)) * 64) + lane;
#pragma unroll
            for (int q = 0; q < CHAIN_DEPTH<BF>; q++) cq[q] = bp[q * 64];
        }
    }
    // Epilogue of a block, written for instruction count like the kernel's ordinary one (this code runs once per wave on a cold
    // instruction cache): shortcut / ReLU / output type are decided once per block (uniform branches around four-instruction rows),
    // the row base is a scalar pointer stepped by additions, the lane offset one 32-bit register.  Rows past M fall into the tensors'
    // 64-pixel slack (rt_plan.cpp); columns past Nvalid are masked once.
    const bool t_of32 = !BF || a.out_f32;
    auto finish = [&](int cb, const f32x16& acc, float bias2, unsigned(&rw)[16]) __attribute__((always_inline)) {
        const int n2 = cb * 32 + col;
        const bool relu2 = cb * 32 < a.relu_cols;  // uniform per block
        const unsigned off0 = (unsigned)(mb * a.ldc + n2);
        // fp32: one generic row sequence (shortcut / ReLU selected per element).  The specialised form below measured 0.3 % SLOWER
        // there (A/B in one call: 1 074 vs 1 078 frames/s) and 2.2 % faster in bf16 (2 453 vs 2 400), where the tail is all latency.
        if constexpr (!BF) {
            gfloat* op = (gfloat*)a.out;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float o = acc[r] + bias2;
                if (a.resid) o = o + __builtin_bit_cast(float, rw[r]);
                if (relu2) o = __builtin_fmaxf(o, 0.f);
                const unsigned oo = off0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * a.ldc);
                if (n2 < a.Nvalid && (WT_DBG != 2 || a.ldc < 0)) put_f32(op + oo, o);
                if constexpr (CH) const_cast<float*>(smem)[CHAIN_OFF<BF> + (4 * hh + (r & 3) + 8 * (r >> 2)) * CHAIN_OS<BF> + n2] = o;
            }
            return;
        }
        if constexpr (BF) {
            if (wide_staged<BF>(a)) {  // (uniform) the block output: to the LDS tile only, wide_write_out stores it
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float o = __builtin_fmaxf(acc[r] + bias2 + __builtin_bit_cast(float, rw[r] << 16), 0.f);
                    ((__bf16*)smem)[CHAIN_OFF<BF> + (4 * hh + (r & 3) + 8 * (r >> 2)) * CHAIN_OS<BF> + n2] = (__bf16)o;
                }
                return;
            }
        }
        auto rows = [&](auto RESID, auto RELU, auto OUTF32) __attribute__((always_inline)) {
            using OT = typename std::conditional<decltype(OUTF32)::value, gfloat, gbf16>::type;
            OT* op = (OT*)a.out;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float o = acc[r] + bias2;
                if constexpr (decltype(RESID)::value) o = o + __builtin_bit_cast(float, BF ? rw[r] << 16 : rw[r]);
                if constexpr (decltype(RELU)::value) o = __builtin_fmaxf(o, 0.f);
                if (WT_DBG != 2 || a.ldc < 0) {
                    if constexpr (decltype(OUTF32)::value) put_f32((gfloat*)op + off0, o);
                    else put_bf16((gbf16*)op + off0, o);
                }
                op += ((r & 3) == 3 ? 5 : 1) * a.ldc;
                if constexpr (CH) {  // the same value, in the precision it has in HBM, for the chain GEMM
                    const int row = 4 * hh + (r & 3) + 8 * (r >> 2);
                    if constexpr (BF) ((__bf16*)smem)[CHAIN_OFF<BF> + row * CHAIN_OS<BF> + n2] = (__bf16)o;
                    else const_cast<float*>(smem)[CHAIN_OFF<BF> + row * CHAIN_OS<BF> + n2] = o;
                }
            }
        };
        if (n2 < a.Nvalid) {
            if (a.resid) {  // (a block output: shortcut + ReLU, stored in the activations' type)
                if (relu2) {
                    if (t_of32) rows(std::true_type{}, std::true_type{}, std::true_type{});
                    else rows(std::true_type{}, std::true_type{}, std::false_type{});
                } else {
                    if (t_of32) rows(std::true_type{}, std::false_type{}, std::true_type{});
                    else rows(std::true_type{}, std::false_type{}, std::false_type{});
                }
            } else {
                if (relu2) {
                    if (t_of32) rows(std::false_type{}, std::true_type{}, std::true_type{});
                    else rows(std::false_type{}, std::true_type{}, std::false_type{});
                } else {
                    if (t_of32) rows(std::false_type{}, std::false_type{}, std::true_type{});
                    else rows(std::false_type{}, std::false_type{}, std::false_type{});
                }
            }
        }
    };
    finish(tw, acc0, bias0, T.rw[0]);
    if (two) finish(tw + 8, acc1, bias1, T.rw[1]);
}

// The chain GEMM: next block's branch2a (1x1, 512 -> 128, bias, ReLU) on the tile the wide tail has just left in LDS -- a third GEMM
// in the launch, [32 x 512] x [512 x 128]: four column blocks, one per SIMD, taken by the four PRODUCER waves (tail waves 0..3; each
// runs its block's whole K = 512 in one accumulator, chunk by chunk like the stand-alone layer: bit-identical).  The weights stream
// from global memory in fragment order (one KiB per instruction) through a ring of registers, CHAIN_DEPTH groups ahead of the MFMAs
// (the first of them requested inside tail_wide, in front of its stores).
template <bool BF>
__device__ __forceinline__ void chain_gemm(const ConvArgs& a, const float* smem, int m0, int cb, int lane, f32x4 (&Bq)[CHAIN_DEPTH<BF>])
{
    typedef __attribute__((address_space(1))) const f32x4 cgf4;
    constexpr int UQ = BF ? 16 : 8, UH = BF ? 8 : 4, NQ = 512 / UQ, OS = CHAIN_OS<BF>, DEPTH = CHAIN_DEPTH<BF>;
    const int col = lane & 31, hh = lane >> 5, mb = m0 + 4 * hh;
    cgf4* bp = (cgf4*)a.chain_w + cb * (NQ * 64) + lane;  // (groups 0 .. DEPTH - 1 were requested by tail_wide)
    const float bias3 = ((cgfloat*)a.chain_bias)[cb * 32 + col];
    auto afrag = [&](int q) __attribute__((always_inline)) {
        if constexpr (BF) return *(const f32x4*)((const __bf16*)smem + CHAIN_OFF<BF> + col * OS + UQ * q + UH * hh);
        else return *(const f32x4*)(smem + CHAIN_OFF<BF> + col * OS + UQ * q + UH * hh);
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    f32x4 An = afrag(0);
#ifndef CH_DBG
#define CH_DBG 0  // timing probes of the chain GEMM (wrong results): 1 = the first group's MFMAs only, 2 = no weight loads behind the first DEPTH groups
#endif
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const f32x4 Af = An, Bf = Bq[q % DEPTH];
        if (q + 1 < NQ) An = afrag(q + 1);
        if (CH_DBG == 1 && q > 0) continue;
        if constexpr (BF) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Af), __builtin_bit_cast(bf16x8, Bf), acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Af[e], Bf[e], acc, 0, 0, 0);
        }
        if (CH_DBG != 2 && q + DEPTH < NQ) Bq[q % DEPTH] = bp[(q + DEPTH) * 64];
    }
    const int n3 = cb * 32 + col;
    const unsigned off0 = (unsigned)(mb * a.chain_ld + n3);
    typedef typename std::conditional<BF, gbf16, gfloat>::type OT;
    OT* op = (OT*)a.chain_out;  // scalar row base stepped by additions, one lane offset
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float o = __builtin_fmaxf(acc[r] + bias3, 0.f);
        if constexpr (BF) put_bf16(op + off0, o);
        else put_f32(op + off0, o);
        op += ((r & 3) == 3 ? 5 : 1) * a.chain_ld;
    }
}

// [64 x 256] x [256 x 64]: 2 row halves x 2 column blocks, one per producer wave (rh = wave & 1, cb = wave >> 1), K = 256 in one
// accumulator (16 MFMAs: the stand-alone layer's order), weights in fragment order from L2.
__device__ __forceinline__ void chain_narrow(const ConvArgs& a, const float* smem, int m0, int wave, int lane)
{
    typedef __attribute__((address_space(1))) const f32x4 cgf4;
    const int rh = wave & 1, cb = wave >> 1, col = lane & 31, hh = lane >> 5;
    cgf4* bp = (cgf4*)a.chain_w + cb * (16 * 64) + lane;
    f32x4 Bq[16];
#pragma unroll
    for (int q = 0; q < 16; q++) Bq[q] = bp[q * 64];
    const float bias3 = ((cgfloat*)a.chain_bias)[cb * 32 + col];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // all eight tail waves have put their blocks of the output tile into LDS
    tail_write_out(a, smem, m0);   // this wave's share of the block output (the consumer waves do theirs behind the same barrier)
    const __bf16* at = (const __bf16*)smem + NCHAIN_OFF + (rh * 32 + col) * NCHAIN_OS + 8 * hh;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 16; q++)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *(const f32x4*)(at + 16 * q)), __builtin_bit_cast(bf16x8, Bq[q]), acc, 0, 0, 0);
    const unsigned off0 = (unsigned)((m0 + rh * 32 + 4 * hh) * a.chain_ld + cb * 32 + col);
    gbf16* op = (gbf16*)a.chain_out;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        put_bf16(op + off0, __builtin_fmaxf(acc[r] + bias3, 0.f));
        op += ((r & 3) == 3 ? 5 : 1) * a.chain_ld;
    }
}

// ---------------------------------------------------------------------------------------------------------
// The conv kernel.  Streaming: a workgroup walks SEVERAL work items (output tile x K slice) and treats their K
// chunks as one stream through the LDS ring.  The producers simply keep issuing -- the first chunks of the next
// tile land while the consumers are still in the previous tile's epilogue -- so the per-tile fixed cost (wave
// launch, argument fetch, first-load latency, store drain: ~5 us, more than the K loop of the 1x1 layers) is paid
// once per workgroup instead of once per tile.  The epilogue works from registers (no LDS scratch: the ring stays
// live): in the MFMA C layout a lane holds one output column, lanes 0-31 / 32-63 cover two rows of 32 consecutive
// columns, so every store / shortcut load instruction moves two full 128-byte lines.
// launch bounds (512, 4): at least 4 waves per SIMD = two workgroups per CU, i.e. a hard 128-VGPR budget
// PROF: 0 = the product kernel (no stamp code at all: a scalar branch between two MFMAs of the dependent chain costs
// ~35 cycles of matrix-pipe idle, tools/ring_rate.hip); 1 = start / end stamps (bench.py's profiling twin);
// 2 = per-phase stamps of workgroup 0 as well (tools/phase_table.py, VNECT_PROF_DETAIL=1).
//
// Tile shapes: the four consumer waves own one 32x32 accumulator each, arranged as (BM/32) x (BN/32) x KG.  KG > 1 is
// the IN-WORKGROUP K split for layers whose 64x64 tiles cannot fill the chip (M = 1587 at 23x23): K group kg takes the
// kg-th 128-byte run of every KG*128-byte step, and the groups' accumulators are summed through LDS in group order before
// the epilogue -- deterministic, no partial slabs in HBM, no second launch.  A step then moves KG*(BM+BN)*128 bytes:
// 24 KiB for 64x32x2, 32 KiB for 32x32x4, which the LDS-DMA sustains at one workgroup per CU (tools/ring_rate.hip).
// FUSE: what else the launch does behind a tile's K loop, in the LDS ring that a workgroup with ONE tile no longer needs by then
// (the host guarantees items <= grid): 1 = TAIL, a 1x1 conv on the tile as a second GEMM (tail_gemm); 2 = BONE, the bone-length
// features of vnect_model.py:198-209 from the transposed conv's delta columns (bone_features).
// SPAN (conv1, fp32): the A operand of a chunk is not gathered row by row.  The 64 output pixels of a tile sit in one output row (or
// in two, where the tile runs over a row's end), and their 8-pixel windows on input row 2 oy + ky - 2 overlap: together they are
// ONE contiguous run of 2 n + 6 NHWC4 pixels per output row -- 2.2 KB instead of the 8 KB of 64 gathered, 32-byte-aligned (two
// cache lines each) 128-byte windows, whose LDS-DMA issue paced the round-1 conv1 (0.9 us per chunk against 0.47 us of MFMA).
// The producers land the run(s) as they lie in memory (three DMA instructions per chunk instead of eight), and a consumer lane
// reads its fragment at pixel slot 2 row + 2 q + h of the run.  The K order is unchanged; the 16 MFMAs per chunk become 12: channel 3
// of the NHWC4 input is the zero padding channel (zero weights), so the e = 3 MFMA of every group adds exact zeros.
// X3 (fp32 activations, VNECT_FP32_SPLIT): the split-product form.  The fp32 matrix instruction runs at 1 / 16 of the bf16 one's rate on
// this chip (157 vs 2 500 TFLOP/s), so an fp32 product is cheaper as SIX bf16 products: every fp32 operand is the exact sum of three
// bf16 pieces (8 + 8 + 8 significand bits), x = xh + xm + xl, and x w = xh wh + xh wm + xm wh + xh wl + xl wh + xm wm + (three terms
// below 2^-23 |x w|, dropped), each bf16 x bf16 product exact in fp32 and all of them accumulated in fp32 by v_mfma_f32_32x32x16_bf16:
// 6 x 32 cycles per 16 K-elements against 8 x 64 for v_mfma_f32_32x32x2_f32.  The A operand stays what it is (fp32 tensors, the same
// LDS image); a consumer lane splits its 8 values per K step in registers (mask, subtract, mask, subtract: exact).  The weights are
// split offline (hostplan.h: pack_split3) and land as three 64-byte planes per row and chunk -- 20 KiB per stage instead of 16, so the
// ring has 4 stages in the same 80 KB.  Results differ from the fp32 instruction's in the last bits only (summation order, the dropped
// terms); the parity gates are the fp32 path's.
// ONE (round 5): the launch has exactly one work item per workgroup and no cross-workgroup K split (grid == items, ksplit == 1: 36 of the 39
// conv launches of a three-scale frame; the launcher checks) -- the streaming machinery (item count, stride, the next item's set-up inside
// the issue path, K slices) is compiled OUT of such a launch's kernel.  A launch's cold start is ~750 one-off instructions in front of the
// first DMA, issued at one per ~4 cycles by a single wave (profiles/r04_phase_producer_start.txt): what is not there is not issued.  Every
// fused form (FUSE != 0) is a one-item launch by construction.
template <int BM, int BN, int KG, int NS, bool BF, int PROF, int FUSE = 0, bool SPAN = false, bool X3 = false, bool ONE = false>
__global__ __launch_bounds__(512, (KG == 1 && BN <= 64) ? 4 : 2) void conv_stream_kernel(const ConvArgs a)
{
    constexpr bool TAIL = FUSE == 1 || FUSE == 3, BONE = FUSE == 2, CHAIN = FUSE == 3;  // 3: the wide tail with a chain GEMM behind it
    constexpr bool SINGLE = ONE || FUSE != 0;  // one item per workgroup, ksplit == 1
    static_assert(!SPAN || (BM == 64 && BN == 64 && KG == 1 && !BF && FUSE == 0), "span mode is conv1's fp32 form");
    // NACC (round 4): 32-column blocks per consumer wave.  64 x 96 x 2 -- two K groups x two row blocks, every wave THREE accumulators that share
    // its A fragments -- exists for the transposed conv in fp32: 300 tiles of 64 x 64 on 256 CUs are two rounds for 44 of them, 200 tiles
    // of 64 x 96 one round of 1.5 block-K-loops per SIMD (vnect_model.py:188-196).
    constexpr int NACC = BN == 96 ? 3 : 1;
    static_assert(NACC == 1 || (BM == 64 && KG == 2 && !BF && !X3 && !SPAN && (FUSE == 0 || FUSE == 2)), "the three-accumulator shape is fp32, plain or with the bone features");
    static_assert(!X3 || (BM == 64 && (BN * KG == 64) && !BF && !SPAN && PROF < 2), "split-product form: 64x64 and 64x32x2 tiles of fp32 layers");
    static_assert(FUSE == 0 || ((FUSE != 3 || (BF && !X3)) && BM == 64 && BN == 64 && KG == 1) || ((FUSE == 1 || FUSE == 3) && BM == 32 && BN == 128 && KG == 1) ||
                      (FUSE == 2 && BM == 64 && BN == 96 && KG == 2),
                  "the fused forms are built for one 64x64 tile per workgroup; the tail GEMM also for one 32x128 tile (tail_wide)");
    constexpr bool WIDE = BN == 128;
    constexpr int ESZ = BF ? 2 : 4;    // bytes per operand element
    constexpr int EPR = BF ? 64 : 32;  // K-elements per 128-B row (= per chunk)
    constexpr int EPU = BF ? 8 : 4;    // elements per 16-B unit
    constexpr int ARB = BM / 32, BRB = BN / 32, WNW = BRB / NACC, WMN = ARB * WNW;  // 32-row blocks of A and B; consumer waves over N; consumer waves per K group
    constexpr int BROWF = X3 ? 48 : 32;                          // floats per B row and chunk (X3: three 64-byte planes, plane-major in the stage)
    constexpr int ROWS = BM + BN, SUB = BM * 32 + BN * BROWF;    // one K group's image: BM A rows x 128 B, then the B rows
    constexpr int STAGE = SUB * KG, NLD = SPAN ? 1 + BRB : (X3 ? KG * ARB + 3 : KG * ROWS / 32);  // floats per ring stage; LDS-DMA instructions per producer wave per step
    constexpr int SCRATCH = NS * STAGE;                          // K-group partial sums: (KG-1) x WMN x NACC x 4 KiB, then WMN*(KG-1) flags
    constexpr int NPART = (KG - 1) * WMN * NACC;                  // 4-KiB partial accumulators behind the ring
    constexpr bool P1 = PROF >= 1, P2 = PROF >= 2;
    static_assert(WMN * KG == 4 && (BM == 32 || BM == 64) && (BN == 32 || BN == 64 || BN == 96 || BN == 128), "four consumer waves, NACC 32x32 accumulators each");
    static_assert(NS >= 3 && NS <= 9, "ring depth");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    {   // The 344-byte argument block spans six scalar-cache lines and the compiler loads fields where they are first
        // used, one scalar-cache round trip after another (cold: 1.7 us from wave start to the first LDS-DMA): touch every line now
        typedef __attribute__((address_space(4))) const int kint;
        kint* kp = (kint*)__builtin_amdgcn_kernarg_segment_ptr();
        const int k0 = kp[0], k1 = kp[16], k2 = kp[32], k3 = kp[48], k4 = kp[64], k5 = kp[80];
        asm volatile("" ::"s"(k0), "s"(k1), "s"(k2), "s"(k3), "s"(k4), "s"(k5));
        static_assert(sizeof(ConvArgs) <= 384 && sizeof(ConvArgs) > 320, "touch every 64-byte line of the argument block");
    }
    // Fields are pinned in SGPRs in three groups -- common, producer-only, consumer-only (the branch is wave-uniform, so
    // each path holds only its own) -- and each group is fetched as a few wide scalar loads with ONE wait.  Left to the
    // compiler, every field is loaded where it is first used: ~0.15 us per field even on scalar-cache hits, 2-4 us
    // in the epilogue alone (measured).
    struct Common {
        int M, Ho, Wo, ntaps, cpt, ksplit, tiles_m, tiles_n, items;
        unsigned mg_wo, mg_ho, mg_tn, mg_tm, mg_ks;
    } h = {a.M, a.Ho, a.Wo, a.ntaps, a.cpt, a.ksplit, a.tiles_m, a.tiles_n, a.items, a.mg_wo, a.mg_ho, a.mg_tn, a.mg_tm, a.mg_ks};
    asm volatile("" : "+s"(h.M), "+s"(h.Ho), "+s"(h.Wo), "+s"(h.ntaps), "+s"(h.cpt), "+s"(h.ksplit), "+s"(h.tiles_m),
                 "+s"(h.tiles_n), "+s"(h.items), "+s"(h.mg_wo), "+s"(h.mg_ho), "+s"(h.mg_tn), "+s"(h.mg_tm), "+s"(h.mg_ks));
    unsigned long long* const prof = a.prof;  // not pinned: stays a global-address-space pointer
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;
    const int kg = wave / WMN, wr = wave % WMN, wm = wr / WNW, wn = (wr % WNW) * NACC;  // consumers: K group, tile row block / first column block
    // start stamp: workgroup 0 (first dispatched; a grid starts first -> last within ~0.5 us).  Plain store: nothing in the
    // twin may queue behind an atomic.
    if (P1 && threadIdx.x == 0 && blockIdx.x == 0) {
        prof[0] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
        prof[24] = (unsigned long long)__builtin_amdgcn_s_memtime();  // shader cycles: with [25], [26] the clock held during this launch
    }
    const bool pstamp = P2 && threadIdx.x == 0 && blockIdx.x == 0;
    if (pstamp) prof[9] = __builtin_amdgcn_s_memrealtime();
    if (P2 && threadIdx.x == 0 && blockIdx.x + 8 >= gridDim.x) atomicMax(prof + 15, (unsigned long long)__builtin_amdgcn_s_memrealtime());

    // Work items of this workgroup.  XCD-aware order (speed only, never correctness): workgroup ids are dealt
    // round-robin over the 8 XCDs, so XCD x takes a contiguous eighth of the logical item sequence (N tile fastest,
    // then M tile, then phase / K slice) -- tiles that share activation rows or weights meet in one L2 -- and its
    // workgroups walk that range with stride = workgroups per XCD.  Every item is taken exactly once for any grid.
    int it_first, it_stride, my_n;
    {
        const int nwg = gridDim.x, id = blockIdx.x, xcd = id & 7, l = id >> 3;
        const int qd = h.items >> 3, rm = h.items & 7;
        const int lo = xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd;
        const int cnt = qd + (xcd < rm ? 1 : 0);
        const int nwx = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);
        it_first = lo + l, it_stride = nwx;
        // one item per workgroup (grid == items, most layers): skip the integer division (~35 cold instructions)
        if (SINGLE || h.items == nwg) my_n = 1;
        else my_n = __builtin_amdgcn_readfirstlane(l < cnt ? (cnt - l + nwx - 1) / nwx : 0);
    }
    if (!SINGLE && my_n == 0) return;  // whole workgroup: no barrier has been issued yet
    const int nch = h.ntaps * h.cpt;  // K steps per tile (the launcher passes cpt in steps of KG chunks)
    struct Item {
        int m0, n0, phase, ks, c0, cnt;
    };
    auto decode = [&](int j) __attribute__((always_inline)) {
        // everything about an item is wave-uniform: say so (the multiply-high runs on the vector ALU), so the item
        // bookkeeping below compiles to scalar code and scalar branches
        const int logical = __builtin_amdgcn_readfirstlane(it_first + j * it_stride);
        const int t2 = __builtin_amdgcn_readfirstlane(fdiv(logical, h.mg_tn, h.tiles_n));
        const int tile_n = logical - t2 * h.tiles_n;
        const int zz = __builtin_amdgcn_readfirstlane(fdiv(t2, h.mg_tm, h.tiles_m));
        const int tile_m = t2 - zz * h.tiles_m;
        Item it;
        if constexpr (SINGLE) it.phase = zz, it.ks = 0;
        else it.phase = __builtin_amdgcn_readfirstlane(fdiv(zz, h.mg_ks, h.ksplit)), it.ks = zz - it.phase * h.ksplit;
        it.m0 = tile_m * BM, it.n0 = tile_n * BN;
        if (SINGLE || h.ksplit == 1) {
            it.c0 = 0, it.cnt = nch;
        } else {  // K slice ks of ksplit: [nch*ks/ksplit, nch*(ks+1)/ksplit)
            it.c0 = (nch * it.ks) / h.ksplit;
            it.cnt = (nch * (it.ks + 1)) / h.ksplit - it.c0;
        }
        return it;
    };
    // SPAN: rows m0 .. m0 + 63 of a tile = n0 pixels at the end of output row t0 (= s * Ho + oy) + 64 - n0 at the start of row t0 + 1;
    // their input runs (x' = x + 2, so that the left padding starts at x' = 0) hold c0 = 2 n0 + 6 and 2 (64 - n0) + 6 pixel slots
    struct SpanGeo {
        int t0, ox0, n0, c0, c1;
    };
    auto span_geo = [&](int m0) __attribute__((always_inline)) {
        SpanGeo g;
        g.t0 = __builtin_amdgcn_readfirstlane(fdiv(m0, h.mg_wo, h.Wo)), g.ox0 = m0 - g.t0 * h.Wo;
        g.n0 = h.Wo - g.ox0 < 64 ? h.Wo - g.ox0 : 64;
        g.c0 = 2 * g.n0 + 6, g.c1 = g.n0 < 64 ? 2 * (64 - g.n0) + 6 : 0;
        return g;
    };
    int G = 0;  // chunks in this workgroup's stream
    if (SINGLE) G = nch;
    else if (h.ksplit == 1) G = my_n * nch;
    else
        for (int j = 0; j < my_n; j++) G += decode(j).cnt;
    const int pix = a.pixmode;
    // the producers' own argument group.  A one-item launch (SINGLE) has the registers to fetch it together with the common group above --
    // ONE scalar-cache round trip in front of the cold start instead of two; the streaming forms pin it inside the producers' branch
    struct Prod {
        const float *in, *w;
        int H, W, Cs, K, stride, tap_bias;
        unsigned mg_cpt;
        long long w_phase_stride;
        unsigned long long dy_pack, dx_pack;
    };
    auto load_prod = [&]() __attribute__((always_inline)) {
        Prod q = {a.in, a.w, a.H, a.W, a.Cs, a.K, a.stride, a.tap_bias, a.mg_cpt, a.w_phase_stride, a.dy_pack, a.dx_pack};
        asm volatile("" : "+s"(q.in), "+s"(q.w), "+s"(q.H), "+s"(q.W), "+s"(q.Cs), "+s"(q.K), "+s"(q.stride), "+s"(q.tap_bias),
                     "+s"(q.mg_cpt), "+s"(q.w_phase_stride), "+s"(q.dy_pack), "+s"(q.dx_pack));
        return q;
    };
    Prod p0 = {};
    if constexpr (SINGLE) p0 = load_prod();

    if (producer) {
        // ---- producer waves: LDS-DMA issue NS-1 chunks ahead of the consumers, across item boundaries -------
        // The loop that feeds the ring must not slow the matrix pipe it shares a SIMD with, so it is written for
        // instruction count: `buffer_load_dwordx4 ... offen lds` takes a 32-bit per-lane offset that is CONSTANT for a
        // whole tap (the lane's input pixel + 16-byte unit) plus a SCALAR offset (filter tap + channel chunk), i.e. a
        // chunk costs 4 DMA instructions and a few scalar adds -- no vector address arithmetic, no zero-page select.
        // Padded taps and rows past M use the buffer bounds check: such a lane's offset is 2^31 (>= num_records), the
        // hardware fetches nothing and writes zeros.  Negative tap offsets are folded into the descriptor's base
        // (in - tap_bias), so the scalar offset is >= 0 (rt_plan.cpp).
        __builtin_amdgcn_s_setprio(3);
        Prod p = p0;
        if constexpr (!SINGLE) p = load_prod();
        // tap entry e = phase*ntaps + tap: (dy, dx) from the packed table, byte offset from the lane's input pixel
        auto tap_dy = [&](int e) __attribute__((always_inline)) { return (int)((p.dy_pack >> (4 * e)) & 15) - 8; };
        auto tap_dx = [&](int e) __attribute__((always_inline)) { return (int)((p.dx_pack >> (4 * e)) & 15) - 8; };
        const srd_t srdA = make_srd((const char*)p.in - p.tap_bias), srdB = make_srd(p.w);
        const int srow = tid >> 3;
        const int unit = (tid & 7) ^ ((tid >> 4) & 7);  // source unit for LDS slot (row 32i + srow, unit tid&7)
        unsigned a_vo[ARB], a_mask[ARB], a_cur[ARB], b_vo[BRB], b_vo3[3] = {0, 0, 0};
        unsigned soA = 0, soB = 0;
        int tb = 0;  // first tap-table entry of the item's phase (indexing the argument block directly keeps it in constant memory)
        int tap = 0, cc = 0, rem = 0, jn = 0;
        auto set_tap = [&](int t) __attribute__((always_inline)) {
            const int e = __builtin_amdgcn_readfirstlane(tb + t);
            soA = (unsigned)__builtin_amdgcn_readfirstlane((tap_dy(e) * p.W + tap_dx(e)) * p.Cs * ESZ + p.tap_bias + cc * (128 * KG));
#pragma unroll
            for (int i = 0; i < ARB; i++) a_cur[i] = a_vo[i] | ((~(a_mask[i] >> t) & 1u) << 31);  // bit 31 set = out of bounds (a select here becomes a divergent branch)
        };
        // an item's set-up in two halves: the weight side needs the item's column block and phase only, the activation side the per-lane
        // pixel decode and tap masks.  A one-item launch requests its first two chunks' WEIGHTS between the two (see the start-up below).
        auto item_A = [&](const Item& it) __attribute__((always_inline)) {
            tb = __builtin_amdgcn_readfirstlane(it.phase * h.ntaps);
            if constexpr (SPAN) {
                // lane -> pixel slot 64 wave + lane of the tile's run(s); slots behind the runs (and all of wave 3) land zeros
                const SpanGeo g = span_geo(it.m0);
                const int slot = wave * 64 + lane;
                const int grow = slot < g.c0 ? g.t0 : g.t0 + 1;                  // output row s * Ho + oy the slot belongs to
                const int xp = slot < g.c0 ? 2 * g.ox0 + slot : slot - g.c0;     // x + 2 of the slot's input pixel
                const int sI = fdiv(grow, h.mg_ho, h.Ho), oy = grow - sI * h.Ho;
                const bool live = slot < g.c0 + g.c1 && grow * h.Wo < h.M;       // (the last tile may run past the last image)
                a_vo[0] = 0, a_mask[0] = 0;
#pragma unroll
                for (int i = 1; i < ARB; i++) a_vo[i] = 0, a_mask[i] = 0;
                if (live) {
                    const int iy = oy * p.stride, ix = xp;
                    a_vo[0] = (unsigned)(((sI * p.H + iy) * p.W + ix) * p.Cs * ESZ);
                    for (int t2 = 0; t2 < h.ntaps; t2++) {
                        const int y = iy + tap_dy(tb + t2), x = ix + tap_dx(tb + t2);
                        a_mask[0] |= ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W ? 1u : 0u) << t2;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < (SPAN ? 0 : ARB); i++) {
                // Written without branches for the 1x1 and 3x3 grids (every layer but conv1 and the transposed conv): a row past M decodes
                // pixel 0 and gets an empty mask (all its taps out of bounds), the grid is chosen by a select.  With `if (m < M)` and an
                // if-chain over the grids the compiler emitted two exec-masked blocks one after the other, four taken branches each --
                // on the one cold pass that stands between a launch and its first DMA.
                const int mraw = it.m0 + srow + 32 * i;
                const bool live = mraw < h.M;
                const int m = live ? mraw : 0;
                const int t = fdiv(m, h.mg_wo, h.Wo), ox = m - t * h.Wo;
                const int s = fdiv(t, h.mg_ho, h.Ho), oy = t - s * h.Ho;
                // conv1 (pixmode): a 16-B unit is one NHWC4 pixel (fp32) or two pixels (bf16; units 4-7 are the next image row)
                const int iy = oy * p.stride + (pix && BF ? unit >> 2 : 0);
                const int ix = ox * p.stride + (pix ? (BF ? (unit & 3) * 2 : unit) : 0);
                a_vo[i] = live ? (unsigned)(((s * p.H + iy) * p.W + ix) * p.Cs * ESZ + (pix ? 0 : unit * 16)) : 0u;
                // bit t2: tap t2 reads inside the image.  1x1 and 3x3 (pad 1) grids in closed form; any other tap
                // table (7x7 rows of conv1, the transposed conv's phases) by walking it
                const unsigned vx = (ix >= 1 ? 1u : 0u) | 2u | (ix + 1 < p.W ? 4u : 0u);
                const unsigned m3 = (iy >= 1 ? vx : 0u) | (vx << 3) | (iy + 1 < p.H ? vx << 6 : 0u);
                unsigned mk = a.tapgrid == 1 ? 1u : m3;
                if (a.tapgrid == 0) {  // (wave-uniform)
                    mk = 0;
                    for (int t2 = 0; t2 < h.ntaps; t2++) {
                        const int y = iy + tap_dy(tb + t2), x = ix + tap_dx(tb + t2);
                        mk |= ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W ? 1u : 0u) << t2;
                    }
                }
                a_mask[i] = live ? mk : 0u;
            }
            tap = __builtin_amdgcn_readfirstlane(fdiv(it.c0, p.mg_cpt, h.cpt)), cc = __builtin_amdgcn_readfirstlane(it.c0 - tap * h.cpt);
            rem = __builtin_amdgcn_readfirstlane(it.cnt);
            set_tap(tap);
        };
        auto item_B = [&](const Item& it) __attribute__((always_inline)) {
            if constexpr (X3) {
                // A step's weights are KG x 3 planes x BN rows x 64 B = 12 blocks of 16 rows: wave w lands blocks 3 w .. 3 w + 2.  Block ->
                // (K group, plane, 16-row group); LDS slot (row, 16-byte unit lane & 3) takes source unit (lane & 3) ^ ((row >> 2) & 3) --
                // the XOR that spreads a fragment read's 16 rows x 64 B over all banks
                constexpr int PER_KG = 3 * BN / 16, RG = BN / 16;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int blk = 3 * wave + j, kgi = blk / PER_KG, rr = blk % PER_KG, pl = rr / RG, rg = rr % RG;
                    const int brow = rg * 16 + (lane >> 2), bun = (lane & 3) ^ ((brow >> 2) & 3);
                    // global layout [chunk][plane][Npad rows][64 B]: the block's 16 rows are one contiguous KiB
                    b_vo3[j] = (unsigned)(((kgi * 3 + pl) * a.Npad + it.n0 + brow) * 64 + bun * 16);
                }
                soB = (unsigned)__builtin_amdgcn_readfirstlane((int)(it.phase * p.w_phase_stride * 6 + (long long)it.c0 * KG * (192 * a.Npad)));
            } else {
#pragma unroll
                for (int i = 0; i < BRB; i++) b_vo[i] = (unsigned)(((it.n0 + srow + 32 * i) * p.K + unit * EPU) * ESZ);
                soB = (unsigned)__builtin_amdgcn_readfirstlane((int)((it.phase * p.w_phase_stride + (long long)it.c0 * (EPR * KG)) * ESZ));
            }
        };
        auto begin_item = [&](int j) __attribute__((always_inline)) {
            const Item it = decode(j);
            item_B(it);
            item_A(it);
        };
        constexpr unsigned STEP_B = X3 ? 0u : 128u * KG;  // (X3: 192 * KG * Npad, a run-time step)
        // the LDS-DMA instructions of one chunk: activations (DO_A) and / or weights (DO_B; offB: the weights of a chunk further on)
        auto loads = [&](int stage, auto DO_A, auto DO_B, unsigned offB) __attribute__((always_inline)) {
            // stage base and scalar offsets are wave-uniform: say so at the use (if register pressure ever pushes this state
            // machine into VGPRs -- the PROF = 2 build did -- each DMA would otherwise be wrapped in a readfirstlane waterfall loop)
            float* sb = smem + __builtin_amdgcn_readfirstlane(stage) * STAGE + wave * (8 * 32);  // the hardware adds lane * 16 B
            const unsigned uA = (unsigned)__builtin_amdgcn_readfirstlane((int)soA), uB = (unsigned)__builtin_amdgcn_readfirstlane((int)(soB + offB));
#pragma unroll
            for (int k = 0; k < KG; k++) {  // K group k: the k-th 128-byte run of the step, landed in its own image
                if constexpr (decltype(DO_A)::value) {
#pragma unroll
                    for (int i = 0; i < (SPAN ? 1 : ARB); i++) bload_lds(srdA, sb + k * SUB + i * (32 * 32), a_cur[i], uA + k * 128);
                }
                if constexpr (!X3 && decltype(DO_B)::value) {
#pragma unroll
                    for (int i = 0; i < BRB; i++) bload_lds(srdB, sb + k * SUB + BM * 32 + i * (32 * 32), b_vo[i], uB + k * 128);
                }
            }
            if constexpr (X3 && decltype(DO_B)::value) {
                constexpr int PER_KG = 3 * BN / 16, RG = BN / 16;
                float* s0 = smem + __builtin_amdgcn_readfirstlane(stage) * STAGE;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int blk = 3 * wave + j, kgi = blk / PER_KG, rr = blk % PER_KG, pl = rr / RG, rg = rr % RG;  // wave-uniform
#if X3_DBG == 3  // timing probe (wrong results): the hi plane's bytes only (OOB lanes fetch nothing and land zeros)
                    bload_lds(srdB, s0 + kgi * SUB + BM * 32 + pl * (BN * 16) + rg * 256, pl == 0 ? b_vo3[j] : 0x80000000u, uB);
                    continue;
#endif
                    bload_lds(srdB, s0 + kgi * SUB + BM * 32 + pl * (BN * 16) + rg * 256, b_vo3[j], uB);
                }
            }
        };
        auto advance = [&]() __attribute__((always_inline)) {
            rem = __builtin_amdgcn_readfirstlane(rem - 1);
            if (rem == 0) {
                if constexpr (!SINGLE) {  // (a one-item launch has nothing behind its last chunk)
                    jn = __builtin_amdgcn_readfirstlane(jn + 1);
                    if (jn < my_n) begin_item(jn);
                }
            } else {
                soA += 128 * KG, soB += X3 ? 192 * KG * a.Npad : 128 * KG;
                if (++cc == h.cpt) cc = 0, set_tap(++tap);
            }
        };
        auto issue = [&](int stage) __attribute__((always_inline)) {
            loads(stage, std::true_type{}, std::true_type{}, 0u);
            advance();
        };
        auto wait_landed = [&](int young) __attribute__((always_inline)) {  // all but the `young` youngest chunks (NLD instructions each) have landed
            switch (young) {
                case 1: wait_vm<NLD>(); break;
                case 2: wait_vm<2 * NLD>(); break;
                case 3: wait_vm<3 * NLD>(); break;
                case 4: wait_vm<4 * NLD>(); break;
                case 5: wait_vm<5 * NLD>(); break;
                case 6: wait_vm<6 * NLD>(); break;
                case 7: wait_vm<7 * NLD>(); break;
                default: wait_vm<0>(); break;
            }
        };
        // TAIL: this wave's share of the second GEMM is column blocks 2 + g, 4 + g, 6 + g, ... of row half wm2; the shortcut values
        // of the first two and the weight fragments of the first are requested NOW (older than every LDS-DMA below, so the
        // counted vmcnt waits of the ring are unaffected) and rest in registers this wave does not otherwise need.
        TailRegs<BF> T;
        WideRegs<BF> TW;
        const int wm2 = wave & 1, g2 = wave >> 1;
        if constexpr (TAIL && WIDE) wide_load_weights<BF>(a, wave, lane, TW), wide_load_resid<BF>(a, decode(0).m0, wave, lane, TW);  // tail waves 0..3
        if constexpr (TAIL && !WIDE) {
            const int mb2 = decode(0).m0 + wm2 * 32 + 4 * (lane >> 5);
            tail_load_resid<BF>(a, mb2, 2 + g2, lane, T.rw[0]);
            tail_load_resid<BF>(a, mb2, 4 + g2, lane, T.rw[1]);
            tail_load_b<BF>(a, 2 + g2, lane, T.Bf);
        }
        const bool pst = P2 && threadIdx.x == 256 && blockIdx.x == 0;
        if (pst) prof[22] = __builtin_amdgcn_s_memrealtime();  // producer wave 0: arguments pinned, about to decode the first item
        // The consumers can start as soon as chunk 0 has landed, so only chunks 0 and 1 are requested before the first
        // barrier; the rest of the ring is filled behind it, two chunks per iteration until NS-1 are in flight.  (Filling
        // the whole ring first kept the matrix pipes waiting for 0.6-1.5 us of producer bookkeeping per launch.)
        int nis = 0, istage = 0, g = 0;  // chunks issued; stage of the next chunk to issue
        auto put = [&]() __attribute__((always_inline)) {
            issue(istage);
            istage = istage + 1 == NS ? 0 : istage + 1, nis++;
        };
        constexpr bool BFIRST = SINGLE && !X3;
        if constexpr (BFIRST) {
            // one-item launch: the WEIGHTS of chunks 0 and 1 are requested as soon as the item is known -- their addresses need no pixel
            // decode -- so their way from the Infinity Cache overlaps the ~100 instructions of the activation side's set-up.  Request
            // order B0 B1 A0 A1: chunk 0 is complete once all but A1's instructions have landed.
            const Item it = decode(0);
            item_B(it);
            loads(0, std::false_type{}, std::true_type{}, 0u);
            if (G > 1) loads(1, std::false_type{}, std::true_type{}, STEP_B);
            item_A(it);
            if (pst) prof[23] = __builtin_amdgcn_s_memrealtime();  // first item decoded
            loads(0, std::true_type{}, std::false_type{}, 0u);
            advance();
            if (G > 1) {
                loads(1, std::true_type{}, std::false_type{}, 0u);
                advance();
            }
            nis = G > 1 ? 2 : 1, istage = nis;  // (NS >= 3)
            if (pst) prof[14] = __builtin_amdgcn_s_memrealtime();  // chunks 0 and 1 requested
            if (G > 1) wait_vm<KG * (SPAN ? 1 : ARB)>();
            else wait_vm<0>();
        } else {
            begin_item(0);
            if (pst) prof[23] = __builtin_amdgcn_s_memrealtime();  // first item decoded
            put();
            if (G > 1) put();
            if (pst) prof[14] = __builtin_amdgcn_s_memrealtime();  // chunks 0 and 1 requested
            wait_landed(nis - 1);
        }
        __builtin_amdgcn_s_barrier();  // chunk 0 visible
        if (G > 2) put();
        const bool pacct = P2 && threadIdx.x == 256 && blockIdx.x == 0;  // tuning aid: where producer wave 0 spends its time
        unsigned pw = 0, pb = 0, pi = 0;  // 32-bit cycle counts (differences of the low halves of s_memtime): the tuning twin is short of SGPRs
        // one barrier per chunk, also after the last one (keeps the consumer loop branch-free).  Steady state: chunk g+1
        // complete in LDS, chunks g+2 .. g+NS-2 still in flight, chunk g+NS-1 issued behind the barrier that retires
        // stage g-1.
        for (; g < G; g++) {
            const unsigned q0 = P2 ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
            const int young = nis - g - 2;  // chunks allowed to be still in flight once chunk g+1 has landed
            if (young == NS - 3) wait_vm<(NS - 3) * NLD>();
            else wait_landed(young > 0 ? young : 0);
            const unsigned q1 = P2 ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
            __builtin_amdgcn_s_barrier();  // consumers are past chunk g-1: its stage may be refilled
            const unsigned q2 = P2 ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
            const int target = G < g + NS ? G : g + NS;
            while (nis < target) put();
            const unsigned q3 = P2 ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
            pw += q1 - q0, pb += q2 - q1, pi += q3 - q2;
        }
        if (pacct) prof[16] = pw, prof[17] = pb, prof[18] = pi;
        if constexpr (TAIL) {
            __builtin_amdgcn_s_barrier();  // the consumers have put the layer's tile into LDS (one tile per workgroup)
            if constexpr (WIDE) {
                f32x4 Cq[CHAIN_DEPTH<BF>];
                tail_wide<BF, CHAIN>(a, smem, decode(0).m0, wave, lane, TW, Cq);
                if constexpr (CHAIN) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();  // all eight tail waves have put their blocks of the output tile into LDS
                    if constexpr (BF) {
                        if (wide_staged<BF>(a)) wide_write_out(a, smem, decode(0).m0);
                    }
                    chain_gemm<BF>(a, smem, decode(0).m0, wave, lane, Cq);
                } else if constexpr (BF) {
                    if (wide_staged<BF>(a)) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        wide_write_out(a, smem, decode(0).m0);
                    }
                }
            } else {
                tail_gemm<BF, 3, true, CHAIN>(a, smem, decode(0).m0, wm2, 2 + g2, 2, lane, T);
                if constexpr (CHAIN) chain_narrow(a, smem, decode(0).m0, wave, lane);
                else if constexpr (BF) {
                    if (tail_staged<BF>(a)) {  // (uniform) the block output leaves from the LDS tile: all eight waves, behind one barrier
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        tail_write_out(a, smem, decode(0).m0);
                    }
                }
            }
        }
        if constexpr (BONE) {
            const Item it0 = decode(0);
            if (it0.n0 <= BONE_N0 && it0.n0 + BN > BONE_N0) {  // the tile that holds the 63 delta columns (uniform per workgroup)
                __builtin_amdgcn_s_barrier();  // ... is in LDS now
                bone_features<BF>(a, smem, it0.m0, it0.phase, tid, h.M, h.Wo, h.Ho, h.mg_wo, h.mg_ho);
            }
        }
        return;
    }

    // ---- consumer waves ------------------------------------------------------------------------------------
    // fragment addresses: row lane&31 of the wave tile, unit (2q + h) ^ ((row >> 1) & 7)
    int fo[4];
#pragma unroll
    for (int q = 0; q < 4; q++) fo[q] = (lane & 31) * 32 + (((2 * q + (lane >> 5)) ^ ((lane >> 1) & 7)) * 4);
    // SPAN: A fragments come from the tile's pixel run(s): row R of the tile starts at slot 2 R of the first run or, behind the
    // row's end, at slot c0 + 2 (R - n0) of the second; fragment q of lane half h is pixel slot + 2 q + h (16 bytes = 4 floats)
    int foA[4] = {0, 0, 0, 0};
    auto set_span = [&](int m0) __attribute__((always_inline)) {
        const SpanGeo g = span_geo(m0);
        const int R = wm * 32 + (lane & 31);
        const int slot = R < g.n0 ? 2 * R : g.c0 + 2 * (R - g.n0);
#pragma unroll
        for (int q = 0; q < 4; q++) foA[q] = (slot + 2 * q + (lane >> 5)) * 4;
    };
    struct Frag {
        f32x4 a[4], b[4];
        f32x4 bx[NACC > 1 ? NACC - 1 : 1][4];  // NACC > 1: the B fragments of the wave's column blocks 1 .. NACC - 1
    };
    Frag F0, F1;
    f32x16 acc;
    f32x16 accx[NACC > 1 ? NACC - 1 : 1];  // NACC > 1: the accumulators of column blocks 1 .. NACC - 1 (block 0 is `acc`)
    WideRegs<BF> TWc;  // (the wide tail's operands of this consumer wave)
    auto rall = [&](int stg, Frag& F) __attribute__((always_inline)) {
        const float* Ab = smem + stg * STAGE + kg * SUB + (SPAN ? 0 : (wm * 32) * 32);
        const float* Bb = smem + stg * STAGE + kg * SUB + (BM + wn * 32) * 32;
#pragma unroll
        for (int q = 0; q < 4; q++) F.a[q] = *(const f32x4*)(Ab + (SPAN ? foA[q] : fo[q])), F.b[q] = *(const f32x4*)(Bb + fo[q]);
        if constexpr (NACC > 1) {
#pragma unroll
            for (int b = 1; b < NACC; b++)
#pragma unroll
                for (int q = 0; q < 4; q++) F.bx[b - 1][q] = *(const f32x4*)(Bb + b * 1024 + fo[q]);
        }
    };
    auto mma = [&](const f32x4& af, const f32x4& bf) __attribute__((always_inline)) {
        if constexpr (BF) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
        }
    };
    int stage = 0;
    unsigned cbw = 0;  // tuning twin: cycles consumer wave 0 waits at the per-chunk barrier
    // One chunk = 16 dependent MFMAs (fp32).  The chunk opens with the barrier that publishes chunk g+1 (and tells the
    // producers this wave is past chunk g-1: its operands of chunk g are already in registers); the 8 fragment reads of
    // chunk g+1 then go out behind every second MFMA.  A wave issues in order, so a read that has to queue at the LDS
    // delays the next MFMA of the chain: spread thin they never do (tools/ring_rate.hip: 92 % of the MFMA bound against
    // 88 % with the reads packed behind MFMAs 8-15, one workgroup per CU).
    auto step = [&](Frag& cur, Frag& nxt) __attribute__((always_inline)) {
        const int nstage = stage + 1 == NS ? 0 : stage + 1;
        if constexpr (P2) {
            const unsigned b0 = (unsigned)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            cbw += (unsigned)__builtin_amdgcn_s_memtime() - b0;
        } else
            __builtin_amdgcn_s_barrier();  // chunk g+1 visible; every consumer is past chunk g-1
        if constexpr (SPAN) {
            // 12 MFMAs per chunk (e = 3 is the zero padding channel); the next chunk's 8 fragment reads behind MFMAs 2, 3 of each group
            const float* Ab = smem + nstage * STAGE;
            const float* Bb = smem + nstage * STAGE + (BM + wn * 32) * 32;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[q][e], cur.b[q][e], acc, 0, 0, 0);
                    if (e == 1) {
                        nxt.a[q] = *(const f32x4*)(Ab + foA[q]);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    } else if (e == 2) {
                        nxt.b[q] = *(const f32x4*)(Bb + fo[q]);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
        } else if constexpr (!BF && NACC > 1) {
            // NACC accumulators per wave: every A value feeds NACC MFMAs (independent chains, so the pipe never waits on a dependency);
            // the next chunk's 4 + 4 NACC fragment reads go out one per MFMA group
            const float* Ab = smem + nstage * STAGE + kg * SUB + (wm * 32) * 32;
            const float* Bb = smem + nstage * STAGE + kg * SUB + (BM + wn * 32) * 32;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[q][e], cur.b[q][e], acc, 0, 0, 0);
#pragma unroll
                    for (int b = 1; b < NACC; b++) accx[b - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[q][e], cur.bx[b - 1][q][e], accx[b - 1], 0, 0, 0);
                    const int r = q * 4 + e;  // 0..15: A fragments 0..3, then the B fragments of blocks 0 .. NACC - 1
                    if (r < 4) nxt.a[r] = *(const f32x4*)(Ab + fo[r]);
                    else if (r < 8) nxt.b[r - 4] = *(const f32x4*)(Bb + fo[r - 4]);
                    else if (r - 8 < 4 * (NACC - 1)) nxt.bx[(r - 8) >> 2][(r - 8) & 3] = *(const f32x4*)(Bb + (((r - 8) >> 2) + 1) * 1024 + fo[(r - 8) & 3]);
                    if (r < 4 + 4 * NACC) {
                        __builtin_amdgcn_sched_group_barrier(0x008, NACC, 0);  // NACC MFMAs ...
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // ... then one DS read
                    }
                }
        } else if constexpr (!BF) {
            const float* Ab = smem + nstage * STAGE + kg * SUB + (wm * 32) * 32;
            const float* Bb = smem + nstage * STAGE + kg * SUB + (BM + wn * 32) * 32;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[q][e], cur.b[q][e], acc, 0, 0, 0);
                    if (e & 1) {
                        const int r = q * 2 + (e >> 1);  // 0..7: fragment r>>1 of A (even r) or B (odd r)
                        if (r & 1) nxt.b[r >> 1] = *(const f32x4*)(Bb + fo[r >> 1]);
                        else nxt.a[r >> 1] = *(const f32x4*)(Ab + fo[r >> 1]);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // two MFMAs ...
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // ... then one DS read
                    }
                }
        } else {
            rall(nstage, nxt);
            __builtin_amdgcn_sched_barrier(0);
            mma(cur.a[0], cur.b[0]);
            mma(cur.a[1], cur.b[1]);
            mma(cur.a[2], cur.b[2]);
            mma(cur.a[3], cur.b[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        stage = nstage;
    };

    // ---- X3: the split-product K loop (see the kernel's head comment).  A chunk = two K steps of 16; the operands of a K step are
    // the lane's 8 fp32 activations (two 16-byte units of the A row) and 8 bf16 of each weight plane (one unit of the plane's row).
    // Software pipeline over K steps i = 2 chunk + s, three stages deep, register sets by parity of i:
    //     iteration i:  read raw A(i + 2) and B(i + 1) from LDS | split A(i + 1) into its three bf16 pieces (VALU) | 6 MFMAs of step i
    // so the split's ~44 VALU instructions sit BETWEEN the dependent MFMAs of the step before (a wave issues in order: left in front of
    // them they cost as much as the matrix work itself) and every LDS read has a whole K step to land.  Steps i + 1, i + 2 lie in
    // chunk g or g + 1, both visible since chunk g's barrier.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    struct Asp {
        u32x4 h, m, l;  // 8 bf16 each: hi, mid, lo pieces
    };
    f32x4 rawA[2][2] = {};
    f32x4 Bq[2][3] = {};
    Asp Aq[2] = {};
    int foA3[2][2] = {{0, 0}, {0, 0}}, foB3[2] = {0, 0};
    if constexpr (X3) {
        const int sw = (lane >> 1) & 7, swb = (lane >> 2) & 3;  // (row >> 1) & 7 of the A row, (row >> 2) & 3 of the B row: rows are 32 wm / wn + (lane & 31)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
#pragma unroll
            for (int j2 = 0; j2 < 2; j2++) foA3[s2][j2] = kg * SUB + (wm * 32 + (lane & 31)) * 32 + (((4 * s2 + 2 * (lane >> 5) + j2) ^ sw) * 4);
            foB3[s2] = kg * SUB + BM * 32 + (wn * 32 + (lane & 31)) * 16 + (((2 * s2 + (lane >> 5)) ^ swb) * 4);
        }
    }
    auto ldA = [&](int par, int stg, int s2) __attribute__((always_inline)) {
        const float* sb = smem + stg * STAGE;
        rawA[par][0] = *(const f32x4*)(sb + foA3[s2][0]), rawA[par][1] = *(const f32x4*)(sb + foA3[s2][1]);
    };
    auto ldB = [&](int par, int stg, int s2) __attribute__((always_inline)) {
        const float* sb = smem + stg * STAGE + foB3[s2];
#pragma unroll
        for (int pl = 0; pl < 3; pl++) Bq[par][pl] = *(const f32x4*)(sb + pl * (BN * 16));
    };
    // pieces of elements 2 q, 2 q + 1 of the raw A set `par`: x = hi + mid + lo by truncation (exact): hi = top 16 bits of x, mid = top 16
    // bits of x - hi, lo = top 16 bits of x - hi - mid; two bf16 per dword (the high halves of elements 2 q + 1 and 2 q).  11 VALU ops.
    auto split2 = [&](int par, int q) __attribute__((always_inline)) {
        unsigned hb[2], mb[2], lb[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float x = rawA[par][q >> 1][2 * (q & 1) + e];
            hb[e] = __builtin_bit_cast(unsigned, x);
#if X3_DBG == 1  // timing probe (wrong results): no split arithmetic
            mb[e] = hb[e], lb[e] = hb[e];
#else
            const float r1 = x - __builtin_bit_cast(float, hb[e] & 0xffff0000u);
            mb[e] = __builtin_bit_cast(unsigned, r1);
            lb[e] = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, mb[e] & 0xffff0000u));
#endif
        }
        Aq[par].h[q] = __builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u);
        Aq[par].m[q] = __builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u);
        Aq[par].l[q] = __builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u);
    };
    auto splitA = [&](int par) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; q++) split2(par, q);
    };
    auto kstep = [&](int par, int stgA, int sA, int stgB, int sB) __attribute__((always_inline)) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, Aq[par].h), am = __builtin_bit_cast(bf16x8, Aq[par].m), al = __builtin_bit_cast(bf16x8, Aq[par].l);
        const bf16x8 wh = __builtin_bit_cast(bf16x8, Bq[par][0]), wmid = __builtin_bit_cast(bf16x8, Bq[par][1]), wl = __builtin_bit_cast(bf16x8, Bq[par][2]);
        ldA(par, stgA, sA);        // raw A of step i + 2
        ldB(par ^ 1, stgB, sB);    // weights of step i + 1
        __builtin_amdgcn_sched_barrier(0);
        // the six (dependent) MFMAs of step i, smallest terms first, with the split of step i + 1's activations between them: a wave
        // issues in order, so VALU work placed in front of (or behind) the chain would cost as much time as the matrix work itself.
        // Hard scheduling fences: the hints (sched_group_barrier) were honoured in one of the two unrolled K steps only.
#if X3_DBG == 2  // timing probe (wrong results): half the matrix work
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wmid, acc, 0, 0, 0);
        split2(par ^ 1, 0);
        split2(par ^ 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wl, acc, 0, 0, 0);
        split2(par ^ 1, 2);
        split2(par ^ 1, 3);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc, 0, 0, 0);
#else
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wmid, acc, 0, 0, 0);
        split2(par ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc, 0, 0, 0);
        split2(par ^ 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc, 0, 0, 0);
        split2(par ^ 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wh, acc, 0, 0, 0);
        split2(par ^ 1, 3);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wmid, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc, 0, 0, 0);
#endif
    };
    auto step3 = [&]() __attribute__((always_inline)) {
        const int nstage = stage + 1 == NS ? 0 : stage + 1;
        __builtin_amdgcn_s_barrier();  // chunk g+1 visible; every consumer is past chunk g-1
        kstep(0, nstage, 0, stage, 1);
        __builtin_amdgcn_sched_barrier(0);
        kstep(1, nstage, 1, nstage, 0);
        __builtin_amdgcn_sched_barrier(0);
        stage = nstage;
    };

    struct Cons {
        const float *bias, *scale, *shift, *resid;
        float *out, *out2, *ws;
        long long slab_pix;
        int Nvalid, Npad, ldr, ldc, ldc2, split_n, relu_cols, out_f32, os, OH, OW;
    } c = {a.bias, a.scale, a.shift, a.resid, a.out, a.out2, a.ws, a.slab_pix, a.Nvalid, a.Npad, a.ldr, a.ldc, a.ldc2, a.split_n,
           a.relu_cols, a.out_f32, a.os, a.OH, a.OW};
    asm volatile("" : "+s"(c.bias), "+s"(c.scale), "+s"(c.shift), "+s"(c.resid), "+s"(c.out), "+s"(c.out2), "+s"(c.ws), "+s"(c.Nvalid),
                 "+s"(c.Npad), "+s"(c.ldr), "+s"(c.ldc), "+s"(c.ldc2), "+s"(c.split_n), "+s"(c.relu_cols), "+s"(c.out_f32), "+s"(c.os),
                 "+s"(c.OH), "+s"(c.OW), "+s"(c.slab_pix));
    const bool fused = SINGLE || h.ksplit == 1;
    const bool direct = (c.os == 1);
    const int col = lane & 31, rhalf = 4 * (lane >> 5);  // C/D map: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    cgfloat* resid = fused && !TAIL ? (cgfloat*)c.resid : nullptr;  // (a tail layer's shortcut belongs to its second GEMM)
    if (pstamp) prof[10] = __builtin_amdgcn_s_memrealtime();
    if constexpr (KG > 1) {
        if (threadIdx.x < (KG - 1) * WMN) ((volatile int*)(smem + SCRATCH + NPART * 1024))[threadIdx.x] = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // chunk 0 visible
    if (pstamp) prof[11] = __builtin_amdgcn_s_memrealtime();
    if constexpr (SPAN) set_span(decode(0).m0);
    if constexpr (X3) {  // pipeline prologue: step 0's weights and pieces, step 1's raw A (chunk 0 is visible)
        ldA(0, 0, 0), ldB(0, 0, 0), ldA(1, 0, 1);
        splitA(0);
    } else
        rall(0, F0);
    for (int j = 0; j < my_n; j++) {
        const Item it = decode(j);
        if constexpr (SPAN) {
            if (j > 0) {  // the A fragments prefetched behind the previous tile's last chunk used that tile's run geometry: read them
                          // again (the stage is not retired before this wave has passed two more chunk barriers)
                set_span(it.m0);
                const float* Ab = smem + stage * STAGE;
#pragma unroll
                for (int q = 0; q < 4; q++) F0.a[q] = *(const f32x4*)(Ab + foA[q]);
            }
        }
        const int n = it.n0 + wn * 32 + col;
        const int mb = it.m0 + wm * 32 + rhalf;
        // bias and shortcut of this tile: requested now so the epilogue never waits for them.  (Straight-line code
        // here and in the epilogue runs once per tile on a cold instruction cache -- ~35 cycles per instruction
        // measured -- so both are written for instruction count: scalar row bases stepped by additions, one lane offset.)
        float bias = 0.f, sc = 1.f, sh = 0.f;
        if constexpr (TAIL && WIDE) wide_load_resid<BF>(a, it.m0, 4 + wave, lane, TWc);  // the tail's shortcut values: requested before the K loop
        unsigned rsw[16];  // shortcut values as loaded (fp32 bits, or a zero-extended bf16): converting a bf16 here would make
                           // the compiler wait for the loads BEFORE the K loop (measured: 15 us per 92x92 shortcut layer)
        float biasx[NACC > 1 ? NACC - 1 : 1] = {}, scx[NACC > 1 ? NACC - 1 : 1] = {}, shx[NACC > 1 ? NACC - 1 : 1] = {};  // column blocks 1 .. NACC - 1
        if (fused && kg == 0) {
            bias = ((cgfloat*)c.bias)[n];
            if (c.scale) sc = ((cgfloat*)c.scale)[n], sh = ((cgfloat*)c.shift)[n];
            if constexpr (NACC > 1) {
#pragma unroll
                for (int b = 1; b < NACC; b++) {
                    biasx[b - 1] = ((cgfloat*)c.bias)[n + 32 * b], scx[b - 1] = 1.f;
                    if (c.scale) scx[b - 1] = ((cgfloat*)c.scale)[n + 32 * b], shx[b - 1] = ((cgfloat*)c.shift)[n + 32 * b];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) rsw[r] = 0u;
        if (resid && kg == 0) {
            // shortcut layers (os == 1).  Row r of the C layout is a UNIFORM distance from the lane's first row, so the
            // 16 requests share one 32-bit lane offset and differ in a scalar base: no address registers, no branches
            // (a branch per load would serialise the requests).  Rows past M fall into the tensors' 64-pixel slack
            // (rt_plan.cpp), columns past Nvalid re-read column 0; both are masked at the store.
            const unsigned off = (unsigned)(mb * c.ldr + (n < c.Nvalid ? n : 0));
            if constexpr (BF) {
                typedef __attribute__((address_space(1))) const unsigned short cgu16;
                cgu16* rp = (cgu16*)resid;
#pragma unroll
                for (int r = 0; r < 16; r++) rsw[r] = rp[off], rp += ((r & 3) == 3 ? 5 : 1) * c.ldr;
            } else {
                typedef __attribute__((address_space(1))) const unsigned cgu32;
                cgu32* rp = (cgu32*)resid;
#pragma unroll
                for (int r = 0; r < 16; r++) rsw[r] = rp[off], rp += ((r & 3) == 3 ? 5 : 1) * c.ldr;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        if constexpr (NACC > 1) {
#pragma unroll
            for (int b = 1; b < NACC; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) accx[b - 1][r] = 0.f;
        }
        if constexpr (X3) {
            for (int t = 0; t < it.cnt; t++) step3();
        } else {
            for (int t = 0; t < it.cnt; t += 2) {
                step(F0, F1);
                if (t + 1 < it.cnt) step(F1, F0);
            }
            if (it.cnt & 1) F0 = F1;  // odd chunk count: the fragments of the next item's first chunk sit in F1
        }
        if (pstamp && j == 0) prof[20] = __builtin_amdgcn_s_memrealtime();  // first item: K loop done
        if constexpr (KG > 1) {
            // K groups 1.. hand their accumulators to group 0 through LDS (lane-linear 16-byte slots: conflict-free) and go
            // on to the next item; group 0 adds them in group order.  The flag carries the item number, so nothing is
            // reset; a partial is not overwritten early because the writer first has to pass the next item's K-loop
            // barriers, which group 0 joins only after this epilogue.
            float* part = smem + SCRATCH;
            volatile int* flags = (volatile int*)(smem + SCRATCH + NPART * 1024);
            if (kg > 0) {
                float* dst = part + ((kg - 1) * WMN + wr) * NACC * 1024 + lane * 4;
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) *(f32x4*)(dst + r4 * 256) = f32x4{acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]};
                if constexpr (NACC > 1) {
#pragma unroll
                    for (int b = 1; b < NACC; b++)
#pragma unroll
                        for (int r4 = 0; r4 < 4; r4++)
                            *(f32x4*)(dst + b * 1024 + r4 * 256) = f32x4{accx[b - 1][4 * r4], accx[b - 1][4 * r4 + 1], accx[b - 1][4 * r4 + 2], accx[b - 1][4 * r4 + 3]};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) flags[(kg - 1) * WMN + wr] = j + 1;
                if constexpr (BONE) {  // (every wave of the workgroup meets at the bone stage's barrier, the K groups that hand their sums over too)
                    if (it.n0 <= BONE_N0 && it.n0 + BN > BONE_N0) __builtin_amdgcn_s_barrier();
                }
                continue;
            }
#pragma unroll
            for (int k = 1; k < KG; k++) {
                while (flags[(k - 1) * WMN + wr] != j + 1) __builtin_amdgcn_s_sleep(1);
                const float* src = part + ((k - 1) * WMN + wr) * NACC * 1024 + lane * 4;
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const f32x4 v = *(const f32x4*)(src + r4 * 256);
                    acc[4 * r4] += v[0], acc[4 * r4 + 1] += v[1], acc[4 * r4 + 2] += v[2], acc[4 * r4 + 3] += v[3];
                }
                if constexpr (NACC > 1) {
#pragma unroll
                    for (int b = 1; b < NACC; b++)
#pragma unroll
                        for (int r4 = 0; r4 < 4; r4++) {
                            const f32x4 v = *(const f32x4*)(src + b * 1024 + r4 * 256);
                            accx[b - 1][4 * r4] += v[0], accx[b - 1][4 * r4 + 1] += v[1], accx[b - 1][4 * r4 + 2] += v[2], accx[b - 1][4 * r4 + 3] += v[3];
                        }
                }
            }
        }

        if constexpr (BONE) {
            if (it.n0 <= BONE_N0 && it.n0 + BN > BONE_N0) {
                // the 63 delta columns of this row block (no bias, no BN on these columns: the accumulators ARE the values the
                // epilogue below stores) go to LDS for the producer waves, which have nothing left to do and compute the
                // bone lengths from them while this wave runs its ordinary epilogue
                if constexpr (NACC == 1) {
#pragma unroll
                    for (int r = 0; r < 16; r++) smem[(wm * 32 + rhalf + (r & 3) + 8 * (r >> 2)) * BONE_LS + wn * 32 + col] = acc[r];
                } else {  // the 96-wide tile 96 .. 191: its column blocks 1 and 2 are the delta columns 128 .. 191
#pragma unroll
                    for (int b = 1; b < NACC; b++) {
                        const int c0 = it.n0 + (wn + b) * 32 - BONE_N0;  // >= 0 for the blocks that hold delta columns (uniform)
                        if (c0 >= 0) {
#pragma unroll
                            for (int r = 0; r < 16; r++) smem[(wm * 32 + rhalf + (r & 3) + 8 * (r >> 2)) * BONE_LS + c0 + col] = accx[b - 1][r];
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        if constexpr (TAIL) {
            // ---- tail GEMM: out2[64][tail_n] = relu(acc + bias)[64][64] x tail_w^T, + tail_bias + shortcut, ReLU --------------
            // This workgroup has ONE tile (the host guarantees items <= grid), so its producers are done and the whole ring
            // is free: every consumer's last real fragment read happened before the final chunk barrier it has just passed.
            // (a) the tile goes to LDS as the A operand of the second GEMM (row stride 64 + pad: conflict-free 16-byte reads),
            // (b) all EIGHT waves meet at a barrier -- the producer waves stay for the tail: the second GEMM's blocks are chains of
            //     dependent memory round trips (weights, shortcut, stores), and eight waves keep twice as many in flight as four --
            // (c) the 2 x 8 blocks of 32 x 32 (K = 64 each) are shared out as tail_gemm describes: weight fragments straight from
            //     global memory (64 KB, L2-resident, read in the MFMA's own fragment layout), the next block's fragments in
            //     flight behind the current block's epilogue, the producers' shortcut values already in registers.
            // The K order of every output is that of the stand-alone 1x1 layer, and the tile in LDS holds exactly the values
            // that layer would have read back from HBM: results are bit-identical to the unfused plan.
            constexpr int MS = WIDE ? WIDE_MS<BF> : TAIL_MS<BF>;
            const int wm2 = wave & 1, g2 = wave >> 1;  // this wave's share of the tail: row half, column block
            if constexpr (WIDE) wide_load_weights<BF>(a, 4 + wave, lane, TWc);  // tail waves 4..7: in flight behind the tile's way to LDS and the producers' MFMAs
            {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias));  // the layer's own bias (requested before the K loop)
                const int k1 = wn * 32 + col;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = wm * 32 + rhalf + (r & 3) + 8 * (r >> 2);
                    const float v = __builtin_fmaxf(acc[r] + bias, 0.f);
                    if constexpr (BF) ((__bf16*)smem)[row * MS + k1] = (__bf16)v;
                    else smem[row * MS + k1] = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (WIDE) {  // (wm = 0: the tile's 32 rows; the four waves wrote columns 32 wn ..)
                tail_wide<BF, CHAIN>(a, smem, it.m0, 4 + wave, lane, TWc);
                if constexpr (CHAIN) {  // the chain GEMM belongs to the producer waves; this wave delivers its blocks and its share of the write-out
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if constexpr (BF) {
                        if (wide_staged<BF>(a)) wide_write_out(a, smem, it.m0);
                    }
                } else if constexpr (BF) {
                    if (wide_staged<BF>(a)) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        wide_write_out(a, smem, it.m0);
                    }
                }
            } else {
                TailRegs<BF> T;
                tail_gemm<BF, 1, false, CHAIN>(a, smem, it.m0, wm2, g2, 0, lane, T);
                if constexpr (CHAIN) {  // the chain GEMM belongs to the producer waves; this wave delivers its block and its share of the write-out
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    tail_write_out(a, smem, it.m0);
                } else if constexpr (BF) {
                    if (tail_staged<BF>(a)) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        tail_write_out(a, smem, it.m0);
                    }
                }
            }
            break;  // one tile per workgroup: nothing of the K-loop state (prefetched fragments, item bookkeeping) lives on
        }
        if constexpr (NACC > 1) {
            // ---- epilogue of the NACC-accumulator shape (fp32; no shortcut, one output tensor, no K slabs: launch_stream checks) ----
            // One pass over the lane's 16 rows: a row's output pixel -- two multiply-high divisions for the transposed conv's scatter
            // to (2 i + py, 2 j + px) -- is computed ONCE and serves the row's NACC stores (columns n, n + 32, ...).
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias), "+v"(sc), "+v"(sh), "+v"(biasx[0]), "+v"(scx[0]), "+v"(shx[0]), "+v"(biasx[NACC - 2]), "+v"(scx[NACC - 2]), "+v"(shx[NACC - 2]));
            const int py = it.phase >> 1, px = it.phase & 1;
            gfloat* const outp = (gfloat*)c.out;
            bool relub[NACC];
#pragma unroll
            for (int b = 0; b < NACC; b++) relub[b] = it.n0 + (wn + b) * 32 < c.relu_cols;  // uniform per block: relu_cols is a multiple of 32
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                int opix = m;
                if (!direct) {
                    const int t = fdiv(m, h.mg_wo, h.Wo), ox = m - t * h.Wo;
                    const int sI = fdiv(t, h.mg_ho, h.Ho), oy = t - sI * h.Ho;
                    opix = (sI * c.OH + oy * c.os + py) * c.OW + ox * c.os + px;
                }
                const unsigned off = (unsigned)(opix * c.ldc + n);
#pragma unroll
                for (int b = 0; b < NACC; b++) {
                    float o = (b == 0 ? acc[r] : accx[b > 0 ? b - 1 : 0][r]) + (b == 0 ? bias : biasx[b > 0 ? b - 1 : 0]);
                    if (c.scale) o = o * (b == 0 ? sc : scx[b > 0 ? b - 1 : 0]) + (b == 0 ? sh : shx[b > 0 ? b - 1 : 0]);
                    if (relub[b]) o = __builtin_fmaxf(o, 0.f);
                    if (m < h.M && n + 32 * b < c.Nvalid) put_f32(outp + off + 32 * b, o);
                }
            }
            if (pstamp && j == 0) prof[21] = __builtin_amdgcn_s_memrealtime();
            continue;
        }
        // epilogue from registers.  One explicit wait for the bias / shortcut values requested before the K loop, with
        // the values passed through it: otherwise the compiler re-waits (vmcnt(0)) for those loads before every use,
        // i.e. after every store below, and the 16 stores complete one by one (measured: 2.5-4.8 us per epilogue).
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(bias), "+v"(sc), "+v"(sh), "+v"(rsw[0]), "+v"(rsw[1]), "+v"(rsw[2]), "+v"(rsw[3]), "+v"(rsw[4]), "+v"(rsw[5]),
                       "+v"(rsw[6]), "+v"(rsw[7]), "+v"(rsw[8]), "+v"(rsw[9]), "+v"(rsw[10]), "+v"(rsw[11]), "+v"(rsw[12]), "+v"(rsw[13]),
                       "+v"(rsw[14]), "+v"(rsw[15]));
        float rs[16];
#pragma unroll
        for (int r = 0; r < 16; r++) rs[r] = __builtin_bit_cast(float, BF ? rsw[r] << 16 : rsw[r]);
        const int py = it.phase >> 1, px = it.phase & 1;
        const bool second = fused && c.out2 != nullptr && it.n0 >= c.split_n;  // two layers sharing one input run as one GEMM: columns [0, split_n) -> out, the rest -> out2 (split_n is a multiple of the tile width)
        gfloat* outp = (gfloat*)(!fused ? c.ws + (long long)it.ks * c.slab_pix * c.Npad : (second ? c.out2 : c.out));
        const int ncol0 = second ? c.split_n : 0;
        const int ldo = !fused ? c.Npad : (second ? c.ldc2 : c.ldc);
        const int nlim = fused ? c.Nvalid : c.Npad;
        const bool of32 = !BF || !fused || c.out_f32;  // split-K slabs and the final maps stay fp32
        const bool relu = fused && it.n0 < c.relu_cols;  // uniform per tile: relu_cols is 0, >= N, or a multiple of the tile width
        if (!fused) bias = 0.f, sc = 1.f, sh = 0.f;       // slabs are raw partial sums
        if (direct) {
            // rows past M of the last tile go to the slack of the tensor / slab: no per-row test
            const unsigned off0 = (unsigned)(mb * ldo + (n - ncol0));
            auto rows = [&](auto RELU, auto OUTF32) __attribute__((always_inline)) {
                using OT = typename std::conditional<decltype(OUTF32)::value, gfloat, gbf16>::type;
                OT* op = (OT*)outp;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float o = acc[r] + bias;
                    if (c.scale) o = o * sc + sh;
                    o = o + rs[r];
                    if constexpr (decltype(RELU)::value) o = __builtin_fmaxf(o, 0.f);
                    if constexpr (decltype(OUTF32)::value) put_f32((gfloat*)op + off0, o);
                    else put_bf16((gbf16*)op + off0, o);
                    op += ((r & 3) == 3 ? 5 : 1) * ldo;
                }
            };
            if (n < nlim) {
                if (of32) {
                    if (relu) rows(std::true_type{}, std::true_type{});
                    else rows(std::false_type{}, std::true_type{});
                } else {
                    if (relu) rows(std::true_type{}, std::false_type{});
                    else rows(std::false_type{}, std::false_type{});
                }
            }
        } else {
            // transposed conv: the 4 phases scatter rows to (2i+py, 2j+px)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                const int t = fdiv(m, h.mg_wo, h.Wo), ox = m - t * h.Wo;
                const int s = fdiv(t, h.mg_ho, h.Ho), oy = t - s * h.Ho;
                const int op = (s * c.OH + oy * c.os + py) * c.OW + ox * c.os + px;
                float o = acc[r] + bias;
                if (c.scale) o = o * sc + sh;
                o = o + rs[r];
                if (relu) o = __builtin_fmaxf(o, 0.f);
                if (m < h.M && n < nlim) {
                    const unsigned off = (unsigned)(op * ldo + (n - ncol0));  // tensors are far below 2^32 elements
                    if (of32) put_f32(outp + off, o);
                    else put_bf16((gbf16*)outp + off, o);
                }
            }
        }
        if (pstamp && j == 0) prof[21] = __builtin_amdgcn_s_memrealtime();  // first item: stores issued
    }
    if (pstamp) prof[12] = __builtin_amdgcn_s_memrealtime(), prof[19] = cbw;
    if (P1 && threadIdx.x == 0) {
        // PROF = 1 stamps the end when this wave's last store has been ISSUED: waiting for the stores first (PROF = 2 does, for
        // its "stores drained" stamp) puts a store round trip in front of the stamp's own store in every workgroup
        if constexpr (P2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (pstamp) prof[13] = __builtin_amdgcn_s_memrealtime();
        if (blockIdx.x == 0) {  // workgroup 0's own span in both clocks (its end, not the launch's: one workgroup, one counter)
            prof[25] = (unsigned long long)__builtin_amdgcn_s_memtime();
            prof[26] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
        }
        // A grid above PROF_WGS (the one-item launches of more than 512 tiles, round 5) shares slots: workgroup id and id + 512 write the
        // same one from different CUs, and plain stores of different CUs are not ordered (advisor, round 5: the host's maximum could be
        // under-reported).  Those grids take the agent-scope atomic maximum -- the slots are zeroed per frame -- and grids that fit keep
        // the plain store (an atomic in every workgroup of every launch put ~1.4 us behind each: round 2).
        if (gridDim.x > PROF_WGS)
            __hip_atomic_fetch_max(&a.prof_end[blockIdx.x & (PROF_WGS - 1)], (unsigned long long)__builtin_amdgcn_s_memrealtime(),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            a.prof_end[blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
    }
}

// split-K second pass: slabs summed in slice order (deterministic), then the same epilogue.  One thread = 4 consecutive channels of one pixel:
// the slab loads, the bias / BN vectors and the shortcut are requested together, the result leaves as ONE 16-byte (fp32) or 8-byte (bf16)
// write-through store where all four channels are valid (round 4: the scalar per-channel form took 7.7 us for 19 MB).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ReduceArgs a)
{
    const int n4 = a.Npad >> 2;
    const long long total = a.npix * n4;
    // vector loads / stores of the shortcut and the output need 4-element pixel strides AND 16-byte (bf16: 8-byte) aligned bases -- checked once
    // here for both (advisor, round 4: the alignment test used to sit at the store only)
    const int esz = a.bf16 ? 2 : 4, oesz = (a.bf16 && !a.out_f32) ? 2 : 4;
    const bool vec_ok = ((a.ldc | a.ldr) & 3) == 0 && (((uintptr_t)a.out) & (4 * oesz - 1)) == 0 && (!a.resid || (((uintptr_t)a.resid) & (4 * esz - 1)) == 0);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pix = idx / n4;
        const int n = (int)(idx - pix * n4) * 4;
        f32x4 part[8];
#pragma unroll
        for (int k = 0; k < 8; k++)  // all slab loads in flight at once
            if (k < a.ksplit) part[k] = *(const f32x4*)(a.ws + ((long long)k * a.slab_pix + pix) * a.Npad + n);
        const bool whole = vec_ok && n + 3 < a.Nvalid;  // (bias / scale / shift are padded to Npad)
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, scv = {1.f, 1.f, 1.f, 1.f}, shv = {0.f, 0.f, 0.f, 0.f}, rv = {0.f, 0.f, 0.f, 0.f};
        if (whole) {
            bv = *(const f32x4*)(a.bias + n);
            if (a.scale) scv = *(const f32x4*)(a.scale + n), shv = *(const f32x4*)(a.shift + n);
            if (a.resid) {
                if (a.bf16) rv = __builtin_convertvector(*(const bf16x4*)((const __bf16*)a.resid + pix * a.ldr + n), f32x4);
                else rv = *(const f32x4*)(a.resid + pix * a.ldr + n);
            }
        }
        f32x4 s = part[0];
#pragma unroll
        for (int k = 1; k < 8; k++)  // summed in slice order: deterministic
            if (k < a.ksplit) s += part[k];
        if (whole) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float t = s[e] + bv[e];
                if (a.scale) t = t * scv[e] + shv[e];
                if (a.resid) t = t + rv[e];
                if (n + e < a.relu_cols) t = t > 0.f ? t : 0.f;
                v[e] = t;
            }
            if (a.bf16 && !a.out_f32) store_wt((bf16x4*)((__bf16*)a.out + pix * a.ldc + n), __builtin_convertvector(v, bf16x4));
            else store_wt((f32x4*)(a.out + pix * a.ldc + n), v);
            continue;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int c = n + e;
            if (c >= a.Nvalid) continue;
            float v = s[e] + a.bias[c];
            if (a.scale) v = v * a.scale[c] + a.shift[c];
            if (a.resid) v = v + (a.bf16 ? (float)((const __bf16*)a.resid)[pix * a.ldr + c] : a.resid[pix * a.ldr + c]);
            if (c < a.relu_cols) v = v > 0.f ? v : 0.f;
            if (a.bf16 && !a.out_f32) ((__bf16*)a.out)[pix * a.ldc + c] = (__bf16)v;
            else a.out[pix * a.ldc + c] = v;
        }
    }
}

template <int BM, int BN, int KG, int NS>
constexpr size_t stream_lds() { return (size_t)NS * (BM + BN) * 32 * KG * 4 + (KG > 1 ? (size_t)(KG - 1) * (4 / KG) * (BN == 96 ? 3 : 1) * 4096 + 64 : 0); }
// split-product form: a stage holds BM x 128 B of activations and BN x 192 B of weight planes per K group
template <int BM, int BN, int KG, int NS>
constexpr size_t x3_stream_lds() { return (size_t)NS * (BM * 32 + BN * 48) * KG * 4 + (KG > 1 ? (size_t)(KG - 1) * (4 / KG) * 4096 + 64 : 0); }
constexpr int X3_NS = 4;  // 64x64: 4 stages of 20 KiB (the same 80 KB as 5 x 16 KiB: two workgroups per CU)
// ring depths of the shapes that run ONE workgroup per CU
static int g_cus = 256;  // compute units of the device the handles run on (conv_setup); MI355X: 256
#ifndef NS_6432
#define NS_6432 5
#endif
#ifndef NS_32128
#define NS_32128 5
#endif
constexpr int X3_NS_6432 = 5;  // the split-product form of 64x32x2: 5 stages of 28 KiB
static_assert(stream_lds<64, 32, 2, NS_6432>() <= 160 * 1024 && stream_lds<32, 128, 1, NS_32128>() <= 160 * 1024 && x3_stream_lds<64, 32, 2, X3_NS_6432>() <= 160 * 1024, "one workgroup's ring fits a CU's LDS");
constexpr size_t x3_lds() { return x3_stream_lds<64, 64, 1, X3_NS>(); }

template <int BM, int BN, int KG, int NS>
static hipError_t launch_stream(ConvArgs a, hipStream_t st)
{
    if (a.cpt % KG != 0 || a.Npad % BN != 0) return hipErrorInvalidValue;
    a.cpt /= KG;  // the kernel counts K in steps of KG chunks
    a.tiles_m = (a.M + BM - 1) / BM, a.tiles_n = a.Npad / BN;
    auto magic = [](int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
    a.mg_wo = magic(a.Wo), a.mg_ho = magic(a.Ho), a.mg_tn = magic(a.tiles_n), a.mg_tm = magic(a.tiles_m);
    a.mg_ks = magic(a.ksplit), a.mg_cpt = magic(a.cpt);
    a.items = a.tiles_m * a.tiles_n * a.nphase * a.ksplit;
    // two workgroups per CU at most (one for the K-group shapes: their ring fills the LDS): more tiles than that are
    // walked by the same workgroups (VNECT_MAXWG: tuning)
    static const int maxwg_env = getenv("VNECT_MAXWG") ? atoi(getenv("VNECT_MAXWG")) : 0;
    const int cus = g_cus;  // compute units of the device (conv_setup; 256 on MI355X)
    const int maxwg = maxwg_env > 0 ? maxwg_env : (KG == 1 && BN <= 64 ? 2 * cus : cus);
    // One workgroup per item also ABOVE two per CU (round 5): the hardware starts the later workgroups as the earlier ones retire, each with the
    // lean one-item kernel's cold start (~1 us) instead of the streaming kernel's (~2 us) in front of every workgroup's few tiles.  Which launches
    // gain is a measurement (tools/ab_multi.sh, two calls, profiles/r05_ab_big_grids.txt), three launches of a 3-scale frame are concerned:
    //   fp32: res3a's projection pair (1 000 tiles of 8 chunks) +0.2 %; res5a's pair (600 tiles of 32 chunks) -1.9 %: short K loops only;
    //   bf16: both pairs as one-item launches +1.0 % (res3a's alone -0.5 %, res5a's alone +0.3 %): every launch up to 1 024 tiles.
    static const bool no_one = getenv("VNECT_NO_ONE") != nullptr, no_big = getenv("VNECT_NO_BIG_GRID") != nullptr;  // A/B runs
    const bool plain64 = KG == 1 && BN <= 64 && maxwg_env <= 0 && !no_big && !a.x3 && !a.pixmode;
    const bool big_grid = plain64 && (a.bf16 ? a.items <= 4 * cus : (a.ntaps * a.cpt <= 8 && a.items <= 8 * cus));
    const bool one = a.ksplit == 1 && !no_one && (a.items <= maxwg || big_grid);
    dim3 grid(one ? a.items : (a.items < maxwg ? a.items : maxwg));
    const size_t lds = stream_lds<BM, BN, KG, NS>();
    // profiling twin: start / end stamps only, or (VNECT_PROF_DETAIL=1, tools/phase_table.py) the per-phase stamps too
    static const bool detail = getenv("VNECT_PROF_DETAIL") && atoi(getenv("VNECT_PROF_DETAIL")) != 0;
    const int prof = a.prof ? (detail ? 2 : 1) : 0;
    // one item per workgroup and no K slabs (`one`, above): the kernel without the streaming machinery (conv_stream_kernel, ONE)
#define LAUNCH_STREAM(BF, PR)                                                                                                        \
    do {                                                                                                                             \
        if (one) hipLaunchKernelGGL((conv_stream_kernel<BM, BN, KG, NS, BF, PR, 0, false, false, true>), grid, dim3(512), lds, st, a); \
        else hipLaunchKernelGGL((conv_stream_kernel<BM, BN, KG, NS, BF, PR>), grid, dim3(512), lds, st, a);                          \
    } while (0)
    if constexpr (BM == 64 && BN == 64 && KG == 1) {
        if (a.x3) {  // split-product form (plain, with the tail GEMM or with the bone features behind it); start / end stamps at most
            if (a.bf16 || a.pixmode || a.K % 32) return hipErrorInvalidValue;
            if (a.bone && (a.items > maxwg || a.ksplit != 1 || a.Npad != 192 || a.ldc < 212 || a.tail_n > 0)) return hipErrorInvalidValue;
            if (a.tail_n > 0 && (a.items > maxwg || a.ksplit != 1 || a.Npad != 64 || a.os != 1 || a.tail_n != 256 || !a.tail_w || !a.tail_bias))
                return hipErrorInvalidValue;
            // (Measured, A/B in one call each: a 7-stage ring for launches with one workgroup per CU 1 162 vs 1 174 frames/s without; the
            // hi plane's bytes only -- 12 KiB per chunk instead of 20 -- +1.6 %; no split arithmetic +13 %; 3 MFMAs instead of 6 +9 %:
            // the loop is bound by the VALU + MFMA issue of the consumer waves, not by bytes into the LDS.  DESIGN 4.1d.)
            const int fu = a.bone ? 2 : (a.tail_n > 0 ? 1 : 0);
#define LAUNCH_X3(NSX, PR, FU) hipLaunchKernelGGL((conv_stream_kernel<64, 64, 1, NSX, false, PR, FU, false, true>), grid, dim3(512), (x3_stream_lds<64, 64, 1, NSX>()), st, a)
#define LAUNCH_X3_FU(NSX, PR)                 \
    do {                                      \
        if (fu == 0) LAUNCH_X3(NSX, PR, 0);   \
        else if (fu == 1) LAUNCH_X3(NSX, PR, 1); \
        else LAUNCH_X3(NSX, PR, 2);           \
    } while (0)
            if (prof == 0) LAUNCH_X3_FU(X3_NS, 0);
            else LAUNCH_X3_FU(X3_NS, 1);
#undef LAUNCH_X3_FU
#undef LAUNCH_X3
            return hipGetLastError();
        }
        if (a.tail_n > 0) {  // tail GEMM variant: one tile per workgroup, start / end stamps at most
            if (a.items > maxwg || a.ksplit != 1 || a.nphase != 1 || a.Npad != 64 || a.os != 1 || a.tail_n != 256 || !a.tail_w || !a.tail_bias)
                return hipErrorInvalidValue;
            // the 64-wide chain: bf16, behind a tail with shortcut + ReLU and bf16 output only
            if (a.chain_n != 0 && (!a.bf16 || a.chain_n != 64 || !a.chain_w || !a.chain_bias || !a.chain_out || a.chain_ld < 64 || !a.resid ||
                                   a.relu_cols < 256 || a.out_f32 || a.Nvalid != 256 ||
                                   (a.ldc & 7) != 0))  // the chain stores the block output from the LDS tile, which tail_gemm fills only in its
                                                       // staged form (tail_staged: whole 16-byte units per row, ldc % 8 == 0; advisor, round 5)
                return hipErrorInvalidValue;
#define LAUNCH_TAIL(BF, PR) hipLaunchKernelGGL((conv_stream_kernel<64, 64, 1, NS, BF, PR, 1>), grid, dim3(512), lds, st, a)
            if (a.bf16 && a.chain_n) {
                if (prof == 0) hipLaunchKernelGGL((conv_stream_kernel<64, 64, 1, NS, true, 0, 3>), grid, dim3(512), lds, st, a);
                else hipLaunchKernelGGL((conv_stream_kernel<64, 64, 1, NS, true, 1, 3>), grid, dim3(512), lds, st, a);
            } else if (a.bf16) {
                if (prof == 0) LAUNCH_TAIL(true, 0);
                else LAUNCH_TAIL(true, 1);
            } else {
                if (prof == 0) LAUNCH_TAIL(false, 0);
                else LAUNCH_TAIL(false, 1);
            }
#undef LAUNCH_TAIL
            return hipGetLastError();
        }
    }
    if constexpr (BM == 64 && BN == 32 && KG == 2) {
        if (a.x3) {  // split-product form of the in-workgroup K-group shape (one workgroup per CU; 5 stages of 28 KiB)
            if (a.bf16 || a.pixmode || a.K % 32 || a.tail_n > 0 || a.bone) return hipErrorInvalidValue;
            const size_t l3 = x3_stream_lds<64, 32, 2, X3_NS_6432>();
            if (prof == 0) hipLaunchKernelGGL((conv_stream_kernel<64, 32, 2, X3_NS_6432, false, 0, 0, false, true>), grid, dim3(512), l3, st, a);
            else hipLaunchKernelGGL((conv_stream_kernel<64, 32, 2, X3_NS_6432, false, 1, 0, false, true>), grid, dim3(512), l3, st, a);
            return hipGetLastError();
        }
    }
    if constexpr (BN == 96) {
        // three accumulators per wave (fp32): no shortcut, one output tensor, no K slabs, no tail; plain or with the bone features
        if (a.bf16 || a.x3 || a.tail_n > 0 || a.resid || a.out2 || a.ksplit != 1 || a.pixmode || (a.relu_cols & 31)) return hipErrorInvalidValue;
        if (a.bone && (a.items > maxwg || a.Npad != 192 || a.ldc < 212)) return hipErrorInvalidValue;
#define LAUNCH_96(PR, FU) hipLaunchKernelGGL((conv_stream_kernel<BM, BN, KG, NS, false, PR, FU>), grid, dim3(512), lds, st, a)
#define LAUNCH_96_ONE(PR) hipLaunchKernelGGL((conv_stream_kernel<BM, BN, KG, NS, false, PR, 0, false, false, true>), grid, dim3(512), lds, st, a)
        if (a.bone) {
            if (prof == 0) LAUNCH_96(0, 2);
            else LAUNCH_96(1, 2);
        } else if (one) {
            if (prof == 0) LAUNCH_96_ONE(0);
            else LAUNCH_96_ONE(1);
        } else {
            if (prof == 0) LAUNCH_96(0, 0);
            else LAUNCH_96(1, 0);
        }
#undef LAUNCH_96_ONE
#undef LAUNCH_96
        return hipGetLastError();
    }
    if constexpr (BM == 32 && BN == 128 && KG == 1) {
        if (a.tail_n > 0) {  // the wide tail (tail_wide): one 32x128 tile per workgroup, K = 128, at most 16 column blocks
            if (a.x3 || a.bone || a.items > maxwg || a.ksplit != 1 || a.nphase != 1 || a.Npad != 128 || a.os != 1 || a.tail_n > 512 || !a.tail_w || !a.tail_bias ||
                (a.chain_n != 0 && (a.chain_n != 128 || a.tail_n != 512 || !a.chain_w || !a.chain_bias || !a.chain_out || a.chain_ld < 128)))
                return hipErrorInvalidValue;
#define LAUNCH_WTAIL(BF, PR)                                                                                                              \
    do {                                                                                                                                  \
        if (a.chain_n) hipLaunchKernelGGL((conv_stream_kernel<32, 128, 1, NS, BF, PR, 3>), grid, dim3(512), lds, st, a);                     \
        else hipLaunchKernelGGL((conv_stream_kernel<32, 128, 1, NS, BF, PR, 1>), grid, dim3(512), lds, st, a);                               \
    } while (0)
            if (a.bf16) {
                if (prof == 0) LAUNCH_WTAIL(true, 0);
                else LAUNCH_WTAIL(true, 1);
            } else {
                if (prof == 0) LAUNCH_WTAIL(false, 0);
                else LAUNCH_WTAIL(false, 1);
            }
#undef LAUNCH_WTAIL
            return hipGetLastError();
        }
    }
    if (a.tail_n > 0 || a.x3) return hipErrorInvalidValue;
    if constexpr (BM == 64 && BN == 64 && KG == 1) {
        if (a.bone) {  // bone-length features inside the transposed conv's launch: one tile per workgroup again
            if (a.items > maxwg || a.ksplit != 1 || a.Npad != 192 || a.ldc < 212) return hipErrorInvalidValue;
#define LAUNCH_BONE(BF, PR) hipLaunchKernelGGL((conv_stream_kernel<64, 64, 1, NS, BF, PR, 2>), grid, dim3(512), lds, st, a)
            if (a.bf16) {
                if (prof == 0) LAUNCH_BONE(true, 0);
                else LAUNCH_BONE(true, 1);
            } else {
                if (prof == 0) LAUNCH_BONE(false, 0);
                else LAUNCH_BONE(false, 1);
            }
#undef LAUNCH_BONE
            return hipGetLastError();
        }
    }
    if (a.bone) return hipErrorInvalidValue;
    if constexpr (BM == 64 && BN == 64 && KG == 1) {
        // conv1, fp32: the span form (VNECT_NO_SPAN=1: the gathered-window form, for A/B runs)
        const bool no_span = getenv("VNECT_NO_SPAN") != nullptr;  // read per launch: a test flips it inside one process
        if (a.pixmode && !a.bf16 && !no_span && a.ksplit == 1 && a.cpt == 1 && a.Wo >= 64 && a.stride == 2) {
#define LAUNCH_SPAN(PR) hipLaunchKernelGGL((conv_stream_kernel<64, 64, 1, NS, false, PR, 0, true>), grid, dim3(512), lds, st, a)
            if (prof == 0) LAUNCH_SPAN(0);
            else LAUNCH_SPAN(1);
#undef LAUNCH_SPAN
            return hipGetLastError();
        }
    }
    if constexpr (BN != 96) {
        if (a.bf16) {
            if (prof == 0) LAUNCH_STREAM(true, 0);
            else if (prof == 1) LAUNCH_STREAM(true, 1);
            else LAUNCH_STREAM(true, 2);
        } else {
            if (prof == 0) LAUNCH_STREAM(false, 0);
            else if (prof == 1) LAUNCH_STREAM(false, 1);
            else LAUNCH_STREAM(false, 2);
        }
    }
#undef LAUNCH_STREAM
    return hipGetLastError();
}

// The three-accumulator shape is an optimisation with a working 64x64 fallback: if a toolchain ever allocates more than 256 registers or
// spills for it, the library stays usable and the launch plan keeps the transposed conv on 64x64 tiles (rt_plan.cpp: choose_tile).
static bool g_deconv96 = false;
bool conv_deconv96_available() { return g_deconv96; }

template <int BM, int BN, int KG, int NS>
static hipError_t setup_stream_rest();
template <int BM, int BN, int KG, int NS>
static hipError_t setup_stream()
{
    hipFuncAttributes fa;
    if constexpr (BN == 96) {
        for (const void* f : {(const void*)conv_stream_kernel<BM, BN, KG, NS, false, 0, 0>, (const void*)conv_stream_kernel<BM, BN, KG, NS, false, 1, 0>,
                              (const void*)conv_stream_kernel<BM, BN, KG, NS, false, 0, 2>, (const void*)conv_stream_kernel<BM, BN, KG, NS, false, 1, 2>,
                              (const void*)conv_stream_kernel<BM, BN, KG, NS, false, 0, 0, false, false, true>,
                              (const void*)conv_stream_kernel<BM, BN, KG, NS, false, 1, 0, false, false, true>}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stream_lds<BM, BN, KG, NS>());
            if (e != hipSuccess) return e;
            e = hipFuncGetAttributes(&fa, f);
            if (e != hipSuccess) return e;
            if (fa.numRegs > 256 || fa.localSizeBytes != 0) return hipErrorLaunchOutOfResources;
        }
        return hipSuccess;
    } else {
        return setup_stream_rest<BM, BN, KG, NS>();
    }
}
template <int BM, int BN, int KG, int NS>
static hipError_t setup_stream_rest()
{
    hipFuncAttributes fa;
    std::vector<const void*> fns = {(const void*)conv_stream_kernel<BM, BN, KG, NS, false, 0>, (const void*)conv_stream_kernel<BM, BN, KG, NS, true, 0>,
                                    (const void*)conv_stream_kernel<BM, BN, KG, NS, false, 1>, (const void*)conv_stream_kernel<BM, BN, KG, NS, true, 1>,
                                    (const void*)conv_stream_kernel<BM, BN, KG, NS, false, 2>, (const void*)conv_stream_kernel<BM, BN, KG, NS, true, 2>};
    const void* twins_one[2] = {(const void*)conv_stream_kernel<BM, BN, KG, NS, false, 2, 0, false, false, true>, (const void*)conv_stream_kernel<BM, BN, KG, NS, true, 2, 0, false, false, true>};
    for (const void* f : {(const void*)conv_stream_kernel<BM, BN, KG, NS, false, 0, 0, false, false, true>, (const void*)conv_stream_kernel<BM, BN, KG, NS, true, 0, 0, false, false, true>,
                          (const void*)conv_stream_kernel<BM, BN, KG, NS, false, 1, 0, false, false, true>, (const void*)conv_stream_kernel<BM, BN, KG, NS, true, 1, 0, false, false, true>,
                          twins_one[0], twins_one[1]})
        fns.push_back(f);
    if constexpr (BM == 64 && BN == 64 && KG == 1) {
        fns.push_back((const void*)conv_stream_kernel<64, 64, 1, NS, false, 0, 1>), fns.push_back((const void*)conv_stream_kernel<64, 64, 1, NS, true, 0, 1>);
        fns.push_back((const void*)conv_stream_kernel<64, 64, 1, NS, false, 1, 1>), fns.push_back((const void*)conv_stream_kernel<64, 64, 1, NS, true, 1, 1>);
        fns.push_back((const void*)conv_stream_kernel<64, 64, 1, NS, false, 0, 0, true>), fns.push_back((const void*)conv_stream_kernel<64, 64, 1, NS, false, 1, 0, true>);
        fns.push_back((const void*)conv_stream_kernel<64, 64, 1, NS, true, 0, 3>), fns.push_back((const void*)conv_stream_kernel<64, 64, 1, NS, true, 1, 3>);
        fns.push_back((const void*)conv_stream_kernel<64, 64, 1, NS, false, 0, 2>), fns.push_back((const void*)conv_stream_kernel<64, 64, 1, NS, true, 0, 2>);
        fns.push_back((const void*)conv_stream_kernel<64, 64, 1, NS, false, 1, 2>), fns.push_back((const void*)conv_stream_kernel<64, 64, 1, NS, true, 1, 2>);
        static_assert(x3_lds() <= stream_lds<64, 64, 1, 5>(), "the split-product ring fits the same LDS allowance");
    }
    if constexpr (BM == 64 && BN == 64 && KG == 1) {  // the split-product instantiations have LDS sizes of their own
        static_assert(x3_lds() <= stream_lds<64, 64, 1, 5>(), "the split-product ring fits the two-workgroups-per-CU LDS allowance");
        const void* x3s[] = {(const void*)conv_stream_kernel<64, 64, 1, X3_NS, false, 0, 0, false, true>, (const void*)conv_stream_kernel<64, 64, 1, X3_NS, false, 1, 0, false, true>,
                             (const void*)conv_stream_kernel<64, 64, 1, X3_NS, false, 0, 1, false, true>, (const void*)conv_stream_kernel<64, 64, 1, X3_NS, false, 1, 1, false, true>,
                             (const void*)conv_stream_kernel<64, 64, 1, X3_NS, false, 0, 2, false, true>, (const void*)conv_stream_kernel<64, 64, 1, X3_NS, false, 1, 2, false, true>};
        for (int i = 0; i < 6; i++) {
            hipError_t e = hipFuncSetAttribute(x3s[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3_stream_lds<64, 64, 1, X3_NS>());
            if (e != hipSuccess) return e;
            e = hipFuncGetAttributes(&fa, x3s[i]);
            if (e != hipSuccess) return e;
            if (fa.numRegs > 128 || fa.localSizeBytes != 0) return hipErrorLaunchOutOfResources;
        }
    }
    if constexpr (BM == 32 && BN == 128 && KG == 1) {
        fns.push_back((const void*)conv_stream_kernel<32, 128, 1, NS, false, 0, 1>), fns.push_back((const void*)conv_stream_kernel<32, 128, 1, NS, true, 0, 1>);
        fns.push_back((const void*)conv_stream_kernel<32, 128, 1, NS, false, 1, 1>), fns.push_back((const void*)conv_stream_kernel<32, 128, 1, NS, true, 1, 1>);
        fns.push_back((const void*)conv_stream_kernel<32, 128, 1, NS, false, 0, 3>), fns.push_back((const void*)conv_stream_kernel<32, 128, 1, NS, true, 0, 3>);
        fns.push_back((const void*)conv_stream_kernel<32, 128, 1, NS, false, 1, 3>), fns.push_back((const void*)conv_stream_kernel<32, 128, 1, NS, true, 1, 3>);
    }
    if constexpr (BM == 64 && BN == 32 && KG == 2) {
        for (const void* f : {(const void*)conv_stream_kernel<64, 32, 2, X3_NS_6432, false, 0, 0, false, true>, (const void*)conv_stream_kernel<64, 32, 2, X3_NS_6432, false, 1, 0, false, true>}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3_stream_lds<64, 32, 2, X3_NS_6432>());
            if (e != hipSuccess) return e;
            e = hipFuncGetAttributes(&fa, f);
            if (e != hipSuccess) return e;
            if (fa.numRegs > 256 || fa.localSizeBytes != 0) return hipErrorLaunchOutOfResources;
        }
    }
    const void* span_twin = nullptr;  // conv1's span form with start / end stamps (VNECT_NO_STEM=1 under the profiling twin): may spill 8 bytes
    if constexpr (BM == 64 && BN == 64 && KG == 1) span_twin = (const void*)conv_stream_kernel<64, 64, 1, NS, false, 1, 0, true>;
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stream_lds<BM, BN, KG, NS>());
        if (e != hipSuccess) return e;
        e = hipFuncGetAttributes(&fa, f);
        if (e != hipSuccess) return e;
        // the launch plan assumes two workgroups per CU (one for the K-group shapes): refuse a build that needs more registers
        // or scratch (the per-phase tuning twins, PROF = 2, may spill a few bytes)
        const bool twin = f == (const void*)conv_stream_kernel<BM, BN, KG, NS, false, 2> || f == (const void*)conv_stream_kernel<BM, BN, KG, NS, true, 2> ||
                          f == twins_one[0] || f == twins_one[1] || f == span_twin;
        if (fa.numRegs > (KG == 1 && BN <= 64 ? 128 : 256) || (fa.localSizeBytes != 0 && !twin && !F32_NOSTORE)) return hipErrorLaunchOutOfResources;  // (the probe build may spill)
    }
    return hipSuccess;
}

// Ring depths: 5 x 16 KiB leaves room for two 64x64 workgroups per CU (measured faster than one deeper ring on every
// layer); the K-group shapes run one workgroup per CU.
int conv_cu_count() { return g_cus; }

hipError_t conv_setup()
{
    hipError_t e;
    {   // the launch plan's "rounds over the chip" are counted in compute units of the CURRENT device (advisor, round 5: no literal 256)
        int dev = 0, n = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        if (n >= 8) g_cus = n;
    }
    if ((e = setup_stream<64, 64, 1, 5>()) != hipSuccess) return e;
    if ((e = setup_stream<64, 32, 2, NS_6432>()) != hipSuccess) return e;
    if ((e = setup_stream<32, 32, 4, 4>()) != hipSuccess) return e;
    if ((e = setup_stream<32, 128, 1, NS_32128>()) != hipSuccess) return e;
    e = setup_stream<64, 96, 2, 3>();
    g_deconv96 = e == hipSuccess;
    if (e != hipSuccess && e != hipErrorLaunchOutOfResources) return e;  // out of registers: deconv96 unavailable, not an error
    return hipSuccess;
}

hipError_t launch_conv(const ConvArgs& a, int BM, int BN, int KG, hipStream_t st)
{
    const int epr = a.bf16 ? 64 : 32;
    if (a.Npad % BN != 0 || a.K != a.ntaps * a.cpt * epr || a.nphase * a.ntaps > MAX_TAPS ||
        a.ksplit < 1 || (a.ksplit > 1 && !a.ws) || (a.Cs & 3) || (a.bf16 && a.out2 && a.split_n % 64))
        return hipErrorInvalidValue;
    // range of the multiply-high divisions in the kernel (x / d exact while x * d < 2^32)
    if ((long long)a.M * (a.Wo > a.Ho ? a.Wo : a.Ho) >= (1ll << 32) || a.M >= (1 << 24)) return hipErrorInvalidValue;
    if (KG == 2 && BM == 64 && BN == 32) return launch_stream<64, 32, 2, NS_6432>(a, st);
    if (KG == 4 && BM == 32 && BN == 32) return launch_stream<32, 32, 4, 4>(a, st);
    if (KG == 1 && BM == 64 && BN == 64) return launch_stream<64, 64, 1, 5>(a, st);
    if (KG == 1 && BM == 32 && BN == 128) return launch_stream<32, 128, 1, NS_32128>(a, st);
    if (KG == 2 && BM == 64 && BN == 96) return g_deconv96 ? launch_stream<64, 96, 2, 3>(a, st) : hipErrorInvalidValue;  // 3 stages of 40 KiB + 24 KiB of partial sums: one workgroup per CU
    return hipErrorInvalidValue;
}

hipError_t launch_reduce(const ReduceArgs& a, hipStream_t st)
{
    long long total = a.npix * (a.Npad >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---- layout helpers (T = float or __bf16 activations) ------------------------------------------------
template <typename T>
__global__ void pad3to4_kernel(const float* __restrict__ in3, T* __restrict__ out4, long long npix)
{
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    out4[p * 4 + 0] = (T)in3[p * 3], out4[p * 4 + 1] = (T)in3[p * 3 + 1], out4[p * 4 + 2] = (T)in3[p * 3 + 2], out4[p * 4 + 3] = (T)0.f;
}
template <typename T>
__global__ void strip4to3_kernel(const T* __restrict__ in4, float* __restrict__ out3, long long npix)
{
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    out3[p * 3] = (float)in4[p * 4], out3[p * 3 + 1] = (float)in4[p * 4 + 1], out3[p * 3 + 2] = (float)in4[p * 4 + 2];
}
hipError_t launch_pad3to4(const float* in3, void* out4, long long npix, int bf16, hipStream_t st)
{
    dim3 g((unsigned)((npix + 255) / 256));
    if (bf16) hipLaunchKernelGGL(pad3to4_kernel<__bf16>, g, dim3(256), 0, st, in3, (__bf16*)out4, npix);
    else hipLaunchKernelGGL(pad3to4_kernel<float>, g, dim3(256), 0, st, in3, (float*)out4, npix);
    return hipGetLastError();
}
hipError_t launch_strip4to3(const void* in4, float* out3, long long npix, int bf16, hipStream_t st)
{
    dim3 g((unsigned)((npix + 255) / 256));
    if (bf16) hipLaunchKernelGGL(strip4to3_kernel<__bf16>, g, dim3(256), 0, st, (const __bf16*)in4, out3, npix);
    else hipLaunchKernelGGL(strip4to3_kernel<float>, g, dim3(256), 0, st, (const float*)in4, out3, npix);
    return hipGetLastError();
}

// MaxPool 3x3 stride 2, TF SAME (pad 0 before / 1 after for 184 -> 92): padding never wins.
// One thread = 4 channels of one output pixel (coalesced over channels).
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ in, T* __restrict__ out, int S, int H, int W, int C, int Ho, int Wo)
{
    typedef T tx4 __attribute__((ext_vector_type(4)));
    const int c4 = C >> 2;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)S * Ho * Wo * c4;
    if (idx >= total) return;
    int c = (int)(idx % c4) * 4;
    long long p = idx / c4;
    int ox = (int)(p % Wo);
    long long t = p / Wo;
    int oy = (int)(t % Ho), s = (int)(t / Ho);
    const float ninf = -__builtin_inff();
    f32x4 m = {ninf, ninf, ninf, ninf};
    for (int ky = 0; ky < 3; ky++) {
        int iy = oy * 2 + ky;
        if (iy >= H) continue;
        for (int kx = 0; kx < 3; kx++) {
            int ix = ox * 2 + kx;
            if (ix >= W) continue;
            f32x4 v = __builtin_convertvector(*(const tx4*)(in + (((long long)s * H + iy) * W + ix) * C + c), f32x4);
#pragma unroll
            for (int e = 0; e < 4; e++) m[e] = v[e] > m[e] ? v[e] : m[e];
        }
    }
    store_wt((tx4*)(out + p * C + c), __builtin_convertvector(m, tx4));  // exact: the maximum is one of the inputs
}
hipError_t launch_maxpool(const void* in, void* out, int S, int H, int W, int C, int Ho, int Wo, int bf16, hipStream_t st)
{
    long long total = (long long)S * Ho * Wo * (C >> 2);
    dim3 g((unsigned)((total + 255) / 256));
    if (bf16) hipLaunchKernelGGL(maxpool_kernel<__bf16>, g, dim3(256), 0, st, (const __bf16*)in, (__bf16*)out, S, H, W, C, Ho, Wo);
    else hipLaunchKernelGGL(maxpool_kernel<float>, g, dim3(256), 0, st, (const float*)in, (float*)out, S, H, W, C, Ho, Wo);
    return hipGetLastError();
}

// Bone-length features (vnect_model.py:198-209): feat[p][191+j] = sqrt((dx^2 + dy^2) + dz^2) from the
// delta channels feat[p][128+j], [149+j], [170+j]; channels 212..ld-1 are zero padding for the next conv.
template <typename T>
__global__ void bone_kernel(T* feat, long long npix, int ld)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = ld - 191;
    if (idx >= npix * per) return;
    long long p = idx / per;
    int j = (int)(idx - p * per);
    T* f = feat + p * ld;
    float v = 0.f;
    if (j < 21) {
        float x = (float)f[128 + j], y = (float)f[149 + j], z = (float)f[170 + j];
        v = bone_len(x, y, z);
    }
    f[191 + j] = (T)v;
}
// What this translation unit was compiled with (vnect_build_info): every timing probe must be off in the shipped library -- most of
// them give WRONG results on purpose -- and the ring depths at their defaults.  `NS_*` are defined further up, next to their launchers.
#define VNECT_STR2(x) #x
#define VNECT_STR(x) VNECT_STR2(x)
const char* conv_build_probes()
{
    return "X3_DBG=" VNECT_STR(X3_DBG) " WT_DBG=" VNECT_STR(WT_DBG) " CH_DBG=" VNECT_STR(CH_DBG) " F32_NOSTORE=" VNECT_STR(F32_NOSTORE)
           " BF16_NOSTORE=" VNECT_STR(BF16_NOSTORE) " VNECT_AB=" VNECT_STR(VNECT_AB) " NS_6432=" VNECT_STR(NS_6432) " NS_32128=" VNECT_STR(NS_32128);
}
bool conv_probes_off()
{
    return X3_DBG == 0 && WT_DBG == 0 && CH_DBG == 0 && F32_NOSTORE == 0 && BF16_NOSTORE == 0 && VNECT_AB == 0 && NS_6432 == 5 && NS_32128 == 5;
}

hipError_t launch_bone(void* feat, long long npix, int ld, int bf16, hipStream_t st)
{
    long long total = npix * (ld - 191);
    dim3 g((unsigned)((total + 255) / 256));
    if (bf16) hipLaunchKernelGGL(bone_kernel<__bf16>, g, dim3(256), 0, st, (__bf16*)feat, npix, ld);
    else hipLaunchKernelGGL(bone_kernel<float>, g, dim3(256), 0, st, (float*)feat, npix, ld);
    return hipGetLastError();
}

}  // namespace vnect
