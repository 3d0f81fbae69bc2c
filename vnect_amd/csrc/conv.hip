// conv.hip -- fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32),
// plus the small layout / pooling / bone-length kernels of the VNect graph.
//
// Replaces the TF1 ops executed by sess.run at /root/reference/src/estimator.py:100-104 for the graph
// of src/vnect_model.py:25-217: Conv2D(+BiasAdd+Add+Relu), Conv2DBackpropInput (as 4 sub-pixel phases),
// FusedBatchNorm (folded into the epilogue), MaxPool, and the bone-length Mul/Add/Sqrt/Concat.
//
// Tiling: one 256-thread workgroup = 4 waves (2x2) computes a BM x BN output tile; each wave owns
// (BM/2)x(BN/2) as 32x32 MFMA accumulators.  K runs in 32-float chunks: a chunk is one filter tap and
// 32 consecutive input channels, i.e. one 128-byte run per NHWC input pixel, so global reads are
// whole cache lines.  A (gathered activation rows) and B (pre-packed weights, [N][K]) chunks are
// register-staged into a double-buffered LDS image with a 36-float row pitch: ds_write_b128 by 8-lane
// row groups and ds_read_b128 by MFMA lane groups are both bank-conflict free at that pitch.
#include "kernels.h"

namespace vnect {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LDT = 36;  // LDS row pitch in floats (32 + 4)

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvArgs a)
{
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32, AR = BM / 32, BR = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BM][LDT]
    float* Bs = smem + 2 * BM * LDT;  // [2][BN][LDT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int phase = blockIdx.z / a.ksplit, ks = blockIdx.z - phase * a.ksplit;
    const int nch = a.ntaps * a.cpt;
    const int c0 = (int)(((long long)nch * ks) / a.ksplit);
    const int c1 = (int)(((long long)nch * (ks + 1)) / a.ksplit);
    const int lrow = tid >> 3, col4 = tid & 7;

    // rows of the A tile this thread stages (same rows for every chunk)
    int a_iy[AR], a_ix[AR];
    long long a_base[AR];
#pragma unroll
    for (int i = 0; i < AR; i++) {
        int m = m0 + lrow + 32 * i;
        if (m < a.M) {
            int ox = m % a.Wo, t = m / a.Wo;
            int oy = t % a.Ho, s = t / a.Ho;
            a_iy[i] = oy * a.stride;
            a_ix[i] = ox * a.stride;
            a_base[i] = (long long)s * a.H * a.W;
        } else {
            a_iy[i] = -(1 << 20);  // fails the bounds test for every tap -> zeros
            a_ix[i] = 0;
            a_base[i] = 0;
        }
    }
    const float* wp = a.w + (long long)phase * a.w_phase_stride + (long long)(n0 + lrow) * a.K + col4 * 4;
    const int8_t* dyp = a.dy + phase * a.ntaps;
    const int8_t* dxp = a.dx + phase * a.ntaps;

    f32x4 ra[AR], rb[BR];
    auto load_chunk = [&](int c) {
        const int tap = c / a.cpt, ci0 = (c - tap * a.cpt) * 32;
        const int dy = dyp[tap], dx = dxp[tap];
#pragma unroll
        for (int i = 0; i < AR; i++) {
            const int iy = a_iy[i] + dy;
            const int ix = a_ix[i] + dx + (a.pixmode ? col4 : 0);
            const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const float* p = a.in + (a_base[i] + (long long)iy * a.W + ix) * a.Cs + (a.pixmode ? 0 : ci0 + col4 * 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *(const f32x4*)p;
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BR; i++) rb[i] = *(const f32x4*)(wp + (long long)i * 32 * a.K + (long long)c * 32);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AR; i++) *(f32x4*)&As[(buf * BM + lrow + 32 * i) * LDT + col4 * 4] = ra[i];
#pragma unroll
        for (int i = 0; i < BR; i++) *(f32x4*)&Bs[(buf * BN + lrow + 32 * i) * LDT + col4 * 4] = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    if (c0 < c1) {
        load_chunk(c0);
        store_chunk(0);
    }
    __syncthreads();

    // MFMA operand map (32x32x2 f32): lane l supplies A[row l&31][k = l>>5] and B[k = l>>5][col l&31].
    // Within a 32-deep chunk, lane half h reads the float4 at k = 8q + 4h .. +3 (q = 0..3) of its row for
    // both operands, so MFMA step (q, j) contracts k = 8q + j (h = 0) and k = 8q + 4 + j (h = 1).
    const int frag_off = (lane & 31) * LDT + 4 * (lane >> 5);
    for (int c = c0; c < c1; c++) {
        const int buf = (c - c0) & 1;
        if (c + 1 < c1) load_chunk(c + 1);  // global loads in flight during the MFMAs below
        const float* Ab = As + (buf * BM + wm * WM) * LDT + frag_off;
        const float* Bb = Bs + (buf * BN + wn * WN) * LDT + frag_off;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) af[i] = *(const f32x4*)(Ab + i * 32 * LDT + 8 * q);
#pragma unroll
            for (int j = 0; j < TN; j++) bf[j] = *(const f32x4*)(Bb + j * 32 * LDT + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        if (c + 1 < c1) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // Epilogue.  C/D map: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    const int py = phase >> 1, px = phase & 1;
    const bool direct = (a.os == 1);
    const long long npix = (long long)a.S * a.OH * a.OW;
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * WN + j * 32 + (lane & 31);
        const float bias = a.bias[n];
        const float sc = a.scale ? a.scale[n] : 1.f;
        const float sh = a.scale ? a.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= a.M) continue;
                long long opix = m;
                if (!direct) {
                    int ox = m % a.Wo, t = m / a.Wo;
                    int oy = t % a.Ho, s = t / a.Ho;
                    opix = ((long long)s * a.OH + oy * a.os + py) * a.OW + ox * a.os + px;
                }
                float v = acc[i][j][r];
                if (a.ksplit > 1) {
                    a.ws[((long long)ks * npix + opix) * a.Npad + n] = v;
                } else {
                    if (n >= a.Nvalid) continue;
                    v = v + bias;
                    if (a.scale) v = v * sc + sh;
                    if (a.resid) v = v + a.resid[opix * a.ldr + n];
                    if (n < a.relu_cols) v = v > 0.f ? v : 0.f;
                    a.out[opix * a.ldc + n] = v;
                }
            }
        }
    }
}

// split-K second pass: slabs summed in slice order (deterministic), then the same epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ReduceArgs a)
{
    const int n4 = a.Npad >> 2;
    const long long total = a.npix * n4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pix = idx / n4;
        const int n = (int)(idx - pix * n4) * 4;
        f32x4 s = *(const f32x4*)(a.ws + pix * a.Npad + n);
        for (int k = 1; k < a.ksplit; k++) {
            f32x4 t = *(const f32x4*)(a.ws + ((long long)k * a.npix + pix) * a.Npad + n);
            s += t;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int c = n + e;
            if (c >= a.Nvalid) continue;
            float v = s[e] + a.bias[c];
            if (a.scale) v = v * a.scale[c] + a.shift[c];
            if (a.resid) v = v + a.resid[pix * a.ldr + c];
            if (c < a.relu_cols) v = v > 0.f ? v : 0.f;
            a.out[pix * a.ldc + c] = v;
        }
    }
}

template <int BM, int BN>
static hipError_t launch_t(const ConvArgs& a, hipStream_t st)
{
    dim3 grid((a.M + BM - 1) / BM, a.Npad / BN, a.nphase * a.ksplit);
    size_t lds = (size_t)2 * (BM + BN) * LDT * sizeof(float);
    hipLaunchKernelGGL((conv_f32_kernel<BM, BN>), grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

hipError_t conv_setup()
{
    hipError_t e;
#define SET(BM, BN)                                                                                  \
    e = hipFuncSetAttribute((const void*)conv_f32_kernel<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            2 * (BM + BN) * LDT * (int)sizeof(float));                                \
    if (e != hipSuccess) return e;
    SET(64, 64) SET(128, 64) SET(64, 128) SET(128, 128)
#undef SET
    return hipSuccess;
}

hipError_t launch_conv(const ConvArgs& a, int BM, int BN, hipStream_t st)
{
    if (a.Npad % BN != 0 || a.K % 32 != 0 || a.K != a.ntaps * a.cpt * 32 || a.nphase * a.ntaps > MAX_TAPS ||
        a.ksplit < 1 || (a.ksplit > 1 && !a.ws) || (a.Cs & 3))
        return hipErrorInvalidValue;
    if (BM == 64 && BN == 64) return launch_t<64, 64>(a, st);
    if (BM == 128 && BN == 64) return launch_t<128, 64>(a, st);
    if (BM == 64 && BN == 128) return launch_t<64, 128>(a, st);
    if (BM == 128 && BN == 128) return launch_t<128, 128>(a, st);
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

// ---- layout helpers -------------------------------------------------------------------------
__global__ void pad3to4_kernel(const float* __restrict__ in3, float* __restrict__ out4, long long npix)
{
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    f32x4 v = {in3[p * 3], in3[p * 3 + 1], in3[p * 3 + 2], 0.f};
    *(f32x4*)(out4 + p * 4) = v;
}
__global__ void strip4to3_kernel(const float* __restrict__ in4, float* __restrict__ out3, long long npix)
{
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    f32x4 v = *(const f32x4*)(in4 + p * 4);
    out3[p * 3] = v[0], out3[p * 3 + 1] = v[1], out3[p * 3 + 2] = v[2];
}
hipError_t launch_pad3to4(const float* in3, float* out4, long long npix, hipStream_t st)
{
    hipLaunchKernelGGL(pad3to4_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, in3, out4, npix);
    return hipGetLastError();
}
hipError_t launch_strip4to3(const float* in4, float* out3, long long npix, hipStream_t st)
{
    hipLaunchKernelGGL(strip4to3_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, in4, out3, npix);
    return hipGetLastError();
}

// MaxPool 3x3 stride 2, TF SAME (pad 0 before / 1 after for 184 -> 92): padding never wins.
// One thread = 4 channels of one output pixel (float4 loads, coalesced over channels).
__global__ void maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, int S, int H, int W, int C,
                               int Ho, int Wo)
{
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
            f32x4 v = *(const f32x4*)(in + (((long long)s * H + iy) * W + ix) * C + c);
#pragma unroll
            for (int e = 0; e < 4; e++) m[e] = v[e] > m[e] ? v[e] : m[e];
        }
    }
    *(f32x4*)(out + p * C + c) = m;
}
hipError_t launch_maxpool(const float* in, float* out, int S, int H, int W, int C, int Ho, int Wo, hipStream_t st)
{
    long long total = (long long)S * Ho * Wo * (C >> 2);
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, out, S, H, W, C, Ho,
                       Wo);
    return hipGetLastError();
}

// Bone-length features (vnect_model.py:198-209): feat[p][191+j] = sqrt((dx^2 + dy^2) + dz^2) from the
// delta channels feat[p][128+j], [149+j], [170+j]; channels 212..ld-1 are zero padding for the next conv.
__global__ void bone_kernel(float* feat, long long npix, int ld)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = ld - 191;
    if (idx >= npix * per) return;
    long long p = idx / per;
    int j = (int)(idx - p * per);
    float* f = feat + p * ld;
    float v = 0.f;
    if (j < 21) {
        float x = f[128 + j], y = f[149 + j], z = f[170 + j];
        v = sqrtf((x * x + y * y) + z * z);
    }
    f[191 + j] = v;
}
hipError_t launch_bone(float* feat, long long npix, int ld, hipStream_t st)
{
    long long total = npix * (ld - 191);
    hipLaunchKernelGGL(bone_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, feat, npix, ld);
    return hipGetLastError();
}

}  // namespace vnect
