// post.hip -- input pyramid and joint extraction kernels (everything of VNectEstimator.__call__ that is
// not the conv stack).  Built with -ffp-contract=off: each multiply and add rounds on its own, exactly
// like the numpy / CPython / OpenCV arithmetic of /root/reference/src/estimator.py:70-81,105-139,
// src/utils.py:13-21,58-79,153-219 and src/OneEuroFilter.py:13-75, so results are bit-identical to it.
#include "kernels.h"
#include "pyramid.h"
#include "axis.h"  // axis_x_at / axis_y_at: one entry of cv2.resize's per-axis tables, computed where it is used

#include <stddef.h>

namespace vnect {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int VNECT_MAX_S = 8;  // VNECT_MAX_SCALES

// gen_input_batch in one kernel (estimator.py:70-81): squarify, utils.img_scale_padding per scale (each stage rounds
// to uint8 exactly like the two cv2.resize calls of the reference) and `/255 - 0.4` -> (S,368,368,4), 4th channel 0
template <typename T>
__global__ void pyramid_kernel(const FrameParams* __restrict__ fp, const FrameDyn dyn, const ScaleTabs* __restrict__ tabs,
                               T* __restrict__ batch4, int scale_base)
{
    typedef T tx4 __attribute__((ext_vector_type(4)));
    // image blockIdx.z of the batch is scale (scale_base + blockIdx.z): a pyramid-sharded rank builds one scale only
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, s = blockIdx.z + scale_base;
    if (x >= BOX) return;
    int v[3];
    pyramid_pixel(fp, dyn, tabs, s, y, x, v);
    f32x4 o = {tabs->lut[v[0]], tabs->lut[v[1]], tabs->lut[v[2]], 0.f};
    store_wt((tx4*)(batch4 + (((long long)blockIdx.z * BOX + y) * BOX + x) * 4), __builtin_convertvector(o, tx4));
}

hipError_t launch_pyramid(const FrameParams* fp, FrameDyn dyn, const ScaleTabs* tabs, void* batch4, int S, int scale_base, int bf16, hipStream_t st)
{
    dim3 g((BOX + 127) / 128, BOX, S);
    if (bf16) hipLaunchKernelGGL(pyramid_kernel<__bf16>, g, dim3(128), 0, st, fp, dyn, tabs, (__bf16*)batch4, scale_base);
    else hipLaunchKernelGGL(pyramid_kernel<float>, g, dim3(128), 0, st, fp, dyn, tabs, (float*)batch4, scale_base);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// vnect_infer's host-to-device copy as a KERNEL on the frame's own stream: rows of `row` bytes from device-mapped PINNED host memory
// (src, stride bytes apart) to the resident frame slot (dst, rows packed).  hipMemcpyAsync of the same 406 KB costs ~21 us of a
// synchronous frame on this runtime (the copy engine and its hand-over to the compute queue); a kernel in front of the stem costs the
// PCIe transfer (~8 us) and one dependent boundary.  Three forms, chosen on the host: whole 16-byte units of one contiguous run, dwords
// per row, and -- for rows that are not dword-aligned -- frame_copy_rows_kernel below.
__global__ __launch_bounds__(256) void frame_copy_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int row, long long stride, int form)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (form == 0) {          // contiguous, both ends 16-byte aligned: the frame as a run of 16-byte units (+ a byte tail)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const long long n = (long long)H * row, n16 = n >> 4;
        if (i < n16) ((u32x4*)dst)[i] = ((const u32x4*)src)[i];
        if (i < (n & 15)) dst[(n16 << 4) + i] = src[(n16 << 4) + i];
    } else {                  // dword-aligned rows
        const int rw = row >> 2;
        const long long y = i / rw;
        const int x = (int)(i - y * rw);
        if (y < H) ((unsigned*)(dst + y * row))[x] = ((const unsigned*)(src + y * stride))[x];
    }
}
// Rows at ANY byte alignment (advisor, round 5): a crop the tracking loop cuts out of a pinned capture buffer starts at byte 3 * x0 of a
// frame row and is 3 * w bytes wide (/root/reference/run_estimator_ps.py:88), so three crops in four are not dword-aligned -- and the
// byte-per-thread form they used to take read the pinned buffer over PCIe one byte per lane.  Here every lane loads ONE aligned source
// dword (a wave: 64 consecutive dwords = whole PCIe read requests), takes its neighbour's through a lane shuffle, and v_alignbyte
// assembles the destination dword; a wave writes 63 destination dwords.  The destination rows are packed (row bytes apart), so a row
// starts at byte (y * row) & 3 of the destination's dword grid: interior dwords are whole stores, the first / last dword of a row is
// shared with its neighbours and written byte by byte.  grid = (ceil(dwords per row / 252), H), 256 threads.
__global__ __launch_bounds__(256) void frame_copy_rows_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int row, long long stride,
                                                              const uint8_t* src_end)
{
    const int y = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uintptr_t s = (uintptr_t)src + (unsigned long long)y * (unsigned long long)stride;  // first source byte of the row
    const uintptr_t d = (uintptr_t)dst + (unsigned long long)y * (unsigned long long)row;      // first destination byte
    const uintptr_t d0 = d & ~(uintptr_t)3;                                                    // the destination's dword grid
    const int ad = (int)(d - d0);                   // bytes of destination dword 0 in front of the row
    const int ndw = (ad + row + 3) >> 2;            // destination dwords the row touches
    const int k0 = ((int)blockIdx.x * 4 + wave) * 63;  // this wave's first destination dword (wave-uniform)
    if (k0 >= ndw) return;
    // destination dword k holds row bytes [4k - ad, 4k - ad + 4): its first source byte is s + 4k - ad
    const uintptr_t sb = s + 4ull * (unsigned)k0 - (unsigned)ad;
    const uintptr_t sa = sb & ~(uintptr_t)3;
    const int sh = (int)(sb - sa);
    const uintptr_t mine = sa + 4ull * (unsigned)lane;
    // (an aligned dword may start up to 3 bytes in front of the row and end up to 3 bytes behind it: inside the pinned buffer, whose
    // base and capacity are multiples of 4 -- `src_end` is its end; bytes outside the row are never stored)
    unsigned lo = 0;
    if (mine + 4 <= (uintptr_t)src_end && mine < s + (unsigned)row + 4) lo = *(const unsigned*)mine;
    const unsigned hi = (unsigned)__shfl_down((int)lo, 1);
    const unsigned v = sh == 0 ? lo : (sh == 1 ? __builtin_amdgcn_alignbyte(hi, lo, 1) : (sh == 2 ? __builtin_amdgcn_alignbyte(hi, lo, 2) : __builtin_amdgcn_alignbyte(hi, lo, 3)));
    const int k = k0 + lane;
    if (lane == 63 || k >= ndw) return;
    const int b0 = 4 * k - ad;                      // row byte index of the dword's byte 0
    if (b0 >= 0 && b0 + 4 <= row) {
        ((unsigned*)d0)[k] = v;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (b0 + j >= 0 && b0 + j < row) ((uint8_t*)d0)[4 * k + j] = (uint8_t)(v >> (8 * j));
    }
}
// `src_end`: end of the pinned buffer `src_dev` lies in (device address; only the any-alignment form reads it)
hipError_t launch_frame_copy(const uint8_t* src_dev, uint8_t* dst, int H, int row, long long stride, const uint8_t* src_end, hipStream_t st)
{
    const long long n = (long long)H * row;
    if (stride == row && (((uintptr_t)src_dev | (uintptr_t)dst) & 15) == 0) {
        const long long threads = (n >> 4) > 16 ? (n >> 4) : 16;
        hipLaunchKernelGGL(frame_copy_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, src_dev, dst, H, row, stride, 0);
    } else if (((row | stride) & 3) == 0 && (((uintptr_t)src_dev | (uintptr_t)dst) & 3) == 0) {
        hipLaunchKernelGGL(frame_copy_kernel, dim3((unsigned)(((n >> 2) + 255) / 256)), dim3(256), 0, st, src_dev, dst, H, row, stride, 1);
    } else {
        const int ndw_max = (3 + row + 3) >> 2;
        for (int y0 = 0; y0 < H; y0 += 65532) {  // grid.y is 16 bits (65532: a multiple of 4, so every chunk's destination keeps the row phase)
            const int hh = H - y0 < 65532 ? H - y0 : 65532;
            hipLaunchKernelGGL(frame_copy_rows_kernel, dim3((unsigned)((ndw_max + 251) / 252), (unsigned)hh), dim3(256), 0, st, src_dev + (long long)y0 * stride,
                               dst + (long long)y0 * row, hh, row, stride, src_end);
        }
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// estimator.py:105-129: one cell of a merged map,
//   avg_q[r][c][j] = (1/S) * sum_i crop(cv2.resize(map_i_q, fx=fy=1/s_i))[r][c][j]   (f32 interpolation, f64 sum)
// evaluated on demand: the four 46x46x21 f64 averages of the reference are never materialised.  Round 3: the taps and weights are
// computed here (axis.h) from the geometry in the kernel arguments -- the kernels used to stage 4 KB of merge tables into LDS and
// load the x8 upsample's columns from global memory before they could request the first map value; now that request comes first.
// SMAX: compile-time bound of the scale loops (3 covers the reference's and BASELINE's pyramids; these kernels run their
// code once, cold, so its size is their time: the 8-scale form is twice as long)
template <int SMAX>
__device__ __forceinline__ double merged_cell(const float* __restrict__ maps, const MergeGeo& geo, int ch, int r, int c)
{
    // All taps of all scales are requested before any is used (no load sits behind a data-dependent branch), so a
    // thread pays one memory round trip, not one per scale.  For a scale that is a plain copy (size unchanged) the
    // entries are the identity: tap (s0(r), s0(c)) IS element (r, c).
    const int S = geo.S;
    float p00[SMAX], p01[SMAX], p10[SMAX], p11[SMAX];
    AxE X[SMAX], Y[SMAX];
#pragma unroll
    for (int i = 0; i < SMAX; i++) {
        if (i < S) {
            X[i] = axis_x_at(c + geo.off[i], HM, geo.scale[i]), Y[i] = axis_y_at(r + geo.off[i], HM, geo.scale[i]);
            const float* M = maps + (long long)i * HM * HM * MAPC + ch;
            const float* R0 = M + (long long)Y[i].s0 * HM * MAPC;
            const float* R1 = M + (long long)Y[i].s1 * HM * MAPC;
            p00[i] = R0[X[i].s0 * MAPC];
            p01[i] = p10[i] = p11[i] = 0.f;
            if (!geo.copy[i])  // (uniform: a scale that is a plain copy uses tap (s0, s0) alone -- a quarter of its requests)
                p01[i] = R0[X[i].s1 * MAPC], p10[i] = R1[X[i].s0 * MAPC], p11[i] = R1[X[i].s1 * MAPC];
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < SMAX; i++) {
        if (i < S) {
            const float a1 = X[i].f, a0 = 1.f - a1, b1 = Y[i].f, b0 = 1.f - b1;
            const float r0 = X[i].edge ? p00[i] : p00[i] * a0 + p01[i] * a1;
            const float r1 = X[i].edge ? p10[i] : p10[i] * a0 + p11[i] * a1;
            const float v = geo.copy[i] ? p00[i] : r0 * b0 + r1 * b1;
            acc += (double)v;
        }
    }
    return acc / (double)S;
}

// ---------------------------------------------------------------------------------------------
// utils.extract_2d_joints (utils.py:153-175) for one joint per workgroup, without materialising the 368x368 f64
// upsample (22.7 MB per frame in the reference): the merged heat-map of the joint is copied to LDS, then one thread per
// (column x, half of the rows) evaluates cv2's separable f64 bilinear -- the horizontal pass h(sy) = M[sy][sx]*a0 +
// M[sy][sx+1]*a1 once per source row, reused by the destination rows that blend it, exactly as resize.cpp does --
// and keeps (max value, lowest flat index) = np.argmax's first maximum.
__device__ __forceinline__ bool better(double v, int i, double bv, int bi) { return v > bv || (v == bv && i < bi); }

#ifndef ARG_ROWSPLIT
#define ARG_ROWSPLIT 1  // thread groups that share a workgroup's row segments (A/B builds: 2 = 768 threads, each column's segments in two halves)
#endif
constexpr int ARG_COLS = 384;                       // one thread per column (368 used) ...
constexpr int ARG_THREADS = ARG_COLS * ARG_ROWSPLIT;  // ... times the row split
// Workgroups per joint and 8-row segments per workgroup (47 segments in all): 8 x 6 = 168 workgroups keep the f64 work off one CU.  Round 5,
// in-frame averages of post_kernel under rocprofv3 (tools/kernel_avg.sh, -DARG_SLABS=n builds, one call): 4 slabs 14.5 us, 6 12.8, 8 12.4-12.5,
// 16 13.1, 24 15.3 -- every workgroup more is one more arrival at the ticket, every workgroup fewer doubles the f64 blends per column.
#ifndef ARG_SLABS
#define ARG_SLABS 8
#endif
static_assert(ARG_SLABS <= ARG_SLABS_MAX, "the partials' buffer is sized for ARG_SLABS_MAX slabs per joint");
constexpr int ARG_SEGS = (47 + ARG_SLABS - 1) / ARG_SLABS;

// (max value, lowest flat index) over the 64 lanes of a wave by butterfly exchanges (DPP / ds_bpermute under __shfl_xor): every lane
// ends with the wave's winner.  `better` is a total order on (v, i) pairs with distinct i, so the result does not depend on the
// exchange pattern.
__device__ __forceinline__ void wave_argmax(double& v, int& i)
{
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const double ov = __shfl_xor(v, m, 64);
        const int oi = __shfl_xor(i, m, 64);
        if (better(ov, oi, v, i)) v = ov, i = oi;
    }
}

#ifndef POST_DBG
#define POST_DBG 0  // tuning builds only (tools/post_breakdown.sh; wrong results): 1 = no joints stage, 2 = also no merge (no map loads),
                    // 3 = also no x8 upsample / arg-max arithmetic, 4 = an empty kernel (the launch's floor)
#endif
template <int SMAX>
__device__ __forceinline__ void argmax_body(const float* __restrict__ maps, const MergeGeo& geo, ArgPartial* __restrict__ part)
{
    __shared__ double map[HM * HM];
    __shared__ double wv[ARG_THREADS / 64];
    __shared__ int wi[ARG_THREADS / 64];
    const int j = blockIdx.x, slab = blockIdx.y, tid = threadIdx.x;
    // this thread's column of the x8 upsample (utils.py:169-171: cv2.resize(hm f64, fx=fy=8); hostplan.h: build_up_tab)
    const int x = tid % ARG_COLS, part_of = tid / ARG_COLS, xc = x < BOX ? x : BOX - 1;
    const AxE ux = axis_x_at(xc, HM, 1.0 / 8.0);
    const int sx = ux.s0, edge = ux.edge;
    const double a0 = (double)(1.f - ux.f), a1 = (double)ux.f;
    {   // the multi-scale merge (estimator.py:105-129) of exactly the heat-map rows this slab blends: at most 7 of the 46,
        // one cell per thread -- no separate merge launch, no f64 plane in HBM
        const int g0 = slab * ARG_SEGS, g1 = g0 + ARG_SEGS < 47 ? g0 + ARG_SEGS : 47;
        const int r_lo = g0 > 0 ? g0 - 1 : 0, r_hi = g1 - 1 < HM - 1 ? g1 - 1 : HM - 1;
        const int cells = (r_hi - r_lo + 1) * HM;
        for (int p = tid; p < cells; p += ARG_THREADS) {
            const int r = r_lo + p / HM, c = p % HM;
            map[r * HM + c] = POST_DBG == 2 ? (double)(r * 3 + c) : merged_cell<SMAX>(maps, geo, j, r, c);
        }
    }
    __syncthreads();
    // Row structure of the x8 upsample (checked on the host when a handle is made, build_up_tab): destination row y
    // belongs to segment g = (y + 4) / 8 and phase p = (y + 4) % 8; it blends source rows max(g-1, 0) and min(g, 45)
    // with the weights of row 4 + p.  Segment 0 has phases 4..7 (rows 0..3), segment 46 phases 0..3 (rows 364..367).
    double bv = -__builtin_inf();
    int bi = 0x7fffffff;
    if (x < BOX && POST_DBG != 3) {
        double w0[8], w1[8];
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const float f = axis_y_at(4 + p, HM, 1.0 / 8.0).f;  // (folds to eight constants)
            w0[p] = (double)(1.f - f), w1[p] = (double)f;
        }
        auto hrow = [&](int sy) {
            const double* R = map + sy * HM;
            return edge ? R[sx] : R[sx] * a0 + R[sx + 1] * a1;
        };
        const int s0 = slab * ARG_SEGS, s1 = s0 + ARG_SEGS < 47 ? s0 + ARG_SEGS : 47;  // the slab's segments, shared out over the row split
        const int per = (s1 - s0 + ARG_ROWSPLIT - 1) / ARG_ROWSPLIT;
        const int g0 = s0 + part_of * per, g1 = g0 + per < s1 ? g0 + per : s1;
        double h1 = hrow(g0 > 0 ? g0 - 1 : 0);
        for (int g = g0; g < g1; g++) {
            const double h0 = h1;
            h1 = hrow(g < HM ? g : HM - 1);
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int y = g * 8 - 4 + p;
                const double v = h0 * w0[p] + h1 * w1[p];
                if (y >= 0 && y < BOX && v > bv) bv = v, bi = y * BOX + x;  // rows ascend: strict > keeps the first maximum
            }
        }
    }
    // wave reduction in registers, then ONE cross-wave step through LDS (was: a 9-barrier LDS tree)
    wave_argmax(bv, bi);
    if ((tid & 63) == 0) wv[tid >> 6] = bv, wi[tid >> 6] = bi;
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < ARG_THREADS / 64; w++)
            if (better(wv[w], wi[w], bv, bi)) bv = wv[w], bi = wi[w];
        // write-through (agent scope): post_kernel's last workgroup reads all 168 partials behind an agent-scope ticket
        ArgPartial* q = part + j * ARG_SLABS + slab;
        __hip_atomic_store(&q->v, bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&q->idx, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int SMAX>
__global__ __launch_bounds__(ARG_THREADS) void heat_argmax_kernel(const float* __restrict__ maps,
                                                                  const MergeGeo geo, ArgPartial* __restrict__ part)
{
    argmax_body<SMAX>(maps, geo, part);
}
hipError_t launch_argmax(const float* maps, MergeGeo geo, ArgPartial* part, hipStream_t st)
{
    if (geo.S <= 3) hipLaunchKernelGGL(heat_argmax_kernel<3>, dim3(NJ, ARG_SLABS), dim3(ARG_THREADS), 0, st, maps, geo, part);
    else hipLaunchKernelGGL(heat_argmax_kernel<VNECT_MAX_S>, dim3(NJ, ARG_SLABS), dim3(ARG_THREADS), 0, st, maps, geo, part);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// OneEuroFilter.py:13-75
__device__ __forceinline__ double oef_alpha(double freq, double cutoff)
{
    const double te = 1.0 / freq;
    const double tau = 1.0 / (2 * 3.141592653589793 * cutoff);
    return 1.0 / (1.0 + tau / te);
}
__device__ __forceinline__ double lowpass(int& init, double& y, double& s, double value, double alpha)
{
    const double r = init ? alpha * value + (1.0 - alpha) * s : value;
    init = 1;
    y = value;
    s = r;
    return r;
}
// A filter step in two halves (round 5): what does not depend on the value being filtered -- the frequency update of OneEuroFilter.py:65-66,
// te = 1 / freq and the derivative filter's alpha -- and the rest.  post_kernel computes the first half while it waits for memory, BEFORE it
// knows the arg-max; the arithmetic (every operation and its order) is that of the one-piece form, which is now written in terms of the halves.
struct OefPrep {
    double freq, te, a_d;
};
__device__ __forceinline__ OefPrep oef_prep(const Filt& f, double t)
{
    OefPrep p;
    // `if self.__lasttime and timestamp:` -- None and 0.0 are falsy.  t == lasttime is rejected by the host.
    p.freq = (f.has_last && f.lasttime != 0.0 && t != 0.0) ? 1.0 / (t - f.lasttime) : f.freq;
    p.te = 1.0 / p.freq;
    p.a_d = 1.0 / (1.0 + (1.0 / (2 * 3.141592653589793 * f.dcutoff)) / p.te);  // oef_alpha(freq, dcutoff)
    return p;
}
__device__ __forceinline__ double oef_alpha_te(double te, double cutoff)  // oef_alpha with 1 / freq at hand
{
    const double tau = 1.0 / (2 * 3.141592653589793 * cutoff);
    return 1.0 / (1.0 + tau / te);
}
__device__ __forceinline__ void oef_commit_time(Filt& f, const OefPrep& p, double t)
{
    f.freq = p.freq, f.lasttime = t, f.has_last = 1;
}
__device__ __forceinline__ double oef_f64_fin(Filt& f, const OefPrep& p, double x, double t)
{
    oef_commit_time(f, p, t);
    const double dx = f.x_init ? (x - f.x_y) * f.freq : 0.0;
    const double edx = lowpass(f.dx_init, f.dx_y, f.dx_s, dx, p.a_d);
    const double cutoff = f.mincutoff + f.beta * fabs(edx);
    return lowpass(f.x_init, f.x_y, f.x_s, x, oef_alpha_te(p.te, cutoff));
}
__device__ double oef_f64(Filt& f, double x, double t) { return oef_f64_fin(f, oef_prep(f, t), x, t); }
// The 3-D filters are fed np.float32 scalars (estimator.py:91-93), so numpy's scalar promotion decides
// the arithmetic: numpy 1.x (nep50 = 0): float32 (op) Python float -> float64, float32 - float32 -> float32;
// numpy >= 2 (nep50 = 1): Python floats are weak, everything stays float32.
__device__ __forceinline__ float oef_f32_fin(Filt& f, const OefPrep& p, float x, double t, int nep50)
{
    oef_commit_time(f, p, t);
    const double a_d = p.a_d;
    if (!nep50) {
        const float diff = x - (float)f.x_y;
        const double dx = f.x_init ? (double)diff * f.freq : 0.0;
        const double edx = lowpass(f.dx_init, f.dx_y, f.dx_s, dx, a_d);
        const double cutoff = f.mincutoff + f.beta * fabs(edx);
        return (float)lowpass(f.x_init, f.x_y, f.x_s, (double)x, oef_alpha_te(p.te, cutoff));
    }
    float edx;
    if (!f.x_init) {
        f.dx_init = 1, f.dx_y = 0.0, f.dx_s = 0.0;
        edx = 0.f;
    } else {
        const float dx = (x - (float)f.x_y) * (float)f.freq;
        const float s = (float)a_d * dx + (float)(1.0 - a_d) * (float)f.dx_s;
        f.dx_y = dx, f.dx_s = s;
        edx = s;
    }
    const double cutoff = f.mincutoff + f.beta * fabs((double)edx);
    const double a_x = oef_alpha_te(p.te, cutoff);
    const float r = f.x_init ? (float)a_x * x + (float)(1.0 - a_x) * (float)f.x_s : x;
    f.x_init = 1, f.x_y = x, f.x_s = r;
    return r;
}
__device__ float oef_f32(Filt& f, float x, double t, int nep50) { return oef_f32_fin(f, oef_prep(f, t), x, t, nep50); }

// utils.hm_pt_interp_bilinear (utils.py:58-79), scale 8, on merged map q (channel ch) evaluated cell by cell
template <int SMAX>
__device__ double pt_interp(const float* __restrict__ maps, const MergeGeo& geo, int ch, double dst_y, double dst_x)
{
    const double src_x = (dst_x + 0.5) / 8.0 - 0.5;
    const double src_y = (dst_y + 0.5) / 8.0 - 0.5;
    int x0 = (int)src_x, y0 = (int)src_y;  // int(): truncation toward zero
    x0 = x0 < 0 ? 0 : (x0 > HM - 1 ? HM - 1 : x0);  // no-ops for finite filtered joints in [0, 367];
    y0 = y0 < 0 ? 0 : (y0 > HM - 1 ? HM - 1 : y0);  // keep NaN inputs from indexing outside the map
    const int x1 = x0 + 1 < HM - 1 ? x0 + 1 : HM - 1;
    const int y1 = y0 + 1 < HM - 1 ? y0 + 1 : HM - 1;
    const double m00 = merged_cell<SMAX>(maps, geo, ch, y0, x0), m01 = merged_cell<SMAX>(maps, geo, ch, y0, x1);
    const double m10 = merged_cell<SMAX>(maps, geo, ch, y1, x0), m11 = merged_cell<SMAX>(maps, geo, ch, y1, x1);
    const double v0 = (x1 - src_x) * m00 + (src_x - x0) * m01;
    const double v1 = (x1 - src_x) * m10 + (src_x - x0) * m11;
    return (y1 - src_y) * v0 + (src_y - y0) * v1;
}

// estimator.py:132-139 for all 21 joints in one workgroup, one thread per filter: 42 2-D filters, then 63 read-offs
// (x/y/z maps merged on demand) + root subtraction + 63 3-D filters, then the un-mapping.
// `f2` / `f3`: this thread's filter states, loaded by the caller -- in post_kernel at the very start of EVERY workgroup, so that the one that turns out to be last does not
// start a memory round trip for them then.
template <int SMAX>
__device__ __forceinline__ void joints_body(const ArgPartial* part, const float* __restrict__ maps, const MergeGeo& geo,
                                            FilterBank* fb, Filt& f2, Filt& f3, double scaler, double off, const FrameDyn& dyn,
                                            int nep50, JointsOut* __restrict__ out)
{
    __shared__ double c2[NJ * 2];
    __shared__ float p3[NJ * 3];
    const int t = threadIdx.x;
    const int j2 = t < NJ * 2 ? t >> 1 : 0, k2 = t & 1;
    const int j3 = t < NJ * 3 ? t / 3 : 0, k3 = t < NJ * 3 ? t - 3 * j3 : 0;
    const double t2d = dyn.t2d, t3d = dyn.t3d;
    double pv[ARG_SLABS];
    int pi[ARG_SLABS];
#pragma unroll
    for (int s = 0; s < ARG_SLABS; s++) {  // `sc1` loads: in post_kernel the partials were written by other workgroups of this launch
        pv[s] = __hip_atomic_load(&part[j2 * ARG_SLABS + s].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pi[s] = __hip_atomic_load(&part[j2 * ARG_SLABS + s].idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t < NJ * 2) {
        double bv = pv[0];
        int bi = pi[0];
#pragma unroll
        for (int s = 1; s < ARG_SLABS; s++)  // slabs ascend in row order
            if (better(pv[s], pi[s], bv, bi)) bv = pv[s], bi = pi[s];
        if (bi == 0x7fffffff) bi = 0;  // all-NaN map: np.argmax would return the first NaN; documented deviation
        const double raw = k2 == 0 ? (double)(bi / BOX) : (double)(bi % BOX);  // [row, col]
        c2[t] = oef_f64(f2, raw, t2d);
        fb->f2[j2][k2] = f2;
    }
    __syncthreads();
    if (t < NJ * 3) p3[t] = (float)(pt_interp<SMAX>(maps, geo, (k3 + 1) * NJ + j3, c2[j3 * 2], c2[j3 * 2 + 1]) * 100);
    __syncthreads();
    if (t < NJ * 3) {
        const float v = p3[t] - p3[14 * 3 + k3];  // joints_3d -= joints_3d[14, :] in float32
        out->j3d[t] = oef_f32(f3, v, t3d, nep50);
        fb->f3[j3][k3] = f3;
    }
    if (t < NJ * 2) out->j2d[t] = (c2[t] - off) / scaler;
    if (t == 0) out->status = 0;
}
// this thread's filter states and un-mapping constants (threads past 63 hold copies of joint 0's: never written back)
__device__ __forceinline__ void load_filters(const FilterBank* fb, const FrameParams* __restrict__ fp, Filt& f2, Filt& f3, double& scaler,
                                             double& off)
{
    const int t = threadIdx.x;
    const int j2 = t < NJ * 2 ? t >> 1 : 0, k2 = t & 1;
    const int j3 = t < NJ * 3 ? t / 3 : 0, k3 = t < NJ * 3 ? t - 3 * j3 : 0;
    f2 = fb->f2[j2][k2];
    f3 = fb->f3[j3][k3];
    scaler = fp->scaler;
    off = k2 == 0 ? (double)fp->offy : (double)fp->offx;
}
template <int SMAX>
__global__ __launch_bounds__(128) void joints_kernel(const ArgPartial* __restrict__ part, const float* __restrict__ maps,
                                                     const MergeGeo geo, FilterBank* fb,
                                                     const FrameParams* __restrict__ fp, const FrameDyn dyn, int nep50,
                                                     JointsOut* __restrict__ out)
{
    // Every global read that does not depend on a computed value is requested up front (both filter states, frame
    // parameters; the partials at the top of joints_body): the kernel is one workgroup and is priced in memory round trips.
    Filt f2, f3;
    double scaler, off;
    load_filters(fb, fp, f2, f3, scaler, off);
    if (dyn.xfail && __hip_atomic_load(dyn.xfail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == dyn.xseq) {
        if (threadIdx.x == 0) out->status = 1;  // as in post_kernel: THIS frame's exchange timed out -- stale maps, the filters stay
        return;
    }
    joints_body<SMAX>(part, maps, geo, fb, f2, f3, scaler, off, dyn, nep50, out);
}

// The joints stage of post_kernel, spread over the last arriver's six waves (round 5; joints_body above is the 128-thread form of the
// stand-alone joints_kernel and gives the same bits).  The one-wave form cost ~7 of the launch's 15 us: ~2 500 instructions on ONE
// wave's dependent stream -- 4 merged cells x 3 scales of axis arithmetic per read-off, two filter steps with their f64 divisions --
// behind three memory round trips.  Here: wave 0 = the 42 2-D filters, wave 1 = the 63 3-D filters, waves 2..5 = one thread per
// (read-off, corner) = 252 merged cells, each evaluated once; the value-independent half of every filter step (oef_prep) is computed
// by every workgroup while it waits for its partial's store, before the ticket says who is last.
struct WideRole {
    int j2, k2, j3, k3;
};
__device__ __forceinline__ WideRole wide_role()
{
    const int t = threadIdx.x, u = t - 64;
    WideRole r;
    r.j2 = t < NJ * 2 ? t >> 1 : 0, r.k2 = t & 1;
    r.j3 = (u >= 0 && u < NJ * 3) ? u / 3 : 0, r.k3 = (u >= 0 && u < NJ * 3) ? u - 3 * r.j3 : 0;
    return r;
}
template <int SMAX>
__device__ __forceinline__ void joints_stage_wide(const ArgPartial* part, const float* __restrict__ maps, const MergeGeo& geo, FilterBank* fb,
                                                  Filt& f2, Filt& f3, const OefPrep& pr2, const OefPrep& pr3, double scaler, double off,
                                                  const FrameDyn& dyn, int nep50, JointsOut* __restrict__ out)
{
    __shared__ double c2[NJ * 2];
    __shared__ double mc[NJ * 3 * 4];
    __shared__ float p3[NJ * 3];
    const int t = threadIdx.x, u = t - 64, v = t - 128;
    const WideRole ro = wide_role();
    if (t < NJ * 2) {  // wave 0: arg-max of the 8 slab partials, 2-D filter step
        double pv[ARG_SLABS];
        int pi[ARG_SLABS];
#pragma unroll
        for (int sl = 0; sl < ARG_SLABS; sl++) {  // `sc1` loads: the partials were written by other workgroups of this launch
            pv[sl] = __hip_atomic_load(&part[ro.j2 * ARG_SLABS + sl].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pi[sl] = __hip_atomic_load(&part[ro.j2 * ARG_SLABS + sl].idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        double bv = pv[0];
        int bi = pi[0];
#pragma unroll
        for (int sl = 1; sl < ARG_SLABS; sl++)  // slabs ascend in row order
            if (better(pv[sl], pi[sl], bv, bi)) bv = pv[sl], bi = pi[sl];
        if (bi == 0x7fffffff) bi = 0;  // all-NaN map: np.argmax would return the first NaN; documented deviation
        const double raw = ro.k2 == 0 ? (double)(bi / BOX) : (double)(bi % BOX);  // [row, col]
        c2[t] = oef_f64_fin(f2, pr2, raw, dyn.t2d);
        fb->f2[ro.j2][ro.k2] = f2;
    }
    __syncthreads();
    // utils.hm_pt_interp_bilinear (utils.py:58-79) of read-off r = (joint, map) at the filtered 2-D joint: its source coordinates and corners
    auto corners = [&](int r, double& src_x, double& src_y, int& x0, int& y0, int& x1, int& y1) {
        const int j = r / 3;
        src_x = (c2[j * 2 + 1] + 0.5) / 8.0 - 0.5;
        src_y = (c2[j * 2] + 0.5) / 8.0 - 0.5;
        x0 = (int)src_x, y0 = (int)src_y;  // int(): truncation toward zero
        x0 = x0 < 0 ? 0 : (x0 > HM - 1 ? HM - 1 : x0);  // no-ops for finite filtered joints in [0, 367];
        y0 = y0 < 0 ? 0 : (y0 > HM - 1 ? HM - 1 : y0);  // keep NaN inputs from indexing outside the map
        x1 = x0 + 1 < HM - 1 ? x0 + 1 : HM - 1;
        y1 = y0 + 1 < HM - 1 ? y0 + 1 : HM - 1;
    };
    if (v >= 0 && v < NJ * 3 * 4) {  // waves 2..5: one merged cell each
        const int r = v >> 2, cell = v & 3, j = r / 3, k = r - 3 * j;
        double sx, sy;
        int x0, y0, x1, y1;
        corners(r, sx, sy, x0, y0, x1, y1);
        mc[v] = merged_cell<SMAX>(maps, geo, (k + 1) * NJ + j, (cell & 2) ? y1 : y0, (cell & 1) ? x1 : x0);
    }
    __syncthreads();
    if (u >= 0 && u < NJ * 3) {  // wave 1: the bilinear blend of pt_interp, x 100 (estimator.py:135), in float32
        double src_x, src_y;
        int x0, y0, x1, y1;
        corners(u, src_x, src_y, x0, y0, x1, y1);
        const double m00 = mc[u * 4], m01 = mc[u * 4 + 1], m10 = mc[u * 4 + 2], m11 = mc[u * 4 + 3];
        const double v0 = (x1 - src_x) * m00 + (src_x - x0) * m01;
        const double v1 = (x1 - src_x) * m10 + (src_x - x0) * m11;
        p3[u] = (float)(((y1 - src_y) * v0 + (src_y - y0) * v1) * 100);
    }
    __syncthreads();
    if (u >= 0 && u < NJ * 3) {
        const float d = p3[u] - p3[14 * 3 + ro.k3];  // joints_3d -= joints_3d[14, :] in float32
        out->j3d[u] = oef_f32_fin(f3, pr3, d, dyn.t3d, nep50);
        fb->f3[ro.j3][ro.k3] = f3;
    }
    if (t < NJ * 2) out->j2d[t] = (c2[t] - off) / scaler;
    if (t == 0) out->status = 0;
}

// Both in ONE launch (round 2): the 168 arg-max workgroups publish their partials write-through and take an agent-scope ticket; the
// workgroup whose ticket comes last -- every partial is then in memory -- runs the joints stage (filters, read-off, un-mapping).
// Nobody waits: the other workgroups are gone by then.  The hand-off is, cell for cell, the first row of
// MI355X_MICROARCH.md's table of hand-offs measured valid with `sc1` loads in place of an acquire: ONE lane per storing workgroup stores
// its bytes `sc1` (8- and 4-byte), waits vmcnt(0), adds to ONE unsharded agent-scope counter; the workgroup whose add came last loads
// (`sc1`, 8- and 4-byte) only after its add has returned, its other waves behind a workgroup barrier.  An explicit release / acquire
// pair instead (buffer_wbl2 sc1 / buffer_inv sc1) is priced at ~1.7 us EACH in the same guide, on the critical path of a kernel
// that is nothing but a chain of round trips -- so the relaxed form stays, and this comment is what keeps it honest: any change
// to the store / wait / add / load sequence must be checked against that table again.
// Round 3: every workgroup requests what the joints stage needs that does not depend on the arg-max (filter states and un-mapping
// constants -> registers) before anything else, so the last arriver starts no memory round trip for them;
// the resize geometry travels in the kernel arguments (MergeGeo) and every table entry is computed where it is used; on a pyramid-sharded handle a frame whose exchange failed (dyn.xfail) skips the
// joints stage -- the filter banks do not advance on stale maps -- and reports status 1.
// Round 5: the joints stage runs on all six waves of the last arriver (joints_stage_wide), and the value-independent half of the filter
// steps is computed here, in every workgroup, between the partial's store and the wait for it.
template <int SMAX>
__global__ __launch_bounds__(ARG_THREADS) void post_kernel(const float* __restrict__ maps, const MergeGeo geo, ArgPartial* part, unsigned* ticket,
                                                           FilterBank* fb, const FrameParams* __restrict__ fp, const FrameDyn dyn,
                                                           int nep50, JointsOut* __restrict__ out)
{
    __shared__ int last;
    if (POST_DBG == 4) return;
    // this thread's filter states and un-mapping constants in the roles of joints_stage_wide (other threads hold copies of filter 0's: never written back)
    const WideRole ro = wide_role();
    Filt f2 = fb->f2[ro.j2][ro.k2], f3 = fb->f3[ro.j3][ro.k3];
    const double scaler = fp->scaler, off = ro.k2 == 0 ? (double)fp->offy : (double)fp->offx;
    argmax_body<SMAX>(maps, geo, part);
    const OefPrep pr2 = oef_prep(f2, dyn.t2d), pr3 = oef_prep(f3, dyn.t3d);  // (thread 0: behind its partial's stores, in front of the wait for them)
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's `sc1` stores of the partial have left (write-through)
        const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = old == gridDim.x * gridDim.y - 1;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next frame
    }
    __syncthreads();  // the other waves of the last arriver load the partials (sc1) only behind this barrier
    if (!last || (POST_DBG >= 1 && POST_DBG <= 3)) return;
    if (dyn.xfail && __hip_atomic_load(dyn.xfail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == dyn.xseq) {
        if (threadIdx.x == 0) out->status = 1;  // the exchange of THIS frame timed out: stale maps, leave the filters alone
        return;
    }
    joints_stage_wide<SMAX>(part, maps, geo, fb, f2, f3, pr2, pr3, scaler, off, dyn, nep50, out);
}
hipError_t launch_post(const float* maps, MergeGeo geo, ArgPartial* part, unsigned* ticket, FilterBank* fb, const FrameParams* fp,
                       FrameDyn dyn, int nep50, JointsOut* out, hipStream_t st)
{
    if (geo.S <= 3) hipLaunchKernelGGL(post_kernel<3>, dim3(NJ, ARG_SLABS), dim3(ARG_THREADS), 0, st, maps, geo, part, ticket, fb, fp, dyn, nep50, out);
    else hipLaunchKernelGGL(post_kernel<VNECT_MAX_S>, dim3(NJ, ARG_SLABS), dim3(ARG_THREADS), 0, st, maps, geo, part, ticket, fb, fp, dyn, nep50, out);
    return hipGetLastError();
}
hipError_t launch_joints(const ArgPartial* part, const float* maps, MergeGeo geo, FilterBank* fb, const FrameParams* fp, FrameDyn dyn,
                         int nep50, JointsOut* out, hipStream_t st)
{
    if (geo.S <= 3) hipLaunchKernelGGL(joints_kernel<3>, dim3(1), dim3(128), 0, st, part, maps, geo, fb, fp, dyn, nep50, out);
    else hipLaunchKernelGGL(joints_kernel<VNECT_MAX_S>, dim3(1), dim3(128), 0, st, part, maps, geo, fb, fp, dyn, nep50, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The pyramid exchange by peer writes (kernels.h: XchgArgs).  Correctness never depends on timing: a slot is read only after
// its flag carries this frame's sequence number, the flag is written after a system-scope release behind ALL of the slot's
// stores (last-arriver ticket over the 8 chunk-workgroups), and every wait is bounded.
typedef float f32x4g __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sys(f32x4g* p, f32x4g v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");  // system scope, write-through (s_nop: kernels.h, store_wt)
#else
    *p = v;
#endif
}
// system-scope 16-byte load (sc0 sc1: served by memory, not by a cache line another agent may have made stale); a buffer load
// so that the compiler tracks its completion itself and several can be in flight
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4g load_sys(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return __builtin_bit_cast(f32x4g, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 1 | 16));
}
#endif
__global__ __launch_bounds__(256) void exchange_kernel(const XchgArgs a)
{
    const int p = blockIdx.x / XCHG_CHUNKS, c = blockIdx.x % XCHG_CHUNKS, tid = threadIdx.x;
    constexpr int N4 = XCHG_MAPS / 4;                       // 44 436 16-byte units per slot
    constexpr int PER = (N4 + XCHG_CHUNKS - 1) / XCHG_CHUNKS;
    const int u0 = c * PER, u1 = u0 + PER < N4 ? u0 + PER : N4;
    const f32x4g* src = (const f32x4g*)a.src;
    // ---- push: chunk c of my maps -> slot `rank` of peer p's block
    f32x4g* dst = (f32x4g*)(a.block[p] + ((size_t)a.parity * 8 + a.rank) * XCHG_MAPS * sizeof(float));
    for (int u = u0 + tid; u < u1; u += 256) store_sys(dst + u, src[u]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int ok;
    if (tid == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);  // system scope: the stores above are visible to p before the flag can be
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(a.tickets + p, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == XCHG_CHUNKS - 1) {  // every chunk of this slot has been stored and released: publish
            __hip_atomic_store(a.tickets + p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned* flag = (unsigned*)(a.block[p] + XCHG_FLAG_OFF + (size_t)a.parity * 128) + a.rank;
            __hip_atomic_store(flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // ---- pull: wait for peer p's slot in MY block
        const unsigned* mine = (const unsigned*)(a.block[a.rank] + XCHG_FLAG_OFF + (size_t)a.parity * 128) + p;
        unsigned seen = 0, spins = 0;
        // relaxed polls (an acquire per poll would invalidate this CU's cache every time), ONE acquire once the flag matches
        while ((seen = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != a.seq && spins < a.spin_limit) {
            __builtin_amdgcn_s_sleep(32);
            spins++;
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        ok = seen == a.seq;
        if (!ok) {
            *a.status = 1;
            __hip_atomic_store(a.dfail, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!ok) return;
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t from = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.block[a.rank] + ((size_t)a.parity * 8 + p) * XCHG_MAPS * sizeof(float)), 0, XCHG_MAPS * 4, 0x00020000);
    f32x4g* to = (f32x4g*)(a.gather + (size_t)p * XCHG_MAPS);
    for (int u = u0 + tid; u < u1; u += 256) to[u] = load_sys(from, u * 16);
#endif
}
hipError_t launch_exchange(const XchgArgs& a, hipStream_t st)
{
    hipLaunchKernelGGL(exchange_kernel, dim3(a.nranks * XCHG_CHUNKS), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// VNectEstimator.joint_filter (estimator.py:83-95) on its own: the handle's filter bank applied to caller-supplied joints,
// one thread per filter.  Bank `dim` (2: 21 x 2 filters, 3: 21 x 3); f32vals: the values are numpy float32 scalars (what
// the reference feeds the 3-D bank, estimator.py:91-93), so numpy's scalar promotion decides the arithmetic (oef_f32);
// otherwise plain float64 (oef_f64).  in / out may be device-mapped pinned host memory.
__global__ __launch_bounds__(64) void filter_kernel(FilterBank* fb, int dim, int f32vals, int nep50, double t,
                                                    const double* __restrict__ in, double* __restrict__ out)
{
    const int i = threadIdx.x;
    if (i >= NJ * dim) return;
    Filt* fp = dim == 2 ? &fb->f2[i >> 1][i & 1] : &fb->f3[i / 3][i % 3];
    Filt f = *fp;
    out[i] = f32vals ? (double)oef_f32(f, (float)in[i], t, nep50) : oef_f64(f, in[i], t);
    *fp = f;
}
hipError_t launch_filter(FilterBank* fb, int dim, bool f32vals, int nep50, double t, const double* in, double* out, hipStream_t st)
{
    hipLaunchKernelGGL(filter_kernel, dim3(1), dim3(64), 0, st, fb, dim, (int)f32vals, nep50, t, in, out);
    return hipGetLastError();
}

// What this translation unit was compiled with (vnect_build_info; conv.hip has the twin of this).
#define VNECT_STR2(x) #x
#define VNECT_STR(x) VNECT_STR2(x)
const char* post_build_probes() { return "POST_DBG=" VNECT_STR(POST_DBG) " ARG_SLABS=" VNECT_STR(ARG_SLABS) " ARG_ROWSPLIT=" VNECT_STR(ARG_ROWSPLIT); }
bool post_probes_off() { return POST_DBG == 0 && ARG_SLABS == 8 && ARG_ROWSPLIT == 1; }

}  // namespace vnect
