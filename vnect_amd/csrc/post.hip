// post.hip -- input pyramid and joint extraction kernels (everything of VNectEstimator.__call__ that is
// not the conv stack).  Built with -ffp-contract=off: each multiply and add rounds on its own, exactly
// like the numpy / CPython / OpenCV arithmetic of /root/reference/src/estimator.py:70-81,105-139,
// src/utils.py:13-21,58-79,153-219 and src/OneEuroFilter.py:13-75, so results are bit-identical to it.
#include "kernels.h"

namespace vnect {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// 8-bit bilinear sample, OpenCV fixed-point form (HResizeLinear<uchar,int,short,2048> then
// VResizeLinear<uchar,...,FixedPtCast<int,uchar,22>>): src rows are `pitch` bytes apart, 3 channels.
__device__ __forceinline__ void sample_u8x3(const uint8_t* src, long long pitch, const ResizeTab& t, int dy, int dx,
                                            int out[3])
{
    const uint8_t* S0 = src + (long long)t.sy0[dy] * pitch;
    const uint8_t* S1 = src + (long long)t.sy1[dy] * pitch;
    const int sx = t.sx[dx] * 3;
    const int b0 = t.b0[dy], b1 = t.b1[dy];
    const bool inner = dx < t.xmax;
    const int a0 = t.a0[dx], a1 = t.a1[dx];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int r0, r1;
        if (inner) {
            r0 = S0[sx + c] * a0 + S0[sx + 3 + c] * a1;
            r1 = S1[sx + c] * a0 + S1[sx + 3 + c] * a1;
        } else {
            r0 = S0[sx + c] * 2048;
            r1 = S1[sx + c] * 2048;
        }
        out[c] = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    }
}

// utils.img_scale_squarify + img_padding (utils.py:82-120): frame (H,W,3) u8 -> 368x368x3 u8 canvas
__global__ void squarify_kernel(const FrameParams* __restrict__ fp, uint8_t* __restrict__ sq)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= BOX) return;
    const ResizeTab& t = fp->sq;
    const int dy = y - fp->offy, dx = x - fp->offx;
    int v[3] = {0, 0, 0};
    if (dy >= 0 && dy < t.dh && dx >= 0 && dx < t.dw) {
        if (t.copy) {
            const uint8_t* p = fp->frame + (long long)dy * fp->row_stride + dx * 3;
            v[0] = p[0], v[1] = p[1], v[2] = p[2];
        } else {
            sample_u8x3(fp->frame, fp->row_stride, t, dy, dx, v);
        }
    }
    uint8_t* o = sq + ((long long)y * BOX + x) * 3;
    o[0] = (uint8_t)v[0], o[1] = (uint8_t)v[1], o[2] = (uint8_t)v[2];
}

// utils.img_scale_padding per scale + `/255 - 0.4` (estimator.py:76-80) -> (S,368,368,4) f32, 4th = 0
template <typename T>
__global__ void pyramid_kernel(const uint8_t* __restrict__ sq, const ScaleTabs* __restrict__ tabs,
                               T* __restrict__ batch4, int scale_base)
{
    typedef T tx4 __attribute__((ext_vector_type(4)));
    // image blockIdx.z of the batch is scale (scale_base + blockIdx.z): a pyramid-sharded rank builds one scale only
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, s = blockIdx.z + scale_base;
    if (x >= BOX) return;
    int v[3] = {0, 0, 0};
    if (!tabs->scaled[s]) {
        const uint8_t* p = sq + ((long long)y * BOX + x) * 3;
        v[0] = p[0], v[1] = p[1], v[2] = p[2];
    } else {
        const ResizeTab& t = tabs->t[s];
        const int dy = y - tabs->pad[s], dx = x - tabs->pad[s];
        if (dy >= 0 && dy < t.dh && dx >= 0 && dx < t.dw) {
            if (t.copy) {
                const uint8_t* p = sq + ((long long)dy * BOX + dx) * 3;
                v[0] = p[0], v[1] = p[1], v[2] = p[2];
            } else {
                sample_u8x3(sq, (long long)BOX * 3, t, dy, dx, v);
            }
        }
    }
    f32x4 o = {tabs->lut[v[0]], tabs->lut[v[1]], tabs->lut[v[2]], 0.f};
    *(tx4*)(batch4 + (((long long)blockIdx.z * BOX + y) * BOX + x) * 4) = __builtin_convertvector(o, tx4);
}

hipError_t launch_squarify(const FrameParams* fp, uint8_t* sq, hipStream_t st)
{
    hipLaunchKernelGGL(squarify_kernel, dim3((BOX + 127) / 128, BOX), dim3(128), 0, st, fp, sq);
    return hipGetLastError();
}
hipError_t launch_pyramid(const uint8_t* sq, const ScaleTabs* tabs, void* batch4, int S, int scale_base, int bf16, hipStream_t st)
{
    dim3 g((BOX + 127) / 128, BOX, S);
    if (bf16) hipLaunchKernelGGL(pyramid_kernel<__bf16>, g, dim3(128), 0, st, sq, tabs, (__bf16*)batch4, scale_base);
    else hipLaunchKernelGGL(pyramid_kernel<float>, g, dim3(128), 0, st, sq, tabs, (float*)batch4, scale_base);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// estimator.py:105-129: avg[q][r][c][j] = (1/S) * sum_i crop(cv2.resize(map_i_q, 1/s_i))[r][c][j], f64 sum
__global__ void merge_kernel(const float* __restrict__ maps, const MergeTabs* __restrict__ tabs,
                             double* __restrict__ avg, int S)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * HM * HM * NJ) return;
    const int j = idx % NJ;
    int t = idx / NJ;
    const int c = t % HM;
    t /= HM;
    const int r = t % HM, q = t / HM;
    const int ch = q * NJ + j;
    double acc = 0.0;
    for (int i = 0; i < S; i++) {
        const MergeTab& mt = tabs->t[i];
        const float* M = maps + (long long)i * HM * HM * MAPC + ch;
        float v;
        if (mt.copy) {
            v = M[(r * HM + c) * MAPC];
        } else {
            const int sx = mt.sx[c];
            const float* R0 = M + (long long)mt.sy0[r] * HM * MAPC;
            const float* R1 = M + (long long)mt.sy1[r] * HM * MAPC;
            float r0, r1;
            if (mt.edge[c]) {
                r0 = R0[sx * MAPC];
                r1 = R1[sx * MAPC];
            } else {
                const float a0 = mt.a0[c], a1 = mt.a1[c];
                r0 = R0[sx * MAPC] * a0 + R0[(sx + 1) * MAPC] * a1;
                r1 = R1[sx * MAPC] * a0 + R1[(sx + 1) * MAPC] * a1;
            }
            v = r0 * mt.b0[r] + r1 * mt.b1[r];
        }
        acc += (double)v;
    }
    avg[idx] = acc / (double)S;
}
hipError_t launch_merge(const float* maps, const MergeTabs* tabs, double* avg, int S, hipStream_t st)
{
    const int total = 4 * HM * HM * NJ;
    hipLaunchKernelGGL(merge_kernel, dim3((total + 255) / 256), dim3(256), 0, st, maps, tabs, avg, S);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// utils.extract_2d_joints (utils.py:153-175) without materialising the 368x368 upsample: every
// thread evaluates cv2's f64 bilinear at its pixels and keeps (max value, lowest flat index).
__device__ __forceinline__ bool better(double v, int i, double bv, int bi) { return v > bv || (v == bv && i < bi); }

// One thread = one column x of the virtual 368x368 upsample, for one 46-row slab of one joint.  cv2's resize is
// separable: the horizontal pass h(sy) = M[sy][sx]*a0 + M[sy][sx+1]*a1 is evaluated once per source row and
// reused by the (up to 8) destination rows that blend it, exactly as resize.cpp does it.
__global__ __launch_bounds__(128) void argmax_kernel(const double* __restrict__ avg, const UpTab* __restrict__ up,
                                                     ArgPartial* __restrict__ part)
{
    __shared__ double map[HM * HM];
    __shared__ double sv[128];
    __shared__ int si[128];
    const int j = blockIdx.x, slab = blockIdx.y, xb = blockIdx.z, tid = threadIdx.x;
    for (int p = tid; p < HM * HM; p += 128) map[p] = avg[(long long)p * NJ + j];  // heatmap = avg[0]
    __syncthreads();
    constexpr int ROWS = BOX / ARG_SLABS;
    const int x = xb * 128 + tid;
    double bv = -__builtin_inf();
    int bi = 0x7fffffff;
    if (x < BOX) {
        const int sx = up->sx[x], edge = up->edge[x];
        const double a0 = up->a0[x], a1 = up->a1[x];
        auto hrow = [&](int sy) {
            const double* R = map + sy * HM;
            return edge ? R[sx] : R[sx] * a0 + R[sx + 1] * a1;
        };
        int c0 = -1, c1 = -1;
        double h0 = 0, h1 = 0;
        for (int y = slab * ROWS; y < (slab + 1) * ROWS; y++) {
            const int s0 = up->sy0[y], s1 = up->sy1[y];  // uniform over the block
            if (s0 != c0) {
                h0 = s0 == c1 ? h1 : hrow(s0);
                c0 = s0;
            }
            if (s1 != c1) {
                h1 = s1 == c0 ? h0 : hrow(s1);
                c1 = s1;
            }
            const double v = h0 * up->b0[y] + h1 * up->b1[y];
            if (v > bv) bv = v, bi = y * BOX + x;  // rows ascend: strict > keeps the first maximum of this column
        }
    }
    sv[tid] = bv, si[tid] = bi;
    __syncthreads();
    for (int s = 64; s > 0; s >>= 1) {
        if (tid < s && better(sv[tid + s], si[tid + s], sv[tid], si[tid])) sv[tid] = sv[tid + s], si[tid] = si[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        ArgPartial& o = part[(j * ARG_SLABS + slab) * ARG_XBLOCKS + xb];
        o.v = sv[0];
        o.idx = si[0];
    }
}
hipError_t launch_argmax(const double* avg, const UpTab* up, ArgPartial* part, hipStream_t st)
{
    hipLaunchKernelGGL(argmax_kernel, dim3(NJ, ARG_SLABS, ARG_XBLOCKS), dim3(128), 0, st, avg, up, part);
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
__device__ __forceinline__ void oef_time(Filt& f, double t)
{
    // `if self.__lasttime and timestamp:` -- None and 0.0 are falsy.  t == lasttime is rejected by the host.
    if (f.has_last && f.lasttime != 0.0 && t != 0.0) f.freq = 1.0 / (t - f.lasttime);
    f.lasttime = t;
    f.has_last = 1;
}
__device__ double oef_f64(Filt& f, double x, double t)
{
    oef_time(f, t);
    const double dx = f.x_init ? (x - f.x_y) * f.freq : 0.0;
    const double edx = lowpass(f.dx_init, f.dx_y, f.dx_s, dx, oef_alpha(f.freq, f.dcutoff));
    const double cutoff = f.mincutoff + f.beta * fabs(edx);
    return lowpass(f.x_init, f.x_y, f.x_s, x, oef_alpha(f.freq, cutoff));
}
// The 3-D filters are fed np.float32 scalars (estimator.py:91-93), so numpy's scalar promotion decides
// the arithmetic: numpy 1.x (nep50 = 0): float32 (op) Python float -> float64, float32 - float32 -> float32;
// numpy >= 2 (nep50 = 1): Python floats are weak, everything stays float32.
__device__ float oef_f32(Filt& f, float x, double t, int nep50)
{
    oef_time(f, t);
    const double a_d = oef_alpha(f.freq, f.dcutoff);
    if (!nep50) {
        const float diff = x - (float)f.x_y;
        const double dx = f.x_init ? (double)diff * f.freq : 0.0;
        const double edx = lowpass(f.dx_init, f.dx_y, f.dx_s, dx, a_d);
        const double cutoff = f.mincutoff + f.beta * fabs(edx);
        return (float)lowpass(f.x_init, f.x_y, f.x_s, (double)x, oef_alpha(f.freq, cutoff));
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
    const double a_x = oef_alpha(f.freq, cutoff);
    const float r = f.x_init ? (float)a_x * x + (float)(1.0 - a_x) * (float)f.x_s : x;
    f.x_init = 1, f.x_y = x, f.x_s = r;
    return r;
}

// utils.hm_pt_interp_bilinear (utils.py:58-79), scale 8, map element (r,c) at m[(r*46+c)*21]
__device__ double pt_interp(const double* m, double dst_y, double dst_x)
{
    const double src_x = (dst_x + 0.5) / 8.0 - 0.5;
    const double src_y = (dst_y + 0.5) / 8.0 - 0.5;
    int x0 = (int)src_x, y0 = (int)src_y;  // int(): truncation toward zero
    x0 = x0 < 0 ? 0 : (x0 > HM - 1 ? HM - 1 : x0);  // no-ops for finite filtered joints in [0, 367];
    y0 = y0 < 0 ? 0 : (y0 > HM - 1 ? HM - 1 : y0);  // keep NaN inputs from indexing outside the map
    const int x1 = x0 + 1 < HM - 1 ? x0 + 1 : HM - 1;
    const int y1 = y0 + 1 < HM - 1 ? y0 + 1 : HM - 1;
    const double v0 = (x1 - src_x) * m[(y0 * HM + x0) * NJ] + (src_x - x0) * m[(y0 * HM + x1) * NJ];
    const double v1 = (x1 - src_x) * m[(y1 * HM + x0) * NJ] + (src_x - x0) * m[(y1 * HM + x1) * NJ];
    return (y1 - src_y) * v0 + (src_y - y0) * v1;
}

// estimator.py:132-139 for all 21 joints: arg-max finish, 2-D filter, read-off, root, 3-D filter, un-map
__global__ __launch_bounds__(64) void joints_kernel(const ArgPartial* __restrict__ part, const double* __restrict__ avg,
                                                    FilterBank* fb, const FrameParams* __restrict__ fp, int nep50,
                                                    JointsOut* __restrict__ out)
{
    __shared__ float root[3];
    const int j = threadIdx.x;
    const bool live = j < NJ;
    float p3[3] = {0.f, 0.f, 0.f};
    double row = 0, col = 0;
    if (live) {
        constexpr int NP = ARG_SLABS * ARG_XBLOCKS;
        double bv = part[j * NP].v;
        int bi = part[j * NP].idx;
        for (int s = 1; s < NP; s++) {
            const double v = part[j * NP + s].v;
            const int i = part[j * NP + s].idx;
            if (better(v, i, bv, bi)) bv = v, bi = i;
        }
        if (bi == 0x7fffffff) bi = 0;  // all-NaN map: np.argmax would return the first NaN; documented deviation
        row = (double)(bi / BOX), col = (double)(bi % BOX);
        row = oef_f64(fb->f2[j][0], row, fp->t2d);
        col = oef_f64(fb->f2[j][1], col, fp->t2d);
        const long long P = (long long)HM * HM * NJ;
        for (int k = 0; k < 3; k++) p3[k] = (float)(pt_interp(avg + (k + 1) * P + j, row, col) * 100);
        if (j == 14) root[0] = p3[0], root[1] = p3[1], root[2] = p3[2];
    }
    __syncthreads();
    if (live) {
        for (int k = 0; k < 3; k++) {
            float v = p3[k] - root[k];  // joints_3d -= joints_3d[14, :] in float32
            v = oef_f32(fb->f3[j][k], v, fp->t3d, nep50);
            out->j3d[j * 3 + k] = v;
        }
        out->j2d[j * 2 + 0] = (row - fp->offy) / fp->scaler;
        out->j2d[j * 2 + 1] = (col - fp->offx) / fp->scaler;
    }
    if (j == 0) out->status = 0;
}
hipError_t launch_joints(const ArgPartial* part, const double* avg, FilterBank* fb, const FrameParams* fp, int nep50,
                         JointsOut* out, hipStream_t st)
{
    hipLaunchKernelGGL(joints_kernel, dim3(1), dim3(64), 0, st, part, avg, fb, fp, nep50, out);
    return hipGetLastError();
}

}  // namespace vnect
