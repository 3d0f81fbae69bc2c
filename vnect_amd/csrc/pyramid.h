// pyramid.h -- one pixel of gen_input_batch's (S,368,368) batch, computed from the uint8 frame on demand (device code shared by
// pyramid_kernel in post.hip and the fused stem kernel in stem.hip).  Integer arithmetic only: OpenCV's 11-bit fixed-point
// bilinear form, /root/reference/src/utils.py:13-21,82-150 via src/estimator.py:70-81.
#pragma once
#include "kernels.h"

namespace vnect {

// ---------------------------------------------------------------------------------------------
// 8-bit bilinear sample, OpenCV fixed-point form (HResizeLinear<uchar,int,short,2048> then
// VResizeLinear<uchar,...,FixedPtCast<int,uchar,22>>).  `px(row, col, v)` yields the 3 channels of a source pixel.
template <typename Px>
__device__ __forceinline__ void sample_u8x3(Px px, const ResizeTab& t, int dy, int dx, int out[3])
{
    const int r0y = t.sy0[dy], r1y = t.sy1[dy];
    const int sx = t.sx[dx];
    const int b0 = t.b0[dy], b1 = t.b1[dy];
    const bool inner = dx < t.xmax;
    const int a0 = t.a0[dx], a1 = t.a1[dx];
    int p00[3], p01[3] = {0, 0, 0}, p10[3], p11[3] = {0, 0, 0};
    px(r0y, sx, p00);
    px(r1y, sx, p10);
    if (inner) {
        px(r0y, sx + 1, p01);
        px(r1y, sx + 1, p11);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int r0, r1;
        if (inner) {
            r0 = p00[c] * a0 + p01[c] * a1;
            r1 = p10[c] * a0 + p11[c] * a1;
        } else {
            r0 = p00[c] * 2048;
            r1 = p10[c] * 2048;
        }
        out[c] = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    }
}

// utils.img_scale_squarify + img_padding (utils.py:82-120): pixel (y, x) of the 368x368 canvas, computed from the
// frame on demand (the canvas itself is never materialised)
__device__ __forceinline__ void square_pixel(const FrameParams* __restrict__ fp, const FrameDyn& dyn, int y, int x, int v[3])
{
    const ResizeTab& t = fp->sq;
    const int dy = y - fp->offy, dx = x - fp->offx;
    v[0] = v[1] = v[2] = 0;
    if (dy >= 0 && dy < t.dh && dx >= 0 && dx < t.dw) {
        const uint8_t* frame = dyn.frame;
        const long long pitch = dyn.row_stride;
        auto fpx = [&](int r, int c, int* o) {
            const uint8_t* p = frame + (long long)r * pitch + c * 3;
            o[0] = p[0], o[1] = p[1], o[2] = p[2];
        };
        if (t.copy) fpx(dy, dx, v);
        else sample_u8x3(fpx, t, dy, dx, v);
    }
}

// pixel (y, x) of image `s` of the batch as uint8 BGR (before `/255 - 0.4`): the square itself for scale 1, else
// utils.img_scale_padding's resize of the square, centred on a black canvas (estimator.py:75-80)
__device__ __forceinline__ void pyramid_pixel(const FrameParams* __restrict__ fp, const FrameDyn& dyn, const ScaleTabs* __restrict__ tabs,
                                              int s, int y, int x, int v[3])
{
    v[0] = v[1] = v[2] = 0;
    if (!tabs->scaled[s]) {
        square_pixel(fp, dyn, y, x, v);
    } else {
        const ResizeTab& t = tabs->t[s];
        const int dy = y - tabs->pad[s], dx = x - tabs->pad[s];
        if (dy >= 0 && dy < t.dh && dx >= 0 && dx < t.dw) {
            auto spx = [&](int r, int c, int* o) { square_pixel(fp, dyn, r, c, o); };
            if (t.copy) spx(dy, dx, v);
            else sample_u8x3(spx, t, dy, dx, v);
        }
    }
}

}  // namespace vnect
