// axis.h -- ONE entry of cv2.resize's per-axis bilinear tables (float and double images: resize.cpp's `fx = (dx + 0.5) * scale_x - 0.5`
// form), computed where it is needed.  Shared by the device (post.hip: the merge and the x8 upsample compute their taps and weights
// themselves -- round 3; they used to load tables) and the host (hostplan.h includes this file and builds its whole tables --
// plan::axis_x / axis_y -- by looping over these two functions, so the sanitizer-tested table code and the kernels cannot drift apart).  The arithmetic is IEEE double / float multiply, subtract, floor and
// convert; post.hip is built with contraction off, so the device gets the host's bits.  HIP-free when compiled by g++.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define VNECT_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define VNECT_HD inline
#endif

namespace vnect {

struct AxE {
    int s0, s1, edge;  // the two taps; edge (x axis only): a column at or past the far border, which takes the single tap s0
    float f;           // weight of s1 (s0 gets 1.f - f)
};
// x axis: offset clamped and fraction zeroed at both borders
VNECT_HD AxE axis_x_at(int d, int ssize, double scale)
{
    float fx = (float)((d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) fx = 0.f, sx = 0;
    const int edge = sx + 1 >= ssize;  // sx is monotonic in d, so this is cv2's `dx >= xmax`
    if (sx >= ssize - 1) fx = 0.f, sx = ssize - 1;
    return AxE{sx, sx + 1 < ssize ? sx + 1 : ssize - 1, edge, fx};
}
// y axis: floor + fraction kept; the two source rows are clipped into the image
VNECT_HD AxE axis_y_at(int d, int ssize, double scale)
{
    float fy = (float)((d + 0.5) * scale - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int s0 = sy < 0 ? 0 : (sy < ssize ? sy : ssize - 1), s1 = sy + 1 < 0 ? 0 : (sy + 1 < ssize ? sy + 1 : ssize - 1);
    return AxE{s0, s1, 0, fy};
}

}  // namespace vnect
