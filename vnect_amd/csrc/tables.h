// tables.h -- constants and plain-data tables shared by the HIP kernels (kernels.h) and the host-side planning code (hostplan.h).
// No HIP in here: hostplan.h and its sanitizer build (make hostplan_asan; tests/test_hostplan.py) compile with plain g++.
#pragma once
#include <stdint.h>

namespace vnect {

constexpr int BOX = 368;
constexpr int HM = 46;
constexpr int NJ = 21;
constexpr int MAPC = 84;       // 4 maps x 21 joints

// ---- pre-processing -------------------------------------------------------------------
struct ResizeTab {  // 8-bit bilinear tables for one destination axis pair (OpenCV fixed point, 11 bits)
    int dh, dw;       // destination size
    int xmax;         // columns >= xmax take S[sx]*2048 (right border)
    int copy;         // destination size == source size: plain copy
    int16_t sx[BOX], a0[BOX], a1[BOX];
    int16_t sy0[BOX], sy1[BOX], b0[BOX], b1[BOX];
};

struct FrameParams {  // crop GEOMETRY: depends on the crop size (H, W) only, so it is uploaded when that changes -- never
                      // for a stream of equally sized crops
    double scaler;
    int offx, offy;
    int H, W;
    ResizeTab sq;          // squarify resize (utils.img_scale_squarify)
};

struct ScaleTabs {  // per handle: pyramid resizes of the 368x368 square (utils.img_scale_padding)
    int S;
    int pad[8];     // leading pad rows/cols per scale
    int scaled[8];  // 1: scale < 1 (resize + pad), 0: the square itself
    ResizeTab t[8];
    float lut[256]; // (float)v / 255 - 0.4 in float32
};


// ---- post-processing ------------------------------------------------------------------
struct MergeTab {  // cv2.resize(map, fx=fy=1/s) restricted to the 46x46 centre crop, per scale
    int copy;
    int sx[HM], edge[HM];     // edge: column >= xmax -> value is S[sx]
    float a0[HM], a1[HM];
    int sy0[HM], sy1[HM];
    float b0[HM], b1[HM];
};
// What the post-processing kernels need to rebuild a MergeTab entry on the fly (round 3: they compute the taps and weights from these few
// numbers with axis.h's two functions, which tests/test_hostplan.py holds to the host's tables, instead of loading 4 KB of tables first): per scale the source
// step of cv2.resize(map, fx = fy = 1 / s) as the host computed it, the centre crop's offset, and whether the resize is a plain copy.
struct MergeGeo {
    int S, pad_;
    double scale[8];
    int off[8], copy[8];
};
struct UpTab {  // x8 upsample tables (utils.extract_2d_joints)
    int sx[BOX], edge[BOX];
    double a0[BOX], a1[BOX];
    int sy0[BOX], sy1[BOX];
    double b0[BOX], b1[BOX];
};

// ---- the fused stem's tiling (stem.hip) -------------------------------------------------
constexpr int STEM_TW = 23;          // pooled columns per tile (92 = 4 x 23)
constexpr int STEM_MAXH = 5;         // pooled rows per tile at most
constexpr int STEM_MAXGROUPS = 92;   // row groups per image at most
constexpr int STEM_PW = 100;         // input-patch row stride in pixels (99 used; even, so a bf16 pixel pair is 16-byte aligned)
constexpr int STEM_REG_ROWS = 64;    // from-the-frame form: the rectangle of frame bytes a tile's patch is made from, in LDS:
constexpr int STEM_REG_PITCH = 640;  // at most 64 rows of 640 bytes (3 bytes per pixel + alignment slack)

}  // namespace vnect
