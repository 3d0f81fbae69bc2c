// hostplan.h -- the pure-host planning logic of libvnect_hip.so: OpenCV-compatible resize tables, the merge / upsample tables of the
// post-processing, weight packing into the kernels' layouts (incl. the transposed conv's four sub-pixel phases), the activation
// arena's first-fit placement, the per-layer tile choice, and the fused stem's row groups.  Header-only and HIP-free on purpose:
// the host runtime (rt_plan.cpp, rt_exec.cpp) uses it for the product, and `make hostplan_asan` builds the same code with g++ -fsanitize=address,undefined behind a
// small C shim (hostplan_capi.cpp) that tests/test_hostplan.py drives on the CPU box -- the ~2 000 lines of the runtime (rt_*.cpp) otherwise
// only ever run next to a GPU.
//
// Reference arithmetic: cv2.resize(INTER_LINEAR) as used by /root/reference/src/utils.py:13-21 (via :107-150) and
// src/estimator.py:112-119, src/utils.py:169-171; TF layouts of src/vnect_model.py:27-217.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "axis.h"
#include "tables.h"

namespace vnect {
namespace plan {

// ---- OpenCV INTER_LINEAR table builders (resize.cpp semantics) ----------------------
inline int cv_round(double v) { return (int)nearbyint(v); }  // round half to even
inline int clipi(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }
inline int16_t sat_short(float v)
{
    int r = (int)nearbyintf(v);
    return (int16_t)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

struct AxisTab {
    std::vector<int> s0, s1, edge;
    std::vector<float> f;
    int xmax = 0;
};
// Whole tables are built by looping over axis.h's single-entry functions -- the ones the kernels call (post.hip: merge, x8 upsample):
// ONE source of this arithmetic for host tables and device code.
// x axis: offset clamped and fraction zeroed at both borders; columns >= xmax use the single tap
inline AxisTab axis_x(int ssize, int dsize, double scale)
{
    AxisTab t;
    t.s0.resize(dsize), t.s1.resize(dsize), t.edge.resize(dsize), t.f.resize(dsize);
    t.xmax = dsize;
    for (int d = 0; d < dsize; d++) {
        const vnect::AxE e = vnect::axis_x_at(d, ssize, scale);
        t.s0[d] = e.s0, t.s1[d] = e.s1, t.edge[d] = e.edge, t.f[d] = e.f;
        if (e.edge) t.xmax = std::min(t.xmax, d);  // the source offset is monotonic in d: edge[d] == (d >= xmax), cv2's test
    }
    return t;
}
// y axis: floor + fraction kept; the two source rows are clipped into the image
inline AxisTab axis_y(int ssize, int dsize, double scale)
{
    AxisTab t;
    t.s0.resize(dsize), t.s1.resize(dsize), t.edge.assign(dsize, 0), t.f.resize(dsize);
    for (int d = 0; d < dsize; d++) {
        const vnect::AxE e = vnect::axis_y_at(d, ssize, scale);
        t.s0[d] = e.s0, t.s1[d] = e.s1, t.f[d] = e.f;
    }
    return t;
}

// cv2.resize(u8 src (sh,sw), (0,0), fx=fy=f): destination size and fixed-point tables
inline bool build_u8_tab(int sh, int sw, double f, ResizeTab* t)
{
    memset(t, 0, sizeof *t);
    if (!(f > 0.0) || sh < 1 || sw < 1) return false;
    const double dwf = sw * f, dhf = sh * f;
    if (!(dwf < 1e6) || !(dhf < 1e6)) return false;  // before the casts below: a hostile factor must not overflow an int
    t->dw = cv_round(dwf), t->dh = cv_round(dhf);
    if (t->dw < 1 || t->dh < 1 || t->dw > BOX || t->dh > BOX) return false;
    t->copy = (t->dw == sw && t->dh == sh);
    const double scale = 1.0 / f;
    AxisTab x = axis_x(sw, t->dw, scale), y = axis_y(sh, t->dh, scale);
    t->xmax = x.xmax;
    for (int d = 0; d < t->dw; d++) {
        t->sx[d] = (int16_t)x.s0[d];
        t->a0[d] = sat_short((1.f - x.f[d]) * 2048.f);
        t->a1[d] = sat_short(x.f[d] * 2048.f);
    }
    for (int d = 0; d < t->dh; d++) {
        t->sy0[d] = (int16_t)y.s0[d], t->sy1[d] = (int16_t)y.s1[d];
        t->b0[d] = sat_short((1.f - y.f[d]) * 2048.f);
        t->b1[d] = sat_short(y.f[d] * 2048.f);
    }
    return true;
}

// One 8-bit bilinear sample through a ResizeTab, exactly as the device code (pyramid.h: sample_u8x3) evaluates it -- the host twin the
// CPU tests compare with the oracle's resize; `px(row, col)` yields a source value.
template <typename Px>
inline int sample_u8(Px px, const ResizeTab& t, int dy, int dx)
{
    if (t.copy) return px(dy, dx);
    const int r0y = t.sy0[dy], r1y = t.sy1[dy], sx = t.sx[dx];
    int r0, r1;
    if (dx < t.xmax) {
        r0 = px(r0y, sx) * t.a0[dx] + px(r0y, sx + 1) * t.a1[dx];
        r1 = px(r1y, sx) * t.a0[dx] + px(r1y, sx + 1) * t.a1[dx];
    } else {
        r0 = px(r0y, sx) * 2048, r1 = px(r1y, sx) * 2048;
    }
    return (((t.b0[dy] * (r0 >> 4)) >> 16) + ((t.b1[dy] * (r1 >> 4)) >> 16) + 2) >> 2;
}

// (float)v / 255 - 0.4 in float32: `batch / 255.0 - 0.4` of estimator.py:79-80
inline void fill_lut(float* lut)
{
    for (int v = 0; v < 256; v++) lut[v] = (float)v / 255.f - 0.4f;
}

// pyramid entry i of the ScaleTabs (utils.img_scale_padding, estimator.py:77: `... if scale < 1 else img_square`).
// Returns nullptr on success, else the reason.
inline const char* build_scale_tab(double s, ScaleTabs* st, int i)
{
    if (!(s > 0.0) || s > 1.0) return "scales must be in (0, 1]";
    st->scaled[i] = s < 1.0;
    st->pad[i] = 0;
    if (st->scaled[i]) {
        if (!build_u8_tab(BOX, BOX, s, &st->t[i])) return "scale too small";
        st->pad[i] = (BOX - st->t[i].dh) / 2;  // utils.py:137-140; the remainder pads the far side
    }
    return nullptr;
}

// estimator.py:112-119: rescale = 1.0 / scale; cv2.resize(map, fx=fy=rescale); centre crop 46x46
inline const char* build_merge_tab(double s, MergeTab* m)
{
    if (!(s > 0.0) || s > 1.0) return "scales must be in (0, 1]";
    const double f = 1.0 / s;
    if (!(HM * f < 1e6)) return "scale too small";
    const int ds = cv_round(HM * f);
    if (ds < HM) return "scale > 1 not supported";
    memset(m, 0, sizeof *m);
    m->copy = ds == HM;
    const double scale = 1.0 / f;
    AxisTab x = axis_x(HM, ds, scale), y = axis_y(HM, ds, scale);
    const int off = ds / 2 - HM / 2;
    for (int r = 0; r < HM; r++) {
        const int d = r + off;
        m->sx[r] = x.s0[d], m->edge[r] = x.edge[d];
        m->a0[r] = 1.f - x.f[d], m->a1[r] = x.f[d];
        m->sy0[r] = y.s0[d], m->sy1[r] = y.s1[d];
        m->b0[r] = 1.f - y.f[d], m->b1[r] = y.f[d];
    }
    return nullptr;
}

// the same resize as the few numbers the device needs to rebuild any table entry itself (tables.h: MergeGeo; axis.h: axis_x_at / axis_y_at)
inline const char* build_merge_geo(double s, MergeGeo* g, int i)
{
    MergeTab m;
    if (const char* why = build_merge_tab(s, &m)) return why;
    const double f = 1.0 / s;
    const int ds = cv_round(HM * f);
    g->scale[i] = 1.0 / f, g->off[i] = ds / 2 - HM / 2, g->copy[i] = ds == HM;
    return nullptr;
}

// utils.py:169-171: cv2.resize(hm (46,46) f64, fx=fy=8) -> 368x368.  Returns false if the table does not have the (segment, phase)
// row structure heat_argmax_kernel relies on.
inline bool build_up_tab(UpTab* u)
{
    AxisTab x = axis_x(HM, BOX, 1.0 / 8.0), y = axis_y(HM, BOX, 1.0 / 8.0);
    for (int d = 0; d < BOX; d++) {
        u->sx[d] = x.s0[d], u->edge[d] = x.edge[d];
        u->a0[d] = (double)(1.f - x.f[d]), u->a1[d] = (double)x.f[d];
        u->sy0[d] = y.s0[d], u->sy1[d] = y.s1[d];
        u->b0[d] = (double)(1.f - y.f[d]), u->b1[d] = (double)y.f[d];
    }
    for (int d = 0; d < BOX; d++) {
        const int g = (d + 4) / 8, ph = (d + 4) % 8;
        if (u->sy0[d] != std::max(g - 1, 0) || u->sy1[d] != std::min(g, HM - 1) || u->b0[d] != u->b0[4 + ph] || u->b1[d] != u->b1[4 + ph])
            return false;
    }
    return true;
}

// utils.img_scale_squarify + img_padding geometry for an (H,W) frame.  nullptr on success, else the reason.
inline const char* squarify(int H, int W, FrameParams* c)
{
    if (H < 1 || W < 1 || H > 8192 || W > 8192) return "frame size out of range";
    memset(c, 0, sizeof *c);
    c->scaler = (double)BOX / std::max(H, W);
    if (!build_u8_tab(H, W, c->scaler, &c->sq)) return "squarify: scaled size exceeds 368";
    const int h2 = c->sq.dh, w2 = c->sq.dw;
    // utils.py:98-103 indexes a 368-long axis with the scaled long side; numpy raises if it is not 368
    if ((h2 > w2 ? h2 : w2) != BOX) return "squarify: scaled long side != 368";
    if (h2 > w2) c->offx = BOX / 2 - w2 / 2;
    else c->offy = BOX / 2 - h2 / 2;
    c->H = H, c->W = W;
    return nullptr;
}

// ---- number formats ------------------------------------------------------------------------------
// fp32 -> bf16, round to nearest even (weights are finite)
inline uint16_t to_bf16(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
inline float from_bf16(uint16_t b)
{
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// fp32 -> three bf16 pieces, hi + mid + lo ~= x to 24 bits (each piece rounded to nearest even from what the previous ones left: the
// pieces may differ in sign).  The split-product conv path (VNECT_FP32_SPLIT) stores its WEIGHTS this way; the kernel splits the
// activations the same way in registers (by truncation, which is exact) and multiplies piece by piece on the bf16 matrix pipe.
inline void split3(float x, uint16_t out[3])
{
    const uint16_t h = to_bf16(x);
    const float r1 = x - from_bf16(h);            // exact: x and its bf16 neighbour share the exponent range
    const uint16_t m = to_bf16(r1);
    const float r2 = r1 - from_bf16(m);           // exact
    out[0] = h, out[1] = m, out[2] = to_bf16(r2);
}
// packed fp32 weights [phases][Npad][K] (K a multiple of 32) -> [phases][K / 32 chunks][3 planes][Npad][32] bf16: chunk-major, so that the
// 64 bytes a plane holds per row and chunk lie NEXT to the neighbouring rows' (a landing of 16 rows x 64 B is one contiguous KiB: whole
// cache lines, like the fp32 layout's 8 rows x 128 B; with the planes stored row by row every request was half a line)
inline void pack_split3(const std::vector<float>& wp, int phases, int Npad, int K, std::vector<uint16_t>& out)
{
    out.assign((size_t)phases * Npad * K * 3, 0);
    const int C = K / 32;
    for (int z = 0; z < phases; z++)
        for (int n = 0; n < Npad; n++)
            for (int c = 0; c < C; c++)
                for (int e = 0; e < 32; e++) {
                    uint16_t pc[3];
                    split3(wp[((size_t)z * Npad + n) * K + c * 32 + e], pc);
                    for (int pl = 0; pl < 3; pl++) out[((((size_t)z * C + c) * 3 + pl) * Npad + n) * 32 + e] = pc[pl];
                }
}

inline void same_pad(int in, int k, int stride, int* out, int* before)
{
    *out = (in + stride - 1) / stride;
    int tot = std::max((*out - 1) * stride + k - in, 0);
    *before = tot / 2;
}
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- weight packing: TF layouts -> the GEMM B operand [Npad][K], K contiguous ------------------------
// k index of filter element (ky, kx, ci) of a tc.layers.conv2d kernel (kh,kw,Cin,Cout):
//   ordinary layers: (ky * k + kx) * cp + ci, cp = the input tensor's pixel stride (channels padded to whole K chunks);
//   conv1 (7x7 on the NHWC4 input): fp32 ky * 32 + kx * 4 + ci (7 chunks of 8 pixels x 4 channels, pixel 7 / channel 3 zero),
//                                   bf16 (ky >> 1) * 64 + (ky & 1) * 32 + kx * 4 + ci (4 chunks of 2 rows, row 7 zero too).
inline size_t conv_kidx(int k, int ky, int kx, int ci, int cp, bool conv1, bool bf16)
{
    if (conv1) return bf16 ? (size_t)(ky >> 1) * 64 + (ky & 1) * 32 + kx * 4 + ci : (size_t)ky * 32 + kx * 4 + ci;
    return (size_t)(ky * k + kx) * cp + ci;
}
// W (k,k,cin,cout) -> wp [Npad][K] starting at row n0 (a paired launch concatenates two layers along N)
inline void pack_conv(const float* W, int k, int cin, int cout, int cp, bool conv1, bool bf16, int K, int n0, std::vector<float>& wp)
{
    for (int ky = 0; ky < k; ky++)
        for (int kx = 0; kx < k; kx++)
            for (int ci = 0; ci < cin; ci++) {
                const float* src = &W[(((size_t)ky * k + kx) * cin + ci) * cout];
                const size_t kidx = conv_kidx(k, ky, kx, ci, cp, conv1, bf16);
                for (int n = 0; n < cout; n++) wp[(size_t)(n0 + n) * K + kidx] = src[n];
            }
}
// A tail GEMM's 1x1 weights (1,1,mid,cout), mid = 64 (tail_gemm) or 128 (tail_wide), in the MFMA's own B-fragment order, so that a
// wave's load instruction reads one contiguous KiB (row by row -- [cout][mid], a lane per row -- the same instruction touched 32 cache
// lines for 32 bytes each, and the wide tail took 16 us instead of 10): [column block of 32][group q][lane = 32 hh + column][e], where
// the lane's elements of group q are k = UQ q + UH hh + e (fp32: UQ = 8, UH = 4 -- the four 32x32x2 MFMAs of a group; bf16: UQ = 16,
// UH = 8 -- one 32x32x16 MFMA).  Columns padded to whole blocks with zero weights.
inline void pack_tail(const float* Wc, int mid, int cout, bool bf16, std::vector<float>& w2)
{
    const int UQ = bf16 ? 16 : 8, UH = bf16 ? 8 : 4, NQ = mid / UQ, nb = round_up(cout, 32) / 32;
    w2.assign((size_t)nb * 32 * mid, 0.f);
    for (int n = 0; n < cout; n++)
        for (int k = 0; k < mid; k++) {
            const int cb = n / 32, col = n % 32, q = k / UQ, hh = (k % UQ) / UH, e = k % UH;
            w2[((((size_t)cb * NQ + q) * 2 + hh) * 32 + col) * UH + e] = Wc[(size_t)k * cout + n];
        }
}

// The two transposed convs (4x4, stride 2, SAME; kernels (kh,kw,Cout,Cin), vnect_model.py:188-196) as 4 sub-pixel phases of ONE
// launch: out[2i-1+ky, 2j-1+kx, oc] += in[i,j,ic] * W[ky,kx,oc,ic]; phase (py,px) = (oy&1, ox&1):
//   py = 0: ky = 1 reads row i', ky = 3 reads row i'-1;  py = 1: ky = 0 reads row i'+1, ky = 2 reads row i'.
// Columns 0..127 = res5c_branch2a (BN + ReLU in the epilogue), 128..190 = res5c_branch1a's 63 deltas.
// wp [4 phases][Npad][4 taps x 256]; dy / dx [phase * 4 + tap].
inline void pack_deconv(const float* W1 /* (4,4,63,256) */, const float* W2 /* (4,4,128,256) */, int Npad, int K, std::vector<float>& wp,
                        int* dy, int* dx)
{
    const int kys[2][2] = {{1, 3}, {0, 2}}, dys[2][2] = {{0, -1}, {1, 0}};
    wp.assign((size_t)4 * Npad * K, 0.f);
    for (int py = 0; py < 2; py++)
        for (int px = 0; px < 2; px++) {
            const int z = py * 2 + px;
            for (int ta = 0; ta < 2; ta++)
                for (int tb = 0; tb < 2; tb++) {
                    const int t = ta * 2 + tb, ky = kys[py][ta], kx = kys[px][tb];
                    dy[z * 4 + t] = dys[py][ta], dx[z * 4 + t] = dys[px][tb];
                    for (int n = 0; n < 191; n++) {
                        const float* src = n < 128 ? &W2[(((size_t)ky * 4 + kx) * 128 + n) * 256] : &W1[(((size_t)ky * 4 + kx) * 63 + (n - 128)) * 256];
                        float* dst = &wp[((size_t)z * Npad + n) * K + (size_t)t * 256];
                        memcpy(dst, src, 256 * sizeof(float));
                    }
                }
        }
}
// FusedBatchNorm inference (contrib batch_norm default epsilon 0.001) as the epilogue's (acc + bias) * scale + shift
inline void fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, int C, int Npad, std::vector<float>& bias,
                    std::vector<float>& scale, std::vector<float>& shift)
{
    bias.assign(Npad, 0.f), scale.assign(Npad, 1.f), shift.assign(Npad, 0.f);
    for (int c = 0; c < C; c++) {
        bias[c] = -mean[c];
        scale[c] = gamma[c] * (1.0f / sqrtf(var[c] + 0.001f));
        shift[c] = beta[c];
    }
}

// ---- activation arena ------------------------------------------------------------------------------
// A tensor lives from the layer that first touches it (`first`) to the last (`last`); tensors with disjoint lifetimes share
// addresses: first fit over the live intervals, in order of first use.  Returns the arena size; off[t] = byte offset of tensor t.
inline size_t arena_first_fit(const std::vector<int>& first, const std::vector<int>& last, const std::vector<size_t>& need,
                              std::vector<size_t>& off)
{
    const int nt = (int)need.size();
    std::vector<int> order(nt);
    for (int i = 0; i < nt; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return first[x] != first[y] ? first[x] < first[y] : x < y; });
    off.assign(nt, 0);
    std::vector<int> placed;
    size_t total = 0;
    for (int t : order) {
        // candidate offsets: 0 and the end of every placed tensor whose lifetime overlaps; take the lowest that fits
        std::vector<std::pair<size_t, size_t>> busy;  // [begin, end) of overlapping placed tensors
        for (int q : placed)
            if (!(last[q] < first[t] || last[t] < first[q])) busy.push_back({off[q], off[q] + need[q]});
        std::sort(busy.begin(), busy.end());
        size_t pos = 0;
        for (auto& b : busy) {
            if (pos + need[t] <= b.first) break;
            pos = std::max(pos, b.second);
        }
        off[t] = pos, total = std::max(total, pos + need[t]);
        placed.push_back(t);
    }
    return total;
}

// ---- tile shape and K split of a layer ---------------------------------------------------------------
// Measured on MI355X with tools/sweep.sh (every layer x {64x64, 128x64, 64x128} x split 1/2/3/5 x ring depth): the 64x64 tile wins
// everywhere (more, smaller workgroups; two per CU), and a 5-way K split pays only where a layer has at most ~half a workgroup per
// CU and a long K loop (the 23x23 stage).  Each K slice writes its own slab and a second kernel sums the slabs in slice order, so
// results stay deterministic.  `force` (VNECT_FORCE_TILE="BM,BN,KG,ks") overrides the choice for every layer that admits it, `plan`
// (VNECT_PLAN="layer=BM,BN,KG,ks;layer=...") for single layers (tools/layer_table.py shows the effect).
struct TileChoice {
    int BM = 64, BN = 64, KG = 1, ks = 1;
};
inline bool tile_shape_ok(int BM, int BN, int KG) { return (BM == 64 && BN == 64 && KG == 1) || (BM == 64 && BN == 32 && KG == 2) || (BM == 32 && BN == 32 && KG == 4) || (BM == 32 && BN == 128 && KG == 1); }
// `cus` = compute units of the device (hipDeviceProp: multiProcessorCount; advisor, round 5): the tile-count thresholds below were
// measured on the 256-CU MI355X and scale with the chip (a partitioned device, another SKU); at 256 they are the measured literals.
inline TileChoice choose_tile(int M, int Nreal, int ntaps, int cpt, int K, int nphase, bool bf16, const std::string& name, const char* force,
                              const char* plan, bool allow_deconv96 = false, int cus = 256)
{
    TileChoice c;
    const long long half = cus / 2, full = cus, t200 = (long long)cus * 200 / 256, t72 = (long long)cus * 72 / 256, t64 = cus / 4;
    const int nch = ntaps * cpt;
    const int kel = K;  // K-elements (a chunk is 32 of them in fp32, 64 in bf16)
    const long long mt = (M + 63) / 64, nreal = Nreal;
    const long long tiles = mt * (round_up((int)nreal, 64) / 64) * nphase;
    // (tiles <= 72 && K >= 512: a single scale's 1x1 layers at 46x46 -- a pyramid rank's plan: `res3*_branch2a` 10.1 -> 7.0 us as 64x32x2)
    if ((tiles <= half && kel >= 768) || (tiles <= t200 && kel >= 4096) || (tiles <= t72 && kel >= 512)) {
        // Too few 64x64 tiles for 256 CUs and a long K: split K.  Inside the workgroup where that alone fills the chip
        // (one workgroup per CU, four K-parallel or M/N-parallel accumulators: no slabs, no reduce launch) ...
        const long long t64x32 = mt * (round_up((int)nreal, 32) / 32) * nphase;
        const long long t32x32 = ((M + 31) / 32) * (round_up((int)nreal, 32) / 32) * nphase;
        if (t64x32 > half && t64x32 <= full && cpt % 2 == 0) c.BM = 64, c.BN = 32, c.KG = 2;
        else if (t32x32 > half && t32x32 <= full && cpt % 4 == 0) c.BM = 32, c.BN = 32, c.KG = 4;
        // ... else across workgroups: 5 partial slabs + splitk_reduce_kernel (bf16 loops are 2-3x shorter: there the extra
        // launch only pays for the smallest, deepest layers)
        else if (!bf16 || (tiles <= t64 && kel >= 2048)) c.ks = std::min(5, nch);
    }
    // The transposed conv in fp32 (round 4): 300 tiles of 64 x 64 on 256 CUs are TWO rounds for 44 of them; 64 x 96 tiles with two K groups
    // and three accumulators per wave (conv.hip: NACC) are 200 tiles, ONE round of 1.5 block-K-loops per SIMD.  Only this layer takes the
    // shape (no shortcut, one output tensor), only while its 64 x 64 plan needs a second round and the 96-wide one does not.
    const bool deconv96_ok = !bf16 && nphase == 4 && cpt % 2 == 0 && round_up((int)nreal, 96) == round_up((int)nreal, 64);
    if (deconv96_ok && allow_deconv96 && tiles > full && mt * (round_up((int)nreal, 96) / 96) * nphase <= full) c.BM = 64, c.BN = 96, c.KG = 2, c.ks = 1;
    auto take = [&](const char* spec) {
        int fBM = 0, fBN = 0, fKG = 0, fks = 0;
        if (sscanf(spec, "%d,%d,%d,%d", &fBM, &fBN, &fKG, &fks) != 4 || fks < 1 || fks > 8 || cpt % std::max(fKG, 1) != 0) return;
        if (fBM == 64 && fBN == 96 && fKG == 2) {  // (the three-accumulator shape: where it is built for, never with K slabs)
            // an override can ask for the shape where the default would not pick it (other tile counts), but never where the caller
            // says the layer or the handle does not admit it -- a split-product handle, VNECT_NO_DECONV96, a build whose 96-wide
            // instantiation did not fit the register file (conv.hip: conv_deconv96_available): `allow_deconv96` covers all three
            if (deconv96_ok && allow_deconv96) c.BM = 64, c.BN = 96, c.KG = 2, c.ks = 1;
            return;
        }
        if (tile_shape_ok(fBM, fBN, fKG)) c.BM = fBM, c.BN = fBN, c.KG = fKG, c.ks = std::max(1, std::min(fks, nch / fKG));
    };
    if (force) take(force);
    if (plan) {
        const std::string key = name + "=";
        const char* p = strstr(plan, key.c_str());
        if (p && (p == plan || p[-1] == ';')) take(p + key.size());
    }
    return c;
}

// ---- head split of a paired 1x1 launch (rt_plan.cpp: add_conv_pair) -------------------------------------
// A launch of T 64x64 tiles takes ceil(T / 256) block K loops per SIMD (a CU runs its tiles two at a time on the same four SIMDs); the
// same columns as 64x32 tiles with two K groups take half a loop per round.  Returns how many leading channels of layer a to run as a
// launch of their own (a multiple of 64 that choose_tile gives the 64x32x2 shape and ONE round), 0 if that does not beat the single
// launch by more than it costs: a block K loop is K / 2 matrix instructions of 64 cycles at ~2.15 GHz, a dependent launch ~4 us
// (floor + cold start).  fp32 only: a bf16 loop is a quarter of that and a launch is not.  `cus` = the device's compute
// units (a round is one tile per CU); the clock and the launch cost are MI355X's (the decision is +0.6 % there: a wrong guess on another
// part costs a fraction of a per cent, never correctness -- tools/switch_matrix.sh covers both plans).
inline int pair_head_cols(int M, int cout_a, int cout_b, int K, bool bf16, int cus = 256)
{
    if (bf16 || K < 768 || K % 64 || cus < 8) return 0;
    const long long mt = (M + 63) / 64, nt = (cout_a + cout_b) / 64;
    const double loop_us = K * 0.5 * 64 / 2150.0, launch_us = 4.0;
    auto rounds = [cus](long long t) { return (double)((t + cus - 1) / cus); };
    double best = rounds(mt * nt) * loop_us - 1.0;
    int head = 0;
    for (int c = 64; c < cout_a; c += 64) {
        const long long th = mt * (c / 32);
        if (mt * (c / 64) > cus / 2 || th <= cus / 2 || th > cus) continue;  // (choose_tile's conditions for 64x32x2)
        const double t = rounds(mt * (nt - c / 64)) * loop_us + 0.5 * loop_us + launch_us;
        if (t < best) best = t, head = c;
    }
    return head;
}

// ---- the fused stem's row groups (stem.hip) ------------------------------------------------------------
// An image's 92 pooled rows in G groups: as MANY tiles (S x G x 4) as one round over the 256 CUs allows (S = 3: G = 21 -> 252 tiles of 4
// or 5 rows; S = 2: 32 groups of 2 or 3; S = 1: 46 of 2 -- a launch takes as long as its tiles do, so fewer images mean shorter tiles, not
// idle CUs: round 3, a pyramid rank's stem 43.6 -> 36.9 us), never more than 5 rows per tile (the LDS patch), never fewer than 2 (every
// tile recomputes one halo row), and for four or more images -- several rounds anyway -- the 4-and-5-row tiles with the least halo.
// row0[g] = first row of group g, row0[G] = 92.  Returns G.
inline int stem_groups(int S, unsigned char* row0)
{
    int G = 256 / (4 * std::max(S, 1));
    G = G < 19 ? 19 : (G > 46 ? 46 : G);
    const int base = 92 / G, rem = 92 % G;
    int r = 0;
    for (int g = 0; g < G; g++) row0[g] = (unsigned char)r, r += base + (g < rem ? 1 : 0);
    row0[G] = (unsigned char)r;
    return G;
}

// The from-the-frame form of the stem lands, per tile, the rectangle of the square its input patch is made from in LDS (stem.hip, phase
// B): at most STEM_REG_ROWS rows of STEM_REG_PITCH bytes.  True if that holds for EVERY tile of every image at these scales -- the same
// bounds arithmetic as the kernel's, on the host's copy of the tables (scales below ~0.48 need more and take the batch-tensor form).
inline bool stem_frame_fits(const ScaleTabs& st, int S, int scale_base, int groups, const unsigned char* row0, bool bf16)
{
    for (int sI = 0; sI < S; sI++) {
        const int s = sI + scale_base;
        if (s < 0 || s >= 8) return false;
        const ResizeTab& t = st.t[s];
        const bool scaled = st.scaled[s] != 0, resize = scaled && !t.copy;
        const int off = scaled ? st.pad[s] : 0, dh = scaled ? t.dh : BOX, dw = scaled ? t.dw : BOX;
        for (int g = 0; g < groups; g++)
            for (int c = 0; c < 4; c++) {
                const int h = row0[g + 1] - row0[g], nrow = 2 * h + 1, prow = 2 * nrow + (bf16 ? 6 : 5);
                const int iy0 = 4 * row0[g] - 2, ix0 = 4 * STEM_TW * c - 2;
                const int ylo = std::max(iy0, off), yhi = std::min(iy0 + prow - 1, off + dh - 1);
                const int xlo = std::max(ix0, off), xhi = std::min(ix0 + STEM_PW - 1, off + dw - 1);
                if (ylo > yhi || xlo > xhi) continue;
                const int qy0 = resize ? t.sy0[ylo - off] : ylo - off, qy1 = resize ? t.sy1[yhi - off] : yhi - off;
                const int qx0 = resize ? t.sx[xlo - off] : xlo - off, qx1 = resize ? std::min(t.sx[xhi - off] + 1, BOX - 1) : xhi - off;
                if (qy1 - qy0 + 1 > STEM_REG_ROWS || 3 * (qx1 - qx0 + 1) + 8 > STEM_REG_PITCH) return false;
            }
    }
    return true;
}

}  // namespace plan
}  // namespace vnect
