// hostplan_capi.cpp -- C shim over hostplan.h for tests/test_hostplan.py.  TEST INFRASTRUCTURE: it is NOT part of libvnect_hip.so.
// Built with plain g++ (`make -C vnect_amd/csrc hostplan`) and with -fsanitize=address,undefined (`make ... hostplan_asan`), so the
// product's host-side planning code -- which otherwise only ever runs next to a GPU -- is exercised under the sanitizers on the
// CPU box: the resize / merge / upsample tables against the oracle's own resizes, the weight packers against index formulas
// written down independently in the test, the arena against interval arithmetic, the tile choice and the stem's row groups against
// their invariants.  Each function applies the tables exactly as the device code does (pyramid.h, post.hip), on the host.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "axis.h"
#include "hostplan.h"

using namespace vnect;

extern "C" {

// cv2.resize(u8 (sh,sw,ch), fx=fy=f) through build_u8_tab + sample_u8.  Returns 0, or -1 if the table cannot be built.
int hp_resize_u8(const uint8_t* src, int sh, int sw, int ch, double f, uint8_t* dst, int* dh, int* dw)
{
    std::vector<ResizeTab> t(1);
    if (!plan::build_u8_tab(sh, sw, f, &t[0])) return -1;
    *dh = t[0].dh, *dw = t[0].dw;
    if (!dst) return 0;
    for (int y = 0; y < t[0].dh; y++)
        for (int x = 0; x < t[0].dw; x++)
            for (int c = 0; c < ch; c++)
                dst[((size_t)y * t[0].dw + x) * ch + c] =
                    (uint8_t)plan::sample_u8([&](int r, int q) { return (int)src[((size_t)r * sw + q) * ch + c]; }, t[0], y, x);
    return 0;
}

// gen_input_batch (estimator.py:70-81) the way pyramid.h computes it: squarify geometry, the square on demand, per-scale resize of
// the square, centre padding, `/255 - 0.4`.  out: (S,368,368,3) float32.  Returns 0 or -1 (message in err).
int hp_gen_input_batch(const uint8_t* frame, int H, int W, int64_t stride, const double* scales, int S, float* out, double* scaler, int* offx,
                       int* offy, char* err, int errlen)
{
    std::vector<FrameParams> fp(1);
    std::vector<ScaleTabs> st(1);
    memset(&st[0], 0, sizeof(ScaleTabs));
    auto bad = [&](const char* why) {
        snprintf(err, errlen, "%s", why);
        return -1;
    };
    if (S < 1 || S > 8) return bad("S out of range");
    if (const char* why = plan::squarify(H, W, &fp[0])) return bad(why);
    plan::fill_lut(st[0].lut);
    for (int i = 0; i < S; i++)
        if (const char* why = plan::build_scale_tab(scales[i], &st[0], i)) return bad(why);
    const FrameParams& F = fp[0];
    auto square = [&](int y, int x, int c) -> int {
        const int dy = y - F.offy, dx = x - F.offx;
        if (dy < 0 || dy >= F.sq.dh || dx < 0 || dx >= F.sq.dw) return 0;
        return plan::sample_u8([&](int r, int q) { return (int)frame[(size_t)r * stride + q * 3 + c]; }, F.sq, dy, dx);
    };
    for (int i = 0; i < S; i++)
        for (int y = 0; y < BOX; y++)
            for (int x = 0; x < BOX; x++)
                for (int c = 0; c < 3; c++) {
                    int v = 0;
                    if (!st[0].scaled[i]) v = square(y, x, c);
                    else {
                        const ResizeTab& t = st[0].t[i];
                        const int dy = y - st[0].pad[i], dx = x - st[0].pad[i];
                        if (dy >= 0 && dy < t.dh && dx >= 0 && dx < t.dw) v = plan::sample_u8([&](int r, int q) { return square(r, q, c); }, t, dy, dx);
                    }
                    out[(((size_t)i * BOX + y) * BOX + x) * 3 + c] = st[0].lut[v];
                }
    *scaler = F.scaler, *offx = F.offx, *offy = F.offy;
    return 0;
}

// the multi-scale merge (estimator.py:105-129) the way post.hip's merged_cell evaluates it: out (46,46,84) float64
int hp_merge(const float* maps, const double* scales, int S, double* out)
{
    if (S < 1 || S > 8) return -1;
    std::vector<MergeTab> mt(S);
    for (int i = 0; i < S; i++)
        if (plan::build_merge_tab(scales[i], &mt[i])) return -1;
    for (int r = 0; r < HM; r++)
        for (int c = 0; c < HM; c++)
            for (int ch = 0; ch < MAPC; ch++) {
                double acc = 0.0;
                for (int i = 0; i < S; i++) {
                    const MergeTab& m = mt[i];
                    const float* M = maps + (size_t)i * HM * HM * MAPC + ch;
                    const int sx = m.sx[c], sx1 = sx + 1 < HM ? sx + 1 : HM - 1;
                    const float p00 = M[((size_t)m.sy0[r] * HM + sx) * MAPC], p01 = M[((size_t)m.sy0[r] * HM + sx1) * MAPC];
                    const float p10 = M[((size_t)m.sy1[r] * HM + sx) * MAPC], p11 = M[((size_t)m.sy1[r] * HM + sx1) * MAPC];
                    const float r0 = m.edge[c] ? p00 : p00 * m.a0[c] + p01 * m.a1[c];
                    const float r1 = m.edge[c] ? p10 : p10 * m.a0[c] + p11 * m.a1[c];
                    const float v = m.copy ? p00 : r0 * m.b0[r] + r1 * m.b1[r];
                    acc += (double)v;
                }
                out[((size_t)r * HM + c) * MAPC + ch] = acc / (double)S;
            }
    return 0;
}

// the same merge the way post.hip evaluates it since round 3: no tables, every tap and weight from axis.h's per-entry functions and
// the MergeGeo the runtime passes to the kernels
int hp_merge_geo(const float* maps, const double* scales, int S, double* out)
{
    if (S < 1 || S > 8) return -1;
    std::vector<MergeGeo> gv(1);
    MergeGeo& g = gv[0];
    memset(&g, 0, sizeof g);
    g.S = S;
    for (int i = 0; i < S; i++)
        if (plan::build_merge_geo(scales[i], &g, i)) return -1;
    for (int r = 0; r < HM; r++)
        for (int c = 0; c < HM; c++)
            for (int ch = 0; ch < MAPC; ch++) {
                double acc = 0.0;
                for (int i = 0; i < S; i++) {
                    const AxE X = axis_x_at(c + g.off[i], HM, g.scale[i]), Y = axis_y_at(r + g.off[i], HM, g.scale[i]);
                    const float* M = maps + (size_t)i * HM * HM * MAPC + ch;
                    const float p00 = M[((size_t)Y.s0 * HM + X.s0) * MAPC], p01 = M[((size_t)Y.s0 * HM + X.s1) * MAPC];
                    const float p10 = M[((size_t)Y.s1 * HM + X.s0) * MAPC], p11 = M[((size_t)Y.s1 * HM + X.s1) * MAPC];
                    const float a1 = X.f, a0 = 1.f - a1, b1 = Y.f, b0 = 1.f - b1;
                    const float r0 = X.edge ? p00 : p00 * a0 + p01 * a1;
                    const float r1 = X.edge ? p10 : p10 * a0 + p11 * a1;
                    const float v = g.copy[i] ? p00 : r0 * b0 + r1 * b1;
                    acc += (double)v;
                }
                out[((size_t)r * HM + c) * MAPC + ch] = acc / (double)S;
            }
    return 0;
}

// axis.h's per-entry functions against the whole-table builders (hostplan.h: axis_x / axis_y), entry by entry: the number of
// differing entries (taps, edge flag, weight bits) over both axes
int hp_axis_mismatches(int ssize, int dsize, double scale)
{
    const plan::AxisTab x = plan::axis_x(ssize, dsize, scale), y = plan::axis_y(ssize, dsize, scale);
    int bad = 0;
    for (int d = 0; d < dsize; d++) {
        const AxE ex = axis_x_at(d, ssize, scale), ey = axis_y_at(d, ssize, scale);
        bad += ex.s0 != x.s0[d] || ex.s1 != x.s1[d] || ex.edge != x.edge[d] || memcmp(&ex.f, &x.f[d], 4) != 0;
        bad += ey.s0 != y.s0[d] || ey.s1 != y.s1[d] || memcmp(&ey.f, &y.f[d], 4) != 0;
    }
    return bad;
}

// utils.extract_2d_joints (utils.py:153-175) through build_up_tab: heat (46,46,nj) float64 -> joints (nj,2) [row, col]
int hp_extract_2d(const double* heat, int nj, double* joints)
{
    std::vector<UpTab> u(1);
    if (!plan::build_up_tab(&u[0])) return -1;
    const UpTab& U = u[0];
    for (int j = 0; j < nj; j++) {
        double bv = -INFINITY;
        int bi = 0;
        for (int y = 0; y < BOX; y++)
            for (int x = 0; x < BOX; x++) {
                auto hrow = [&](int sy) {
                    const double a = heat[((size_t)sy * HM + U.sx[x]) * nj + j];
                    return U.edge[x] ? a : a * U.a0[x] + heat[((size_t)sy * HM + U.sx[x] + 1) * nj + j] * U.a1[x];
                };
                const double v = hrow(U.sy0[y]) * U.b0[y] + hrow(U.sy1[y]) * U.b1[y];
                if (v > bv) bv = v, bi = y * BOX + x;
            }
        joints[2 * j] = bi / BOX, joints[2 * j + 1] = bi % BOX;
    }
    return 0;
}

int hp_squarify(int H, int W, double* scaler, int* offx, int* offy, int* dh, int* dw, int* copy, char* err, int errlen)
{
    std::vector<FrameParams> fp(1);
    if (const char* why = plan::squarify(H, W, &fp[0])) {
        snprintf(err, errlen, "%s", why);
        return -1;
    }
    *scaler = fp[0].scaler, *offx = fp[0].offx, *offy = fp[0].offy, *dh = fp[0].sq.dh, *dw = fp[0].sq.dw, *copy = fp[0].sq.copy;
    return 0;
}

// ---- weight packing ----
void hp_pack_conv(const float* W, int k, int cin, int cout, int cp, int conv1, int bf16, int Npad, int K, int n0, float* wp)
{
    std::vector<float> v((size_t)Npad * K, 0.f);
    plan::pack_conv(W, k, cin, cout, cp, conv1 != 0, bf16 != 0, K, n0, v);
    memcpy(wp, v.data(), v.size() * sizeof(float));
}
void hp_pack_tail(const float* Wc, int mid, int cout, int bf16, float* w2)
{
    std::vector<float> v;
    plan::pack_tail(Wc, mid, cout, bf16 != 0, v);
    memcpy(w2, v.data(), v.size() * sizeof(float));
}
void hp_pack_deconv(const float* W1, const float* W2, int Npad, int K, float* wp, int* dy16, int* dx16)
{
    std::vector<float> v;
    plan::pack_deconv(W1, W2, Npad, K, v, dy16, dx16);
    memcpy(wp, v.data(), v.size() * sizeof(float));
}
void hp_fold_bn(const float* g, const float* b, const float* m, const float* v, int C, int Npad, float* bias, float* scale, float* shift)
{
    std::vector<float> bb, sc, sh;
    plan::fold_bn(g, b, m, v, C, Npad, bb, sc, sh);
    memcpy(bias, bb.data(), Npad * sizeof(float)), memcpy(scale, sc.data(), Npad * sizeof(float)), memcpy(shift, sh.data(), Npad * sizeof(float));
}
void hp_pack_split3(const float* wp, int Npad, int K, uint16_t* out)
{
    std::vector<float> v(wp, wp + (size_t)Npad * K);
    std::vector<uint16_t> o;
    plan::pack_split3(v, 1, Npad, K, o);
    memcpy(out, o.data(), o.size() * 2);
}
uint16_t hp_to_bf16(float f) { return plan::to_bf16(f); }
float hp_from_bf16(uint16_t b) { return plan::from_bf16(b); }

// ---- arena, tiles, stem ----
uint64_t hp_arena(const int* first, const int* last, const uint64_t* need, int n, uint64_t* off)
{
    std::vector<int> f(first, first + n), l(last, last + n);
    std::vector<size_t> nd(need, need + n), o;
    const size_t total = plan::arena_first_fit(f, l, nd, o);
    for (int i = 0; i < n; i++) off[i] = o[i];
    return total;
}
void hp_choose_tile(int M, int Nreal, int ntaps, int cpt, int K, int nphase, int bf16, const char* name, const char* force, const char* plan_s,
                    int* out4)
{
    const plan::TileChoice c = plan::choose_tile(M, Nreal, ntaps, cpt, K, nphase, bf16 != 0, name ? name : "", force, plan_s);
    out4[0] = c.BM, out4[1] = c.BN, out4[2] = c.KG, out4[3] = c.ks;
}
// the same with the transposed conv's three-accumulator shape allowed (what rt_plan.cpp passes for the fp32 instruction path)
void hp_choose_tile96(int M, int Nreal, int ntaps, int cpt, int K, int nphase, int bf16, const char* name, const char* force, const char* plan_s,
                      int* out4)
{
    const plan::TileChoice c = plan::choose_tile(M, Nreal, ntaps, cpt, K, nphase, bf16 != 0, name ? name : "", force, plan_s, true);
    out4[0] = c.BM, out4[1] = c.BN, out4[2] = c.KG, out4[3] = c.ks;
}
int hp_pair_head_cols(int M, int cout_a, int cout_b, int K, int bf16) { return plan::pair_head_cols(M, cout_a, cout_b, K, bf16 != 0); }
int hp_stem_groups(int S, uint8_t* row0) { return plan::stem_groups(S, row0); }
int hp_stem_frame_fits(const double* scales, int S, int scale_base, int bf16)
{
    std::vector<ScaleTabs> st(1);
    memset(&st[0], 0, sizeof(ScaleTabs));
    for (int i = 0; i < S + scale_base && i < 8; i++)
        if (plan::build_scale_tab(scales[i], &st[0], i)) return -1;
    uint8_t row0[STEM_MAXGROUPS + 1];
    const int G = plan::stem_groups(S, row0);
    return plan::stem_frame_fits(st[0], S, scale_base, G, row0, bf16 != 0) ? 1 : 0;
}

}  // extern "C"
