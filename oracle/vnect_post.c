/*
 * vnect_post.c -- CPU oracle, pre/post-processing half.  TEST INFRASTRUCTURE (see vnect_oracle.h).
 * Compiled with -ffp-contract=off: every multiply and add below rounds separately, like the
 * numpy / CPython / OpenCV-generic arithmetic it restates.
 */
#include "vnect_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BOX 368
#define HM 46
#define NJ 21

/* cvRound: round half to even (OpenCV saturate_cast<int>(double) == lrint in default mode) */
int vo_cvround(double v) { return (int)nearbyint(v); }

void vo_resize_size(int sh, int sw, double f, int* dh, int* dw)
{
    *dw = vo_cvround(sw * f);
    *dh = vo_cvround(sh * f);
}

/* OpenCV resize.cpp, INTER_LINEAR tables.  scale = 1/f (the user fx is kept, not recomputed
 * from dsize, when dsize=(0,0)).  x: offsets clamped and fraction zeroed at both borders;
 * y: floor + fraction kept, the two rows are clipped when used. */
static void tab_x(int ssize, int dsize, double scale, int* ofs, float* frac, int* xmax_out)
{
    int xmax = dsize;
    for (int d = 0; d < dsize; d++) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) fx = 0, sx = 0;
        if (sx + 1 >= ssize) {
            if (xmax > d) xmax = d;
            if (sx >= ssize - 1) fx = 0, sx = ssize - 1;
        }
        ofs[d] = sx;
        frac[d] = fx;
    }
    *xmax_out = xmax;
}

static void tab_y(int dsize, double scale, int* ofs, float* frac)
{
    for (int d = 0; d < dsize; d++) {
        float fy = (float)((d + 0.5) * scale - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        ofs[d] = sy;
        frac[d] = fy;
    }
}

static int clipi(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

static short sat_short_round(float v)
{
    int r = (int)nearbyintf(v);
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

/* 8-bit path: HResizeLinear<uchar,int,short,2048> + VResizeLinear<uchar,int,short,FixedPtCast<..,22>> */
void vo_resize_u8(const uint8_t* src, int sh, int sw, int cn, double f, uint8_t* dst)
{
    int dh, dw;
    vo_resize_size(sh, sw, f, &dh, &dw);
    if (dh == sh && dw == sw) {
        memcpy(dst, src, (size_t)sh * sw * cn);
        return;
    }
    double scale = 1.0 / f;
    int* xo = (int*)malloc(sizeof(int) * dw);
    float* xf = (float*)malloc(sizeof(float) * dw);
    int* yo = (int*)malloc(sizeof(int) * dh);
    float* yf = (float*)malloc(sizeof(float) * dh);
    int xmax;
    tab_x(sw, dw, scale, xo, xf, &xmax);
    tab_y(dh, scale, yo, yf);
    int* r0 = (int*)malloc(sizeof(int) * dw * cn);
    int* r1 = (int*)malloc(sizeof(int) * dw * cn);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = clipi(yo[dy], 0, sh), sy1 = clipi(yo[dy] + 1, 0, sh);
        short b0 = sat_short_round((1.f - yf[dy]) * 2048.f), b1 = sat_short_round(yf[dy] * 2048.f);
        for (int k = 0; k < 2; k++) {
            const uint8_t* S = src + (size_t)(k ? sy1 : sy0) * sw * cn;
            int* D = k ? r1 : r0;
            for (int dx = 0; dx < dw; dx++) {
                short a0 = sat_short_round((1.f - xf[dx]) * 2048.f), a1 = sat_short_round(xf[dx] * 2048.f);
                for (int c = 0; c < cn; c++) {
                    int sx = xo[dx] * cn + c;
                    D[dx * cn + c] = dx < xmax ? S[sx] * a0 + S[sx + cn] * a1 : S[sx] * 2048;
                }
            }
        }
        uint8_t* o = dst + (size_t)dy * dw * cn;
        for (int x = 0; x < dw * cn; x++)
            o[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xo), free(xf), free(yo), free(yf), free(r0), free(r1);
}

/* float paths: HResizeLinear<T,WT,float,1> then VResizeLinear: dst = S0*b0 + S1*b1 in WT */
#define DEF_RESIZE_F(NAME, T)                                                                 \
    void NAME(const T* src, int sh, int sw, int cn, double f, T* dst)                         \
    {                                                                                         \
        int dh, dw;                                                                           \
        vo_resize_size(sh, sw, f, &dh, &dw);                                                  \
        if (dh == sh && dw == sw) {                                                           \
            memcpy(dst, src, (size_t)sh * sw * cn * sizeof(T));                               \
            return;                                                                           \
        }                                                                                     \
        double scale = 1.0 / f;                                                               \
        int* xo = (int*)malloc(sizeof(int) * dw);                                             \
        float* xf = (float*)malloc(sizeof(float) * dw);                                       \
        int* yo = (int*)malloc(sizeof(int) * dh);                                             \
        float* yf = (float*)malloc(sizeof(float) * dh);                                       \
        int xmax;                                                                             \
        tab_x(sw, dw, scale, xo, xf, &xmax);                                                  \
        tab_y(dh, scale, yo, yf);                                                             \
        T* r0 = (T*)malloc(sizeof(T) * dw * cn);                                              \
        T* r1 = (T*)malloc(sizeof(T) * dw * cn);                                              \
        for (int dy = 0; dy < dh; dy++) {                                                     \
            int sy0 = clipi(yo[dy], 0, sh), sy1 = clipi(yo[dy] + 1, 0, sh);                   \
            float b0 = 1.f - yf[dy], b1 = yf[dy];                                             \
            for (int k = 0; k < 2; k++) {                                                     \
                const T* S = src + (size_t)(k ? sy1 : sy0) * sw * cn;                         \
                T* D = k ? r1 : r0;                                                           \
                for (int dx = 0; dx < dw; dx++) {                                             \
                    float a0 = 1.f - xf[dx], a1 = xf[dx];                                     \
                    for (int c = 0; c < cn; c++) {                                            \
                        int sx = xo[dx] * cn + c;                                             \
                        D[dx * cn + c] = dx < xmax ? S[sx] * a0 + S[sx + cn] * a1 : S[sx];    \
                    }                                                                         \
                }                                                                             \
            }                                                                                 \
            T* o = dst + (size_t)dy * dw * cn;                                                \
            for (int x = 0; x < dw * cn; x++) o[x] = r0[x] * b0 + r1[x] * b1;                 \
        }                                                                                     \
        free(xo), free(xf), free(yo), free(yf), free(r0), free(r1);                           \
    }

DEF_RESIZE_F(vo_resize_f32, float)
DEF_RESIZE_F(vo_resize_f64, double)

/* estimator.gen_input_batch (estimator.py:70-81) */
int vo_gen_input_batch(const uint8_t* img, int H, int W, int64_t row_stride, const double* scales, int S,
                       float* batch, double* scaler_out, int* offset_x, int* offset_y)
{
    if (H <= 0 || W <= 0) return -1;
    /* utils.img_scale_squarify (utils.py:107-120) */
    double scaler = (double)BOX / (H > W ? H : W);
    int h2, w2;
    vo_resize_size(H, W, scaler, &h2, &w2);
    if ((H > W ? h2 : w2) != BOX || h2 > BOX || w2 > BOX) return -1; /* numpy would raise on the slice assign */
    uint8_t* tight = (uint8_t*)malloc((size_t)H * W * 3);
    for (int y = 0; y < H; y++) memcpy(tight + (size_t)y * W * 3, img + (size_t)y * row_stride, (size_t)W * 3);
    uint8_t* scaled = (uint8_t*)malloc((size_t)h2 * w2 * 3);
    vo_resize_u8(tight, H, W, 3, scaler, scaled);
    free(tight);
    /* utils.img_padding (utils.py:82-104), black; uses the SCALED h,w */
    uint8_t* sq = (uint8_t*)calloc((size_t)BOX * BOX * 3, 1);
    int offx = 0, offy = 0;
    if (h2 > w2) {
        offx = BOX / 2 - w2 / 2;
        for (int y = 0; y < BOX; y++) memcpy(sq + ((size_t)y * BOX + offx) * 3, scaled + (size_t)y * w2 * 3, (size_t)w2 * 3);
    } else {
        offy = BOX / 2 - h2 / 2;
        for (int y = 0; y < h2; y++) memcpy(sq + (size_t)(y + offy) * BOX * 3, scaled + (size_t)y * w2 * 3, (size_t)w2 * 3);
    }
    free(scaled);
    for (int s = 0; s < S; s++) {
        float* o = batch + (size_t)s * BOX * BOX * 3;
        const uint8_t* im = sq;
        uint8_t* padded = NULL;
        if (scales[s] < 1) {
            /* utils.img_scale_padding (utils.py:123-150) */
            int d, d2;
            vo_resize_size(BOX, BOX, scales[s], &d, &d2);
            uint8_t* sm = (uint8_t*)malloc((size_t)d * d * 3);
            vo_resize_u8(sq, BOX, BOX, 3, scales[s], sm);
            int pad = (BOX - d) / 2; /* after-pad = pad + (BOX-d)%2 fills the rest */
            padded = (uint8_t*)calloc((size_t)BOX * BOX * 3, 1);
            for (int y = 0; y < d; y++) memcpy(padded + ((size_t)(y + pad) * BOX + pad) * 3, sm + (size_t)y * d * 3, (size_t)d * 3);
            free(sm);
            im = padded;
        }
        /* np.asarray(batch, float32) / 255 - 0.4 : float32 arithmetic */
        for (size_t i = 0; i < (size_t)BOX * BOX * 3; i++) o[i] = (float)im[i] / 255.f - 0.4f;
        free(padded);
    }
    free(sq);
    *scaler_out = scaler, *offset_x = offx, *offset_y = offy;
    return 0;
}

/* estimator.py:105-129 */
void vo_merge_scales(const float* maps, const double* scales, int S, double* avg)
{
    memset(avg, 0, sizeof(double) * 4 * HM * HM * NJ);
    float* one = (float*)malloc(sizeof(float) * HM * HM * NJ);
    for (int i = 0; i < S; i++) {
        double rescale = 1.0 / scales[i];
        int dh, dw;
        vo_resize_size(HM, HM, rescale, &dh, &dw);
        float* sc = (float*)malloc(sizeof(float) * dh * dw * NJ);
        int mid0 = dh / 2, mid1 = dw / 2;
        for (int q = 0; q < 4; q++) {
            /* tf.split(.., 4, axis=3): map q = channels [21q, 21q+21) of the 84 */
            const float* m = maps + (size_t)i * HM * HM * 84;
            for (int p = 0; p < HM * HM; p++)
                for (int j = 0; j < NJ; j++) one[p * NJ + j] = m[p * 84 + q * NJ + j];
            vo_resize_f32(one, HM, HM, NJ, rescale, sc);
            double* a = avg + (size_t)q * HM * HM * NJ;
            for (int r = 0; r < HM; r++)
                for (int c = 0; c < HM; c++)
                    for (int j = 0; j < NJ; j++)
                        a[(r * HM + c) * NJ + j] += (double)sc[((size_t)(mid0 - HM / 2 + r) * dw + (mid1 - HM / 2 + c)) * NJ + j];
        }
        free(sc);
    }
    free(one);
    for (int i = 0; i < 4 * HM * HM * NJ; i++) avg[i] /= (double)S;
}

/* utils.extract_2d_joints (utils.py:153-175) */
void vo_extract_2d(const double* hm_avg, double* joints_2d)
{
    double* one = (double*)malloc(sizeof(double) * HM * HM);
    double* up = (double*)malloc(sizeof(double) * BOX * BOX);
    for (int j = 0; j < NJ; j++) {
        for (int p = 0; p < HM * HM; p++) one[p] = hm_avg[p * NJ + j];
        vo_resize_f64(one, HM, HM, 1, 8.0, up);
        int best = 0;
        for (int p = 1; p < BOX * BOX; p++)
            if (up[p] > up[best]) best = p; /* np.argmax: first maximum in row-major order */
        joints_2d[j * 2 + 0] = best / BOX;
        joints_2d[j * 2 + 1] = best % BOX;
    }
    free(one), free(up);
}

/* utils.hm_pt_interp_bilinear (utils.py:58-79); map element (r,c) at map[(r*46+c)*stride] */
double vo_hm_pt_interp(const double* map, int stride, double scale, double dst_y, double dst_x)
{
    double src_x = (dst_x + 0.5) / scale - 0.5;
    double src_y = (dst_y + 0.5) / scale - 0.5;
    int x0 = (int)src_x, y0 = (int)src_y; /* Python int(): truncation toward zero */
    int x1 = x0 + 1 < HM - 1 ? x0 + 1 : HM - 1;
    int y1 = y0 + 1 < HM - 1 ? y0 + 1 : HM - 1;
#define AT(r, c) map[((size_t)(r) * HM + (c)) * stride]
    double v0 = (x1 - src_x) * AT(y0, x0) + (src_x - x0) * AT(y0, x1);
    double v1 = (x1 - src_x) * AT(y1, x0) + (src_x - x0) * AT(y1, x1);
#undef AT
    return (y1 - src_y) * v0 + (src_y - y0) * v1;
}

/* utils.extract_3d_joints (utils.py:178-219) */
void vo_extract_3d(const double* joints_2d, const double* xm, const double* ym, const double* zm, float* joints_3d)
{
    for (int j = 0; j < NJ; j++) {
        double y = joints_2d[j * 2], x = joints_2d[j * 2 + 1];
        joints_3d[j * 3 + 0] = (float)(vo_hm_pt_interp(xm + j, NJ, 8.0, y, x) * 100);
        joints_3d[j * 3 + 1] = (float)(vo_hm_pt_interp(ym + j, NJ, 8.0, y, x) * 100);
        joints_3d[j * 3 + 2] = (float)(vo_hm_pt_interp(zm + j, NJ, 8.0, y, x) * 100);
    }
    float root[3] = {joints_3d[14 * 3], joints_3d[14 * 3 + 1], joints_3d[14 * 3 + 2]};
    for (int j = 0; j < NJ; j++)
        for (int k = 0; k < 3; k++) joints_3d[j * 3 + k] -= root[k]; /* float32 in-place subtract */
}

/* OneEuroFilter.py:13-75 */
struct vo_oef {
    double freq, mincutoff, beta, dcutoff;
    int has_last;
    double lasttime;
    int x_init, dx_init;
    double x_y, x_s, dx_y, dx_s;
};

vo_oef* vo_oef_create(double freq, double mincutoff, double beta, double dcutoff)
{
    vo_oef* f = (vo_oef*)calloc(1, sizeof *f);
    f->freq = freq, f->mincutoff = mincutoff, f->beta = beta, f->dcutoff = dcutoff;
    return f;
}
void vo_oef_destroy(vo_oef* f) { free(f); }

static double oef_alpha(double freq, double cutoff)
{
    double te = 1.0 / freq;
    double tau = 1.0 / (2 * M_PI * cutoff);
    return 1.0 / (1.0 + tau / te);
}

static double lowpass(int* init, double* y, double* s, double value, double alpha)
{
    double r = *init ? alpha * value + (1.0 - alpha) * (*s) : value;
    *init = 1;
    *y = value;
    *s = r;
    return r;
}

int vo_oef_call(vo_oef* f, double x, double t, double* out)
{
    /* `if self.__lasttime and timestamp:` -- None and 0.0 are both falsy */
    if (f->has_last && f->lasttime != 0.0 && t != 0.0) {
        if (t - f->lasttime == 0.0) return -1; /* ZeroDivisionError */
        f->freq = 1.0 / (t - f->lasttime);
    }
    f->lasttime = t;
    f->has_last = 1;
    double dx = f->x_init ? (x - f->x_y) * f->freq : 0.0;
    double edx = lowpass(&f->dx_init, &f->dx_y, &f->dx_s, dx, oef_alpha(f->freq, f->dcutoff));
    double cutoff = f->mincutoff + f->beta * fabs(edx);
    *out = lowpass(&f->x_init, &f->x_y, &f->x_s, x, oef_alpha(f->freq, cutoff));
    return 0;
}

/* The 3-D filters are fed np.float32 scalars (estimator.py:91-93 indexes a float32 array), so the
 * arithmetic inside OneEuroFilter.__call__ depends on numpy's scalar promotion rules:
 *   legacy (numpy 1.x, the only numpy TF1 runs with; DEFAULT): float32 (op) Python float -> float64,
 *     float32 - float32 -> float32.  Everything is f64 except `x - prev_x`, which rounds to f32.
 *   nep50 (numpy >= 2, what the build container has): Python floats are weak, float32 (op) Python
 *     float -> float32 with the Python float rounded to f32 first.  Used only to match the fixtures
 *     tests/golden/make_golden.py records from the reference under numpy 2.2. */
static int oef_call_f32(vo_oef* f, float x, double t, int nep50, float* out)
{
    if (f->has_last && f->lasttime != 0.0 && t != 0.0) {
        if (t - f->lasttime == 0.0) return -1;
        f->freq = 1.0 / (t - f->lasttime);
    }
    f->lasttime = t;
    f->has_last = 1;
    double a_d = oef_alpha(f->freq, f->dcutoff);
    if (!nep50) {
        float diff = x - (float)f->x_y; /* np.float32 - np.float32 */
        double dx = f->x_init ? (double)diff * f->freq : 0.0;
        double edx = lowpass(&f->dx_init, &f->dx_y, &f->dx_s, dx, a_d);
        double cutoff = f->mincutoff + f->beta * fabs(edx);
        *out = (float)lowpass(&f->x_init, &f->x_y, &f->x_s, (double)x, oef_alpha(f->freq, cutoff));
        return 0;
    }
    /* nep50: states hold f32 values (stored in the double fields) */
    float edx;
    if (!f->x_init) { /* dx = 0.0 (Python float); LowPass first call returns it */
        f->dx_init = 1, f->dx_y = 0.0, f->dx_s = 0.0;
        edx = 0.f;
    } else {
        float dx = (x - (float)f->x_y) * (float)f->freq;
        /* alpha*value + (1.0-alpha)*s ; on the 2nd call s is the Python float 0.0 -> product is 0.0 */
        float s = (float)a_d * dx + (float)(1.0 - a_d) * (float)f->dx_s;
        f->dx_y = dx, f->dx_s = s;
        edx = s;
    }
    double cutoff = f->mincutoff + f->beta * fabs((double)edx);
    double a_x = oef_alpha(f->freq, cutoff);
    float r = f->x_init ? (float)a_x * x + (float)(1.0 - a_x) * (float)f->x_s : x;
    f->x_init = 1, f->x_y = x, f->x_s = r;
    *out = r;
    return 0;
}

/* estimator (estimator.py:27-68, 83-142) */
struct vo_estimator {
    vo_net* net;
    int S;
    int nep50;
    double scales[8];
    vo_oef f2[NJ][2], f3[NJ][3];
};

void vo_est_set_nep50(vo_estimator* e, int on) { e->nep50 = on; }

void vo_est_reset(vo_estimator* e)
{
    vo_oef a = {30, 1.7, 0.3, 0.4, 0, 0, 0, 0, 0, 0, 0, 0}; /* filter_config_2d :34-39 */
    vo_oef b = {30, 0.8, 0.4, 0.4, 0, 0, 0, 0, 0, 0, 0, 0}; /* filter_config_3d :40-45 */
    for (int j = 0; j < NJ; j++) {
        e->f2[j][0] = e->f2[j][1] = a;
        e->f3[j][0] = e->f3[j][1] = e->f3[j][2] = b;
    }
}

vo_estimator* vo_est_create(vo_net* net, const double* scales, int S)
{
    if (S < 1 || S > 8) return NULL;
    vo_estimator* e = (vo_estimator*)calloc(1, sizeof *e);
    e->net = net, e->S = S;
    memcpy(e->scales, scales, sizeof(double) * S);
    vo_est_reset(e);
    return e;
}
void vo_est_destroy(vo_estimator* e) { free(e); }

int vo_est_postprocess(vo_estimator* e, const float* maps, double t2d, double t3d, double scaler, int offx, int offy,
                       double* j2, float* j3)
{
    double* avg = (double*)malloc(sizeof(double) * 4 * HM * HM * NJ);
    vo_merge_scales(maps, e->scales, e->S, avg);
    int rc = 0;
    vo_extract_2d(avg, j2);
    for (int j = 0; j < NJ && !rc; j++) /* joint_filter dim=2 :85-88 */
        for (int k = 0; k < 2 && !rc; k++) rc = vo_oef_call(&e->f2[j][k], j2[j * 2 + k], t2d, &j2[j * 2 + k]);
    if (!rc) {
        size_t P = (size_t)HM * HM * NJ;
        vo_extract_3d(j2, avg + P, avg + 2 * P, avg + 3 * P, j3);
        for (int j = 0; j < NJ && !rc; j++) /* joint_filter dim=3 :89-93, stored back into the f32 array */
            for (int k = 0; k < 3 && !rc; k++) rc = oef_call_f32(&e->f3[j][k], j3[j * 3 + k], t3d, e->nep50, &j3[j * 3 + k]);
    }
    if (!rc)
        for (int j = 0; j < NJ; j++) { /* :138-139 */
            j2[j * 2 + 0] = (j2[j * 2 + 0] - offy) / scaler;
            j2[j * 2 + 1] = (j2[j * 2 + 1] - offx) / scaler;
        }
    free(avg);
    return rc;
}

int vo_est_infer(vo_estimator* e, const uint8_t* img, int H, int W, int64_t row_stride, double t2d, double t3d,
                 double* j2, float* j3)
{
    float* batch = (float*)malloc(sizeof(float) * (size_t)e->S * BOX * BOX * 3);
    float* maps = (float*)malloc(sizeof(float) * (size_t)e->S * HM * HM * 84);
    double scaler;
    int offx, offy;
    int rc = vo_gen_input_batch(img, H, W, row_stride, e->scales, e->S, batch, &scaler, &offx, &offy);
    if (!rc) rc = vo_net_forward(e->net, batch, e->S, maps);
    if (!rc) rc = vo_est_postprocess(e, maps, t2d, t3d, scaler, offx, offy, j2, j3);
    free(batch), free(maps);
    return rc;
}
