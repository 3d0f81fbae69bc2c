/*
 * vnect_net.c -- CPU oracle, network half.  TEST INFRASTRUCTURE (see vnect_oracle.h).
 *
 * fp32 restatement of the TF1 graph built by /root/reference/src/vnect_model.py:25-217,
 * NHWC, one im2col-free blocked SGEMM (OpenMP over row blocks; K is never split across
 * threads, so results do not depend on the thread count).
 */
#include "vnect_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* SGEMM: C[M][N] = A[M][K] * B[K][N], A supplied through a packing callback              */
/* ------------------------------------------------------------------------------------ */
#define MR 6
#define KC 256
#define MC 96 /* multiple of MR */

typedef float v8f __attribute__((vector_size(32)));
typedef float v8f_u __attribute__((vector_size(32), aligned(4)));
typedef float v16f __attribute__((vector_size(64)));
typedef float v16f_u __attribute__((vector_size(64), aligned(4)));

/* micro-kernel: tmp[MR][NR] = sum_k a[k][MR] (x) b[k][NR]; every output element is one
 * sequential fma chain over k, so the AVX2 and AVX-512 variants give identical bits. */
#define DEF_UKR(NAME, VT, VTU, VW, TGT)                                                        \
    __attribute__((target(TGT))) static void NAME(int kc, const float* a, const float* b,     \
                                                  float* c, int ldc, int mr, int nr, int acc) \
    {                                                                                          \
        VT c0[MR], c1[MR];                                                                     \
        for (int i = 0; i < MR; i++) {                                                         \
            c0[i] = (VT){0};                                                                   \
            c1[i] = (VT){0};                                                                   \
        }                                                                                      \
        for (int k = 0; k < kc; k++) {                                                         \
            VT b0 = *(const VTU*)(b + (size_t)k * 2 * VW);                                     \
            VT b1 = *(const VTU*)(b + (size_t)k * 2 * VW + VW);                                \
            const float* ak = a + (size_t)k * MR;                                              \
            for (int i = 0; i < MR; i++) {                                                     \
                VT av = (VT){0} + ak[i];                                                       \
                c0[i] += av * b0;                                                              \
                c1[i] += av * b1;                                                              \
            }                                                                                  \
        }                                                                                      \
        float tmp[MR][2 * VW];                                                                 \
        for (int i = 0; i < MR; i++) {                                                         \
            *(VTU*)&tmp[i][0] = c0[i];                                                         \
            *(VTU*)&tmp[i][VW] = c1[i];                                                        \
        }                                                                                      \
        for (int i = 0; i < mr; i++)                                                           \
            for (int j = 0; j < nr; j++)                                                       \
                c[(size_t)i * ldc + j] = acc ? c[(size_t)i * ldc + j] + tmp[i][j] : tmp[i][j]; \
    }

DEF_UKR(ukr_avx2, v8f, v8f_u, 8, "avx2,fma")
DEF_UKR(ukr_avx512, v16f, v16f_u, 16, "avx512f,fma")

typedef void (*ukr_fn)(int, const float*, const float*, float*, int, int, int, int);

/* row-gather description for the implicit-GEMM A operand */
typedef struct {
    const float* in; /* (S,H,W,C) */
    int S, H, W, C;
    int Ho, Wo;   /* logical output grid the M index runs over: m = (s*Ho + oy)*Wo + ox */
    int stride;   /* input step per output step */
    int ntaps;    /* K = ntaps * C, k = tap*C + ci */
    int dy[49], dx[49]; /* input pixel = (oy*stride + dy[t], ox*stride + dx[t]); outside -> 0 */
} gather_t;

static void pack_a(const gather_t* g, int m0, int mc, int pc, int kc, int M, float* Ap)
{
    int C = g->C;
    for (int ip = 0; ip * MR < mc; ip++) {
        float* P = Ap + (size_t)ip * kc * MR;
        for (int i = 0; i < MR; i++) {
            int m = m0 + ip * MR + i;
            if (ip * MR + i >= mc || m >= M) {
                for (int k = 0; k < kc; k++) P[(size_t)k * MR + i] = 0.f;
                continue;
            }
            int ox = m % g->Wo, oy = (m / g->Wo) % g->Ho, s = m / (g->Wo * g->Ho);
            int k = pc;
            while (k < pc + kc) {
                int t = k / C, ci = k % C;
                int run = C - ci;
                if (run > pc + kc - k) run = pc + kc - k;
                int iy = oy * g->stride + g->dy[t], ix = ox * g->stride + g->dx[t];
                float* d = P + (size_t)(k - pc) * MR + i;
                if (iy >= 0 && iy < g->H && ix >= 0 && ix < g->W) {
                    const float* src = g->in + (((size_t)s * g->H + iy) * g->W + ix) * C + ci;
                    for (int r = 0; r < run; r++) d[(size_t)r * MR] = src[r];
                } else {
                    for (int r = 0; r < run; r++) d[(size_t)r * MR] = 0.f;
                }
                k += run;
            }
        }
    }
}

static int g_use512 = -1;

static void gemm_gather(const gather_t* g, int M, int N, int K, const float* B, int ldb, float* Cm, int ldc)
{
    if (g_use512 < 0) {
        __builtin_cpu_init();
        const char* e = getenv("VO_NO_AVX512");
        g_use512 = (__builtin_cpu_supports("avx512f") && !(e && e[0] == '1')) ? 1 : 0;
    }
    const int NR = g_use512 ? 32 : 16;
    ukr_fn ukr = g_use512 ? ukr_avx512 : ukr_avx2;
    int npan = (N + NR - 1) / NR;
    float* Bp = (float*)aligned_alloc(64, (size_t)npan * NR * KC * sizeof(float));
    int nth = omp_get_max_threads();
    float* Apool = (float*)aligned_alloc(64, (size_t)nth * MC * KC * sizeof(float));
    for (int pc = 0; pc < K; pc += KC) {
        int kc = K - pc < KC ? K - pc : KC;
#pragma omp parallel
        {
#pragma omp for schedule(static)
            for (int jp = 0; jp < npan; jp++) {
                float* P = Bp + (size_t)jp * kc * NR;
                int n0 = jp * NR;
                for (int k = 0; k < kc; k++) {
                    const float* src = B + (size_t)(pc + k) * ldb + n0;
                    for (int j = 0; j < NR; j++) P[(size_t)k * NR + j] = (n0 + j < N) ? src[j] : 0.f;
                }
            }
            float* Ap = Apool + (size_t)omp_get_thread_num() * MC * KC;
            /* 2-D decomposition (row block x group of column panels) so that small-M layers still give every
             * thread work; K is never split, so results do not depend on the thread count. */
            const int NG = 4; /* column panels per task */
            const int ngrp = (npan + NG - 1) / NG, nblk = (M + MC - 1) / MC;
#pragma omp for schedule(dynamic, 1)
            for (int task = 0; task < nblk * ngrp; task++) {
                const int ic = (task / ngrp) * MC, jg = task % ngrp;
                int mc = M - ic < MC ? M - ic : MC;
                pack_a(g, ic, mc, pc, kc, M, Ap);
                for (int jp = jg * NG; jp < npan && jp < (jg + 1) * NG; jp++) {
                    int nr = N - jp * NR < NR ? N - jp * NR : NR;
                    for (int ip = 0; ip * MR < mc; ip++) {
                        int mr = mc - ip * MR < MR ? mc - ip * MR : MR;
                        ukr(kc, Ap + (size_t)ip * kc * MR, Bp + (size_t)jp * kc * NR,
                            Cm + (size_t)(ic + ip * MR) * ldc + jp * NR, ldc, mr, nr, pc > 0);
                    }
                }
            }
        }
    }
    free(Bp);
    free(Apool);
}

void vo_sgemm(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc)
{
    /* expressed as a 1x1 "conv" over an (1, M, 1, lda) image whose first K channels are used */
    if (lda == K) {
        gather_t g = {A, 1, M, 1, K, M, 1, 1, 1, {0}, {0}};
        gemm_gather(&g, M, N, K, B, ldb, C, ldc);
    } else {
        float* tmp = (float*)malloc((size_t)M * K * sizeof(float));
        for (int m = 0; m < M; m++) memcpy(tmp + (size_t)m * K, A + (size_t)m * lda, K * sizeof(float));
        gather_t g = {tmp, 1, M, 1, K, M, 1, 1, 1, {0}, {0}};
        gemm_gather(&g, M, N, K, B, ldb, C, ldc);
        free(tmp);
    }
}

int vo_sgemm_threads(void) { return omp_get_max_threads(); }
void vo_set_threads(int n) { if (n >= 1) omp_set_num_threads(n); }

/* ------------------------------------------------------------------------------------ */
/* network                                                                              */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    char name[64];
    float* data;
    int shape[4];
    int ndim;
    size_t count;
} vo_array;

#define MAXW 128
#define MAXA 96

struct vo_net {
    vo_array w[MAXW];
    int nw;
    vo_array act[MAXA];
    int nact;
    int keep, paper_res2c;
    char err[256];
};

vo_net* vo_net_create(void) { return (vo_net*)calloc(1, sizeof(vo_net)); }

static void clear_acts(vo_net* n)
{
    for (int i = 0; i < n->nact; i++) free(n->act[i].data);
    n->nact = 0;
}

void vo_net_destroy(vo_net* n)
{
    if (!n) return;
    clear_acts(n);
    for (int i = 0; i < n->nw; i++) free(n->w[i].data);
    free(n);
}

const char* vo_net_error(vo_net* n) { return n->err; }

void vo_net_options(vo_net* n, int keep, int paper_res2c)
{
    n->keep = keep;
    n->paper_res2c = paper_res2c;
}

int vo_net_set_weight(vo_net* n, const char* name, const float* data, const int64_t* shape, int ndim)
{
    if (ndim < 1 || ndim > 4 || strlen(name) >= 64) {
        snprintf(n->err, sizeof n->err, "bad weight %s", name);
        return -1;
    }
    vo_array* a = NULL;
    for (int i = 0; i < n->nw; i++)
        if (!strcmp(n->w[i].name, name)) a = &n->w[i];
    if (!a) {
        if (n->nw == MAXW) {
            snprintf(n->err, sizeof n->err, "too many weights");
            return -1;
        }
        a = &n->w[n->nw++];
        memset(a, 0, sizeof *a);
        strcpy(a->name, name);
    }
    free(a->data);
    a->ndim = ndim;
    a->count = 1;
    for (int i = 0; i < ndim; i++) {
        a->shape[i] = (int)shape[i];
        a->count *= (size_t)shape[i];
    }
    a->data = (float*)malloc(a->count * sizeof(float));
    memcpy(a->data, data, a->count * sizeof(float));
    return 0;
}

static const vo_array* find_w(vo_net* n, const char* scope, const char* leaf)
{
    char nm[96];
    snprintf(nm, sizeof nm, "%s/%s", scope, leaf);
    for (int i = 0; i < n->nw; i++)
        if (!strcmp(n->w[i].name, nm)) return &n->w[i];
    snprintf(n->err, sizeof n->err, "missing weight %s", nm);
    return NULL;
}

typedef struct {
    float* d;
    int n, h, w, c;
} tens;

static tens talloc(int n, int h, int w, int c)
{
    tens t = {(float*)malloc((size_t)n * h * w * c * sizeof(float)), n, h, w, c};
    return t;
}

/* register a layer output: kept (ownership moves to the net) or remembered for freeing */
static void reg(vo_net* net, const char* name, tens t)
{
    vo_array* a = &net->act[net->nact++];
    memset(a, 0, sizeof *a);
    snprintf(a->name, sizeof a->name, "%s", name);
    a->data = t.d;
    a->shape[0] = t.n, a->shape[1] = t.h, a->shape[2] = t.w, a->shape[3] = t.c;
    a->ndim = 4;
    a->count = (size_t)t.n * t.h * t.w * t.c;
}

const float* vo_net_activation(vo_net* n, const char* name, int* shape)
{
    for (int i = 0; i < n->nact; i++)
        if (!strcmp(n->act[i].name, name)) {
            for (int k = 0; k < 4; k++) shape[k] = n->act[i].shape[k];
            return n->act[i].data;
        }
    return NULL;
}

/* TF 'SAME': out = ceil(in/stride), pad_total = max((out-1)*stride + k - in, 0), before = total/2 */
static void same_pad(int in, int k, int stride, int* out, int* before)
{
    *out = (in + stride - 1) / stride;
    int tot = (*out - 1) * stride + k - in;
    if (tot < 0) tot = 0;
    *before = tot / 2;
}

/* tc.layers.conv2d (vnect_model.py): conv + BiasAdd (+ residual add) (+ ReLU) */
static int conv(vo_net* net, const char* scope, tens x, int k, int stride, int cout, int relu, const tens* resid,
                tens* y)
{
    /* with a shortcut the stored tensor is the block output relu(branch2c + shortcut), named resNx */
    char regname[64];
    snprintf(regname, sizeof regname, "%s", scope);
    if (resid) *strchr(regname, '_') = 0;
    const vo_array* W = find_w(net, scope, "weights");
    const vo_array* Bv = W ? find_w(net, scope, "biases") : NULL;
    if (!W || !Bv) return -1;
    if (W->shape[0] != k || W->shape[2] != x.c || W->shape[3] != cout || (int)Bv->count != cout) {
        snprintf(net->err, sizeof net->err, "shape mismatch at %s", scope);
        return -1;
    }
    int ho, wo, pt, pl;
    if (k == 1) { /* VALID */
        ho = (x.h - 1) / stride + 1, wo = (x.w - 1) / stride + 1, pt = pl = 0;
    } else {
        same_pad(x.h, k, stride, &ho, &pt);
        same_pad(x.w, k, stride, &wo, &pl);
    }
    gather_t g = {x.d, x.n, x.h, x.w, x.c, ho, wo, stride, k * k, {0}, {0}};
    for (int ky = 0; ky < k; ky++)
        for (int kx = 0; kx < k; kx++) g.dy[ky * k + kx] = ky - pt, g.dx[ky * k + kx] = kx - pl;
    *y = talloc(x.n, ho, wo, cout);
    int M = x.n * ho * wo;
    gemm_gather(&g, M, cout, k * k * x.c, W->data, cout, y->d, cout);
    const float* b = Bv->data;
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; m++) {
        float* r = y->d + (size_t)m * cout;
        const float* q = resid ? resid->d + (size_t)m * cout : NULL;
        for (int c = 0; c < cout; c++) {
            float v = r[c] + b[c];
            if (q) v = v + q[c];
            r[c] = relu ? (v > 0.f ? v : 0.f) : v;
        }
    }
    reg(net, regname, *y);
    return 0;
}

/* tc.layers.max_pool2d(kernel 3, stride 2 (contrib default), SAME) vnect_model.py:29 */
static tens maxpool3s2(tens x)
{
    int ho, wo, pt, pl;
    same_pad(x.h, 3, 2, &ho, &pt);
    same_pad(x.w, 3, 2, &wo, &pl);
    tens y = talloc(x.n, ho, wo, x.c);
#pragma omp parallel for collapse(2) schedule(static)
    for (int s = 0; s < x.n; s++)
        for (int oy = 0; oy < ho; oy++)
            for (int ox = 0; ox < wo; ox++) {
                float* o = y.d + (((size_t)s * ho + oy) * wo + ox) * x.c;
                for (int c = 0; c < x.c; c++) o[c] = -INFINITY;
                for (int ky = 0; ky < 3; ky++)
                    for (int kx = 0; kx < 3; kx++) {
                        int iy = oy * 2 - pt + ky, ix = ox * 2 - pl + kx;
                        if (iy < 0 || iy >= x.h || ix < 0 || ix >= x.w) continue;
                        const float* p = x.d + (((size_t)s * x.h + iy) * x.w + ix) * x.c;
                        for (int c = 0; c < x.c; c++)
                            if (p[c] > o[c]) o[c] = p[c];
                    }
            }
    return y;
}

/* tf.layers.conv2d_transpose(kernel 4, strides 2, SAME, no bias) vnect_model.py:188-193.
 * Conv2DBackpropInput of a forward 4x4/s2/SAME conv 46->23 (pad 1/1):
 *   out[2i-1+ky, 2j-1+kx, oc] += in[i,j,ic] * W[ky,kx,oc,ic].
 * Evaluated per output phase (oy&1, ox&1) as a 2x2-tap GEMM with K = 4*Cin. */
static int deconv4s2(vo_net* net, const char* scope, tens x, int cout, tens* y)
{
    const vo_array* W = find_w(net, scope, "kernel");
    if (!W) return -1;
    if (W->shape[0] != 4 || W->shape[2] != cout || W->shape[3] != x.c) {
        snprintf(net->err, sizeof net->err, "shape mismatch at %s", scope);
        return -1;
    }
    int cin = x.c, ho = 2 * x.h, wo = 2 * x.w, M = x.n * x.h * x.w;
    *y = talloc(x.n, ho, wo, cout);
    float* Bm = (float*)malloc((size_t)4 * cin * cout * sizeof(float));
    float* Cm = (float*)malloc((size_t)M * cout * sizeof(float));
    for (int py = 0; py < 2; py++)
        for (int px = 0; px < 2; px++) {
            /* oy = 2i' + py.  py=0: ky in {1,3} -> i = i', i'-1;  py=1: ky in {0,2} -> i = i'+1, i' */
            int kys[2], dys[2], kxs[2], dxs[2];
            if (py == 0) { kys[0] = 1, dys[0] = 0, kys[1] = 3, dys[1] = -1; }
            else { kys[0] = 0, dys[0] = 1, kys[1] = 2, dys[1] = 0; }
            if (px == 0) { kxs[0] = 1, dxs[0] = 0, kxs[1] = 3, dxs[1] = -1; }
            else { kxs[0] = 0, dxs[0] = 1, kxs[1] = 2, dxs[1] = 0; }
            gather_t g = {x.d, x.n, x.h, x.w, cin, x.h, x.w, 1, 4, {0}, {0}};
            for (int a = 0; a < 2; a++)
                for (int b = 0; b < 2; b++) {
                    int t = a * 2 + b;
                    g.dy[t] = dys[a], g.dx[t] = dxs[b];
                    const float* wk = W->data + ((size_t)(kys[a] * 4 + kxs[b]) * cout) * cin; /* [oc][ic] */
                    for (int ic = 0; ic < cin; ic++)
                        for (int oc = 0; oc < cout; oc++)
                            Bm[((size_t)t * cin + ic) * cout + oc] = wk[(size_t)oc * cin + ic];
                }
            gemm_gather(&g, M, cout, 4 * cin, Bm, cout, Cm, cout);
            for (int m = 0; m < M; m++) {
                int j = m % x.w, i = (m / x.w) % x.h, s = m / (x.w * x.h);
                memcpy(y->d + (((size_t)s * ho + 2 * i + py) * wo + 2 * j + px) * cout, Cm + (size_t)m * cout,
                       cout * sizeof(float));
            }
        }
    free(Bm);
    free(Cm);
    reg(net, scope, *y);
    return 0;
}

#define CK(e)            \
    do {                 \
        if ((e)) goto fail; \
    } while (0)

/* bottleneck with identity shortcut: relu(c(b(a(x))) + x)  (e.g. vnect_model.py:44-51) */
static int block_id(vo_net* net, const char* pfx, tens x, int mid, int out, tens* y)
{
    char nm[64];
    tens a, b;
    snprintf(nm, 64, "%s_branch2a", pfx);
    if (conv(net, nm, x, 1, 1, mid, 1, NULL, &a)) return -1;
    snprintf(nm, 64, "%s_branch2b", pfx);
    if (conv(net, nm, a, 3, 1, mid, 1, NULL, &b)) return -1;
    snprintf(nm, 64, "%s_branch2c", pfx);
    if (conv(net, nm, b, 1, 1, out, 1, &x, y)) return -1; /* (conv+bias) + shortcut, then relu */
    return 0;
}

/* bottleneck with projection shortcut (vnect_model.py:31-41, 63-73, 105-115) */
static int block_proj(vo_net* net, const char* pfx, tens x, int mid, int out, int stride, tens* y)
{
    char nm[64];
    tens s, a, b;
    snprintf(nm, 64, "%s_branch1", pfx);
    if (conv(net, nm, x, 1, stride, out, 0, NULL, &s)) return -1;
    snprintf(nm, 64, "%s_branch2a", pfx);
    if (conv(net, nm, x, 1, stride, mid, 1, NULL, &a)) return -1;
    snprintf(nm, 64, "%s_branch2b", pfx);
    if (conv(net, nm, a, 3, 1, mid, 1, NULL, &b)) return -1;
    snprintf(nm, 64, "%s_branch2c", pfx);
    if (conv(net, nm, b, 1, 1, out, 1, &s, y)) return -1; /* tf.add(branch2c, branch1) then relu */
    return 0;
}

int vo_net_forward(vo_net* net, const float* batch, int S, float* out)
{
    clear_acts(net);
    net->err[0] = 0;
    tens x = talloc(S, 368, 368, 3);
    memcpy(x.d, batch, (size_t)S * 368 * 368 * 3 * sizeof(float));
    reg(net, "input", x);
    tens conv1, pool1, r2a, r2b, r2c, r3, r4, t, u;
    /* vnect_model.py:27-29 */
    CK(conv(net, "conv1", x, 7, 2, 64, 1, NULL, &conv1));
    pool1 = maxpool3s2(conv1);
    reg(net, "pool1", pool1);
    /* res2a :31-41, res2b :43-51 */
    CK(block_proj(net, "res2a", pool1, 64, 256, 1, &r2a));
    CK(block_id(net, "res2b", r2a, 64, 256, &r2b));
    /* res2c :53-61 -- QUIRK: branch2b consumes res2b_branch2a, branch2a is dead (TF prunes it) */
    {
        tens a, b;
        int sh[4];
        if (net->paper_res2c) {
            CK(conv(net, "res2c_branch2a", r2b, 1, 1, 64, 1, NULL, &a));
        } else {
            a.d = (float*)vo_net_activation(net, "res2b_branch2a", sh);
            a.n = sh[0], a.h = sh[1], a.w = sh[2], a.c = sh[3];
        }
        CK(conv(net, "res2c_branch2b", a, 3, 1, 64, 1, NULL, &b));
        CK(conv(net, "res2c_branch2c", b, 1, 1, 256, 1, &r2b, &r2c));
    }
    /* res3a-d :63-103 */
    CK(block_proj(net, "res3a", r2c, 128, 512, 2, &r3));
    CK(block_id(net, "res3b", r3, 128, 512, &t)); r3 = t;
    CK(block_id(net, "res3c", r3, 128, 512, &t)); r3 = t;
    CK(block_id(net, "res3d", r3, 128, 512, &t)); r3 = t;
    /* res4a-f :105-165 */
    CK(block_proj(net, "res4a", r3, 256, 1024, 2, &r4));
    CK(block_id(net, "res4b", r4, 256, 1024, &t)); r4 = t;
    CK(block_id(net, "res4c", r4, 256, 1024, &t)); r4 = t;
    CK(block_id(net, "res4d", r4, 256, 1024, &t)); r4 = t;
    CK(block_id(net, "res4e", r4, 256, 1024, &t)); r4 = t;
    CK(block_id(net, "res4f", r4, 256, 1024, &t)); r4 = t;
    /* res5a :167-177 (stride 1), res5b :179-185 (all ReLU, no shortcut) */
    tens r5a, r5b;
    {
        tens a, b, s;
        CK(conv(net, "res5a_branch2a_new", r4, 1, 1, 512, 1, NULL, &a));
        CK(conv(net, "res5a_branch2b_new", a, 3, 1, 512, 1, NULL, &b));
        CK(conv(net, "res5a_branch1_new", r4, 1, 1, 1024, 0, NULL, &s));
        CK(conv(net, "res5a_branch2c_new", b, 1, 1, 1024, 1, &s, &r5a));
        CK(conv(net, "res5b_branch2a_new", r5a, 1, 1, 256, 1, NULL, &a));
        CK(conv(net, "res5b_branch2b_new", a, 3, 1, 128, 1, NULL, &b));
        CK(conv(net, "res5b_branch2c_new", b, 1, 1, 256, 1, NULL, &r5b));
    }
    /* transposed convs + BN + bone length + concat :187-209 */
    tens d1, d2, feat;
    CK(deconv4s2(net, "res5c_branch1a", r5b, 63, &d1));
    CK(deconv4s2(net, "res5c_branch2a", r5b, 128, &d2));
    {
        const vo_array* ga = find_w(net, "bn5c_branch2a", "gamma");
        const vo_array* be = find_w(net, "bn5c_branch2a", "beta");
        const vo_array* mu = find_w(net, "bn5c_branch2a", "moving_mean");
        const vo_array* va = find_w(net, "bn5c_branch2a", "moving_variance");
        if (!ga || !be || !mu || !va) goto fail;
        float sf[128];
        /* FusedBatchNorm inference, contrib batch_norm default epsilon 0.001:
         * y = (x - mean) * (gamma * rsqrt(var + eps)) + beta */
        for (int c = 0; c < 128; c++) sf[c] = ga->data[c] * (1.0f / sqrtf(va->data[c] + 0.001f));
        feat = talloc(S, 46, 46, 212);
        size_t P = (size_t)S * 46 * 46;
#pragma omp parallel for schedule(static)
        for (size_t p = 0; p < P; p++) {
            float* f = feat.d + p * 212;
            const float* a = d2.d + p * 128;
            const float* dl = d1.d + p * 63;
            for (int c = 0; c < 128; c++) {
                float v = (a[c] - mu->data[c]) * sf[c] + be->data[c];
                f[c] = v > 0.f ? v : 0.f;
            }
            for (int c = 0; c < 63; c++) f[128 + c] = dl[c]; /* delta x | y | z */
            for (int j = 0; j < 21; j++) {
                float sx = dl[j] * dl[j], sy = dl[21 + j] * dl[21 + j], sz = dl[42 + j] * dl[42 + j];
                f[191 + j] = sqrtf((sx + sy) + sz);
            }
        }
        reg(net, "res5c_branch2a_feat", feat);
    }
    /* head :211-217 */
    CK(conv(net, "res5c_branch2b", feat, 3, 1, 128, 1, NULL, &t));
    {
        const vo_array* W = find_w(net, "res5c_branch2c", "kernel");
        if (!W) goto fail;
        gather_t g = {t.d, t.n, t.h, t.w, t.c, t.h, t.w, 1, 1, {0}, {0}};
        u = talloc(S, 46, 46, 84);
        gemm_gather(&g, S * 46 * 46, 84, 128, W->data, 84, u.d, 84);
        reg(net, "res5c_branch2c", u);
    }
    memcpy(out, u.d, (size_t)S * 46 * 46 * 84 * sizeof(float));
    if (!net->keep) clear_acts(net);
    return 0;
fail:
    clear_acts(net);
    return -1;
}
