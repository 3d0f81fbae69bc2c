/* How much of a synchronous frame is the PYTHON call path?  The same loop bench.py times -- vnect_infer_resident on 8 resident frames -- from C:
 * reads the weights + frames blob of tests/c/infer_frame.c's format (tools/host_path_cost.py writes it), prints frames/s and the median frame.
 *   gcc -O2 -std=c99 -Iinclude tools/c_loop_rate.c -o tools/c_loop_rate -Lvnect_amd/lib -lvnect_hip -Wl,-rpath,$PWD/vnect_amd/lib -Wl,-rpath,/opt/rocm/lib -Wl,--allow-shlib-undefined */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "vnect_abi.h"
static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void rd(void* p, size_t n, FILE* f)
{
    if (fread(p, 1, n, f) != n) exit(3);
}
static int cmp(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }
int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    const int bf16 = argc > 2 && !strcmp(argv[2], "bf16");
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    vnect_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (int32_t)sizeof cfg, cfg.num_scales = 3, cfg.scales[0] = 1.0, cfg.scales[1] = 0.8, cfg.scales[2] = 0.6;
    cfg.precision = bf16 ? VNECT_BF16 : VNECT_FP32, cfg.use_graph = 2, cfg.num_frame_slots = 8, cfg.lanes = 3;
    vnect_handle* h = NULL;
    if (vnect_create(&cfg, &h)) return 2;
    int32_t nw;
    rd(&nw, 4, f);
    for (int i = 0; i < nw; i++) {
        int32_t len, ndim;
        char name[128];
        int64_t shape[4];
        size_t n = 1;
        rd(&len, 4, f), rd(name, (size_t)len, f), name[len] = 0, rd(&ndim, 4, f), rd(shape, 8 * (size_t)ndim, f);
        for (int d = 0; d < ndim; d++) n *= (size_t)shape[d];
        float* data = (float*)malloc(n * 4);
        rd(data, n * 4, f);
        if (vnect_set_weight(h, name, data, shape, ndim)) return 2;
        free(data);
    }
    if (vnect_finalize(h)) return 2;
    int32_t nf, H, W;
    rd(&nf, 4, f), rd(&H, 4, f), rd(&W, 4, f);
    uint8_t* bgr = (uint8_t*)malloc((size_t)H * W * 3);
    for (int k = 0; k < nf && k < 8; k++) {
        rd(bgr, (size_t)H * W * 3, f);
        if (vnect_upload_frame(h, k, bgr, H, W, (int64_t)W * 3)) return 2;
    }
    const int N = 2000, WARM = 100;
    static double lat[2000];
    double j2[42];
    float j3[63];
    double t = 1.7e9, t0 = 0;
    for (int i = 0; i < WARM + N; i++) {
        if (i == WARM) t0 = now();
        const double a = now();
        t += 1.0 / 30;
        if (vnect_infer_resident(h, i % (nf < 8 ? nf : 8), t, t + 1e-3, j2, j3)) return 2;
        if (i >= WARM) lat[i - WARM] = now() - a;
    }
    const double el = now() - t0;
    qsort(lat, N, sizeof(double), cmp);
    printf("%s C loop: %.1f frames/s over %d frames, median frame %.4f ms (%.1f frames/s)\n", bf16 ? "bf16" : "fp32", N / el, N, lat[N / 2] * 1e3, 1.0 / lat[N / 2]);
    vnect_destroy(h);
    return 0;
}
