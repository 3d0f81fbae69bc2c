/* A C caller of libvnect_hip.so with no Python in the process (tests/test_gpu_surface.py::test_a_c_program_runs_frames_through_the_abi):
 * reads the weights (the schema of /root/reference/src/vnect_model.py:219-236) and BGR frames from a flat binary file the test wrote, runs
 * VNectEstimator.__call__ (/root/reference/src/estimator.py:97-142) on each frame through vnect_infer and prints the joints as hex floats,
 * which the test compares bit for bit with the ctypes path.
 *
 * file layout (little endian): int32 n_weights; per weight: int32 name_len, name bytes, int32 ndim, int64 shape[ndim], float32 data[];
 *                              int32 n_frames, int32 H, int32 W; per frame: uint8 bgr[H * W * 3]
 * usage: infer_frame weights_and_frames.bin [bf16]                                                                                      */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vnect_abi.h"

static void die(vnect_handle* h, const char* what, int rc)
{
    fprintf(stderr, "%s: code %d: %s\n", what, rc, vnect_last_error(h));
    exit(2);
}
static void rd(void* p, size_t n, FILE* f)
{
    if (fread(p, 1, n, f) != n) {
        fprintf(stderr, "short read\n");
        exit(3);
    }
}

int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    vnect_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (int32_t)sizeof cfg;
    cfg.device = 0;
    cfg.num_scales = 3;
    cfg.scales[0] = 1.0, cfg.scales[1] = 0.8, cfg.scales[2] = 0.6;
    cfg.precision = (argc > 2 && !strcmp(argv[2], "bf16")) ? VNECT_BF16 : VNECT_FP32;
    cfg.use_graph = 2;
    vnect_handle* h = NULL;
    int rc = vnect_create(&cfg, &h);
    if (rc) die(h, "vnect_create", rc);
    int32_t nw;
    rd(&nw, 4, f);
    for (int i = 0; i < nw; i++) {
        int32_t len, ndim;
        char name[128];
        int64_t shape[4];
        size_t n = 1;
        rd(&len, 4, f);
        if (len <= 0 || len >= (int32_t)sizeof name) return 4;
        rd(name, (size_t)len, f);
        name[len] = 0;
        rd(&ndim, 4, f);
        if (ndim < 1 || ndim > 4) return 4;
        rd(shape, 8 * (size_t)ndim, f);
        for (int d = 0; d < ndim; d++) n *= (size_t)shape[d];
        float* data = (float*)malloc(n * 4);
        rd(data, n * 4, f);
        if ((rc = vnect_set_weight(h, name, data, shape, ndim))) die(h, name, rc);
        free(data);
    }
    if ((rc = vnect_finalize(h))) die(h, "vnect_finalize", rc);
    int32_t nf, H, W;
    rd(&nf, 4, f), rd(&H, 4, f), rd(&W, 4, f);
    uint8_t* bgr = (uint8_t*)malloc((size_t)H * W * 3);
    for (int k = 0; k < nf; k++) {
        double j2[VNECT_JOINTS * 2];
        float j3[VNECT_JOINTS * 3];
        const double t = 1.7e9 + k / 30.0;
        rd(bgr, (size_t)H * W * 3, f);
        if ((rc = vnect_infer(h, bgr, H, W, (int64_t)W * 3, t, t + 0.001, j2, j3))) die(h, "vnect_infer", rc);
        printf("frame %d", k);
        for (int i = 0; i < VNECT_JOINTS * 2; i++) printf(" %a", j2[i]);
        for (int i = 0; i < VNECT_JOINTS * 3; i++) printf(" %a", (double)j3[i]);
        printf("\n");
    }
    free(bgr);
    fclose(f);
    vnect_destroy(h);
    return 0;
}
