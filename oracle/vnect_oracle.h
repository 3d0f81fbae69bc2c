/*
 * vnect_oracle.h -- CPU oracle for the VNect inference path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.  The shipped path is the HIP library
 * declared in include/vnect_abi.h; it never calls into this file.
 *
 * What is restated (reference = XinArkh/VNect, paths relative to /root/reference):
 *   network graph          src/vnect_model.py:25-217   (vnect_net.c)
 *   input pyramid          src/estimator.py:70-81, src/utils.py:13-21,82-150
 *   multi-scale merge      src/estimator.py:105-129
 *   2-D joint extraction   src/utils.py:153-175
 *   OneEuro filter         src/OneEuroFilter.py:13-75, src/estimator.py:83-95
 *   3-D read-off           src/utils.py:58-79,178-219
 *   un-mapping             src/estimator.py:137-139
 *
 * Third-party arithmetic that is NOT in /root/reference and is restated from its published
 * semantics: TensorFlow 1.x ops (Conv2D / MaxPool / Conv2DBackpropInput / FusedBatchNorm, SAME
 * padding rule) and OpenCV cv2.resize INTER_LINEAR (8-bit fixed-point path, 11-bit
 * coefficients; float path).  Neither library nor trained weights are available, so:
 *
 *   PARITY PINNING: the OneEuro filter, the 3-D read-off and the __call__ glue are pinned
 *   against the reference's own Python run in the build container (the .npz files under tests/golden,
 *   made by tests/golden/make_golden.py).  The TF conv stack and cv2.resize are "parity
 *   unpinned" (no golden vectors exist in the reference); the conv stack is cross-checked
 *   against an independently written torch-CPU float64 restatement instead.
 */
#ifndef VNECT_ORACLE_H
#define VNECT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- network (vnect_net.c) ------------------------------------------------------- */
typedef struct vo_net vo_net;
vo_net* vo_net_create(void);
void vo_net_destroy(vo_net*);
/* name = reference schema key, e.g. "conv1/weights"; data is copied. returns 0 / -1 */
int vo_net_set_weight(vo_net*, const char* name, const float* data, const int64_t* shape, int ndim);
/* keep!=0: retain every named layer output for vo_net_activation().
 * paper_res2c!=0: feed res2c_branch2b from res2c_branch2a (paper wiring) instead of the
 * reference's res2b_branch2a (src/vnect_model.py:56). */
void vo_net_options(vo_net*, int keep, int paper_res2c);
/* batch: (S,368,368,3) f32 NHWC; out: (S,46,46,84) f32 = [heatmap|x|y|z] x 21. 0 / -1 */
int vo_net_forward(vo_net*, const float* batch, int S, float* out);
/* layer output kept by the last forward: returns pointer (owned by net) and fills shape[4]=N,H,W,C */
const float* vo_net_activation(vo_net*, const char* name, int* shape);
const char* vo_net_error(vo_net*);
int vo_sgemm_threads(void);
void vo_set_threads(int n);  /* OpenMP threads of the GEMM and element-wise loops (bench.py's cpu_baseline picks the fastest count) */
/* plain C = A*B helper exported for unit tests */
void vo_sgemm(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc);

/* ---- pre/post-processing (vnect_post.c) ------------------------------------------- */
int vo_cvround(double v);
/* cv2.resize(src,(0,0),fx=f,fy=f,INTER_LINEAR) restated; dst size = cvRound(size*f). */
void vo_resize_size(int sh, int sw, double f, int* dh, int* dw);
void vo_resize_u8(const uint8_t* src, int sh, int sw, int cn, double f, uint8_t* dst);
void vo_resize_f32(const float* src, int sh, int sw, int cn, double f, float* dst);
void vo_resize_f64(const double* src, int sh, int sw, int cn, double f, double* dst);

/* estimator.gen_input_batch: img (H,W,3) u8 with row_stride bytes -> batch (S,368,368,3) f32 */
int vo_gen_input_batch(const uint8_t* img, int H, int W, int64_t row_stride, const double* scales, int S,
                       float* batch, double* scaler, int* offset_x, int* offset_y);
/* estimator.py:105-129: maps (S,46,46,84) f32 -> avg (4,46,46,21) f64 [hm,x,y,z] */
void vo_merge_scales(const float* maps, const double* scales, int S, double* avg);
/* utils.extract_2d_joints on hm_avg (46,46,21) f64 -> (21,2) f64 [row,col] */
void vo_extract_2d(const double* hm_avg, double* joints_2d);
/* utils.extract_3d_joints: (21,2) f64 + x/y/z (46,46,21) f64 -> (21,3) f32 */
void vo_extract_3d(const double* joints_2d, const double* xm, const double* ym, const double* zm, float* joints_3d);
double vo_hm_pt_interp(const double* map46x46, int stride, double scale, double py, double px);

typedef struct vo_oef vo_oef;
vo_oef* vo_oef_create(double freq, double mincutoff, double beta, double dcutoff);
void vo_oef_destroy(vo_oef*);
/* returns 0 and *y, or -1 when timestamp == previous timestamp (ZeroDivisionError in the reference) */
int vo_oef_call(vo_oef*, double x, double timestamp, double* y);

/* whole estimator: owns a net + 42 + 63 filters (estimator.py:27-68) */
typedef struct vo_estimator vo_estimator;
vo_estimator* vo_est_create(vo_net* net /*borrowed*/, const double* scales, int S);
void vo_est_destroy(vo_estimator*);
void vo_est_reset(vo_estimator*);
/* numpy scalar-promotion flavour for the float32-fed 3-D filters: 0 = numpy 1.x (default), 1 = NEP 50 */
void vo_est_set_nep50(vo_estimator*, int on);
/* maps (S,46,46,84) -> joints; the post-network half of __call__ (estimator.py:105-139) */
int vo_est_postprocess(vo_estimator*, const float* maps, double t2d, double t3d, double scaler, int offset_x,
                       int offset_y, double* joints_2d, float* joints_3d);
/* full __call__ (estimator.py:97-142); t2d/t3d replace the two time.time() calls */
int vo_est_infer(vo_estimator*, const uint8_t* img, int H, int W, int64_t row_stride, double t2d, double t3d,
                 double* joints_2d, float* joints_3d);
#ifdef __cplusplus
}
#endif
#endif
