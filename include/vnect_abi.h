/*
 * vnect_abi.h -- C ABI of libvnect_hip.so, the MI355X (gfx950) VNect inference path.
 *
 * Drop-in boundary.  In the reference (XinArkh/VNect, paths relative to /root/reference) the
 * per-frame path is src/estimator.py:97-142 (VNectEstimator.__call__); its only native boundary
 * is tf.Session.run at src/estimator.py:100-104.  This library replaces that call AND the Python
 * pre/post-processing around it, so the host keeps one call per frame.  Every entry point cites
 * the reference code it replaces.  Plain C types only; caller owns every host pointer (valid only
 * for the duration of the call); the library owns all device memory, streams, graphs and the
 * OneEuro filter state (one handle == one video stream == one GPU).
 *
 * Conventions: every int-returning function returns VNECT_OK (0) or a negative VNECT_E_* code;
 * no C++ exception crosses the boundary; vnect_last_error() gives the message.  Calls on one
 * handle must be serialised by the caller; different handles are independent.
 */
#ifndef VNECT_ABI_H
#define VNECT_ABI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VNECT_ABI_VERSION 6
#define VNECT_MAX_SCALES 8
#define VNECT_MAX_STREAMS 4 /* independent video streams one handle can serve (vnect_submit_stream) */
#define VNECT_BOX 368      /* src/estimator.py:19 box_size   */
#define VNECT_HM 46        /* box_size / hm_factor (:21)     */
#define VNECT_JOINTS 21    /* src/estimator.py:23 joints_sum */

enum {
    VNECT_OK = 0,
    VNECT_E_ARG = -1,       /* bad argument / shape / name                          */
    VNECT_E_STATE = -2,     /* call order (e.g. infer before finalize)              */
    VNECT_E_HIP = -3,       /* HIP runtime error (message has hipGetErrorString)    */
    VNECT_E_NODEVICE = -4,  /* no usable gfx950 device                              */
    VNECT_E_TIMESTAMP = -5, /* timestamp equals the previous one: the reference raises
                               ZeroDivisionError at src/OneEuroFilter.py:66          */
    VNECT_E_COMM = -6,      /* RCCL / peer-access error                             */
    VNECT_E_TIMEORDER = -7, /* timestamp earlier than the previous one: the reference's negative freq drives
                               alpha out of (0, 1] and LowPassFilter.__setAlpha raises ValueError
                               (src/OneEuroFilter.py:19-23, 58-66)                   */
    VNECT_E_INTERNAL = -8   /* host allocation failure or an internal C++ exception, caught at the boundary */
};

enum { VNECT_FP32 = 0,        /* fp32 tensors, v_mfma_f32_32x32x2_f32 (BASELINE.json configs[1])                                      */
       VNECT_BF16 = 1,        /* bf16 tensors and weights, fp32 accumulate (configs[2])                                               */
       VNECT_FP32_SPLIT = 2   /* fp32 tensors, fp32 accumulate; the PRODUCTS of the 64x64-tile layers run on the bf16 matrix pipe as
                                 exact three-way splits (x = xh + xm + xl, 6 of the 9 piece products; conv.hip, X3): fp32-class
                                 results -- gated like VNECT_FP32 -- at 2.7x the matrix rate of the fp32 instruction                  */ };

typedef struct vnect_handle vnect_handle;

typedef struct vnect_config {
    int32_t struct_size;              /* = sizeof(vnect_config)                                   */
    int32_t device;                   /* HIP device ordinal                                       */
    int32_t num_scales;               /* len(self.scales), src/estimator.py:32                    */
    double scales[VNECT_MAX_SCALES];  /* each in (0, 1]                                           */
    int32_t precision;                /* VNECT_FP32 | VNECT_BF16 | VNECT_FP32_SPLIT               */
    int32_t paper_res2c;              /* 0 = reference wiring src/vnect_model.py:56 (default)     */
    int32_t use_graph;                /* 0 = eager launches; 1 = replay the frame as one hipGraph;
                                         2 = auto: eager for a frame submitted while none is in flight
                                         (synchronous use), graph replay behind frames in flight    */
    int32_t numpy_promotion;          /* float32-fed 3-D filters: 0 = numpy 1.x rules (the only
                                         numpy TF1 runs with), 1 = NEP 50 (numpy >= 2)            */
    int32_t max_frame_bytes;          /* capacity of one resident frame slot; 0 -> 4096*4096*3     */
    int32_t num_frame_slots;          /* resident frame slots (>=1); 0 -> 4                        */
    int32_t pyramid_nranks;           /* 0/1 = off; else must equal num_scales: this handle runs the
                                         pre-processing and conv stack of ONE scale (see vnect_comm_init) */
    int32_t pyramid_rank;             /* which scale (0 .. pyramid_nranks-1)                       */
    int32_t keep_activations;         /* 0 (default) = layer outputs share an activation arena sized for the peak
                                         live set, so weights + activations stay in the Infinity Cache between
                                         frames; 1 = one private buffer per layer output, which is what
                                         vnect_read_activation needs to return an inner layer (tests, debugging) */
    int32_t lanes;                    /* 0/1 (default): frames run one after the other.  2 or 3: a frame submitted while others
                                         are in flight (vnect_submit_resident before vnect_collect; up to `lanes` frames)
                                         runs on a lane of its own -- own stream, activation arena and graph, same weights --
                                         and overlaps with them; only the joints kernels stay in order (the OneEuro filters
                                         are a chain).  Results are bit-identical to sequential execution; each extra lane
                                         costs one more activation arena (~0.1 GB)                                          */
    int32_t preprocess_only;          /* 1: the handle serves vnect_preprocess (and vnect_set_scales) only -- the static
                                         gen_input_batch of src/estimator.py:70-81 needs no session either: no weights, no
                                         launch plan, ~20 MB of device memory; vnect_finalize and inference are refused      */
    int32_t exchange;                 /* pyramid sharding: how a rank's maps reach the others (VNECT_XCHG_*)                 */
} vnect_config;

enum { VNECT_XCHG_RCCL = 0,   /* ncclAllGather on the handle's stream (default)                                             */
       VNECT_XCHG_P2P = 1 };  /* peer writes over xGMI: a copy kernel stores the rank's 710 976 B straight into every peer's
                                 gather slot (hipDeviceEnablePeerAccess + IPC-mapped buffers), one flag per rank          */

/* Replaces VNectEstimator.__init__ (src/estimator.py:27-68): session + graph + 42 + 63 filters. */
int vnect_create(const vnect_config* cfg, vnect_handle** out);
void vnect_destroy(vnect_handle* h);
/* Message of the last failure on h (h == NULL: last vnect_create failure). Never NULL. */
const char* vnect_last_error(vnect_handle* h);
int vnect_abi_version(void);

/* ABI v6.  How THIS binary was built, as one line of `key=value` text (static storage, never NULL): `abi`, `compiler`, `flags`, `variant`
 * (the Makefile's VARIANT, empty for the product), `test_hooks` (1 only in the test build that can inject warm-start failures), every
 * compile-time probe switch of the kernel sources with its value (X3_DBG, WT_DBG, CH_DBG, F32_NOSTORE, BF16_NOSTORE, VNECT_AB, NS_6432,
 * NS_32128, POST_DBG, ARG_SLABS, ARG_ROWSPLIT -- most of them give wrong results on purpose) and `probes_off` = 1 iff all of them are
 * at their product values.  A deployment check (and tests/test_abi_cpu.py) reads `probes_off=1 test_hooks=0`.  Replaces nothing in the
 * reference: its counterpart is knowing which TensorFlow build `import tensorflow` loaded (src/estimator.py:9).                        */
const char* vnect_build_info(void);

/* Replaces VNect.load_weights / assign_weights_from_dict (src/vnect_model.py:219-236).
 * name is a key of the reference's pickle schema (src/caffe2pkl.py:51-80), e.g.
 * "conv1/weights" (kh,kw,Cin,Cout), "conv1/biases", "res5c_branch1a/kernel" (kh,kw,Cout,Cin),
 * "bn5c_branch2a/gamma".  Data is copied.  All 109 arrays must be set before vnect_finalize. */
int vnect_set_weight(vnect_handle* h, const char* name, const float* data, const int64_t* shape, int ndim);
/* Packs weights into kernel layouts, uploads them, plans buffers, builds the launch sequence, and warms the handle up on a few grey
 * frames (so that the first real frame runs at steady-state speed; the filters and frame slots are those of a fresh handle
 * afterwards).  A launch or device error DURING the warm start is returned (VNECT_E_HIP / VNECT_E_INTERNAL: the plan this handle
 * would run for every frame is broken); a refused grey frame only skips the warm start and leaves a note in vnect_last_error.
 * (The reference's counterpart is saver.restore, src/estimator.py:55-60.) */
int vnect_finalize(vnect_handle* h);

/* `self.scales = [...]` (src/estimator.py:32 is a plain attribute callers may overwrite). */
int vnect_set_scales(vnect_handle* h, const double* scales, int num_scales);

/* Replaces sess.run([heatmap,x,y,z], {input: batch}) (src/estimator.py:100-104):
 * batch (S,368,368,3) f32 NHWC -> out (S,46,46,84) f32, channels [hm | x | y | z] x 21
 * (the tf.split at src/vnect_model.py:216-217 is a view of this tensor). S = num_scales. */
int vnect_forward(vnect_handle* h, const float* batch_nhwc, int num_images, float* out_maps);

/* Replaces VNectEstimator.gen_input_batch (src/estimator.py:70-81) on the device.
 * bgr: (H,W,3) uint8, row_stride bytes per row.  batch_out may be NULL (batch stays on device). */
int vnect_preprocess(vnect_handle* h, const uint8_t* bgr, int H, int W, int64_t row_stride, float* batch_out,
                     double* scaler, int32_t* offset_x, int32_t* offset_y);

/* Replaces src/estimator.py:105-139 (multi-scale merge, extract_2d_joints, joint_filter,
 * extract_3d_joints, joint_filter, un-mapping) for maps supplied by the caller.
 * t2d / t3d stand for the two time.time() reads at src/estimator.py:84. */
int vnect_postprocess(vnect_handle* h, const float* maps, double t2d, double t3d, double scaler, int32_t offset_x,
                      int32_t offset_y, double* joints_2d /*21x2 [row,col]*/, float* joints_3d /*21x3*/);

/* Replaces VNectEstimator.__call__ (src/estimator.py:97-142) end to end. */
int vnect_infer(vnect_handle* h, const uint8_t* bgr, int H, int W, int64_t row_stride, double t2d, double t3d,
                double* joints_2d, float* joints_3d);

/* Where a capture pipeline should put its frames so that vnect_infer's host-to-device copy needs no CPU copy first: the handle's
 * two PINNED staging buffers (index 0 / 1, at least min_bytes each, <= max_frame_bytes; the pointer stays valid until a larger
 * request for the same index or vnect_destroy).  The reference's loop gets its frame from cv2.VideoCapture.read into pageable numpy
 * memory (run_estimator_ps.py:75-82) and hands it to __call__ (src/estimator.py:97-99); vnect_infer accepts ANY host pointer -- a
 * pageable one is copied into these buffers by the CPU (one memcpy), a pointer INSIDE one of them (the whole frame or a strided crop
 * of it) is DMA-ed from where it lies.  Either way the device copy is asynchronous on the frame's stream. */
int vnect_frame_buffer(vnect_handle* h, int index, int64_t min_bytes, uint8_t** ptr_out);

/* Frames resident in HBM (throughput measurement; a capture pipeline that DMA-writes frames).
 * upload copies a frame into slot; infer_resident runs __call__ on it without a host->device copy. */
int vnect_upload_frame(vnect_handle* h, int slot, const uint8_t* bgr, int H, int W, int64_t row_stride);
int vnect_infer_resident(vnect_handle* h, int slot, double t2d, double t3d, double* joints_2d, float* joints_3d);
/* Pipelining of the same call: submit enqueues a frame, collect waits for the oldest un-collected one (FIFO).
 * At most max(lanes, 2) frames may be in flight (one lane: the second frame queues behind the first on its stream). */
int vnect_submit_resident(vnect_handle* h, int slot, double t2d, double t3d);
int vnect_collect(vnect_handle* h, double* joints_2d, float* joints_3d);
/* Several independent video streams on ONE handle (one weight copy, `lanes` activation arenas): the reference runs one
 * VNectEstimator -- one set of 42 + 63 OneEuro filters, src/estimator.py:46-52 -- per video, each in a process of its own
 * (run_estimator_ps.py:120-129); here stream s (0 .. VNECT_MAX_STREAMS-1) is a filter bank + its timestamps on the device, and
 * vnect_submit_stream(h, s, ...) is that stream's __call__.  Frames of DIFFERENT streams have no dependency at all, so with
 * lanes = 2 / 3 they overlap completely (frames of one stream still chain through their filters, exactly as vnect_submit_resident
 * -- which is stream 0 -- does).  Results come back in submission order; vnect_collect_stream also says whose they are.  Each
 * stream's results are bit-identical to a handle of its own fed the same frames. */
int vnect_submit_stream(vnect_handle* h, int stream, int slot, double t2d, double t3d);
int vnect_collect_stream(vnect_handle* h, int32_t* stream_out, double* joints_2d, float* joints_3d);
/* New filters for ONE stream (vnect_reset_filters resets all of them). */
int vnect_reset_filters_stream(vnect_handle* h, int stream);

/* Replaces VNectEstimator.joint_filter(joints, dim) (src/estimator.py:83-95) on its own: the handle's 2-D (dim 2: 21x2)
 * or 3-D (dim 3: 21x3) OneEuro bank applied to caller-supplied joints at timestamp t (the reference reads time.time() once
 * per call, :84).  Values travel as float64; values_are_f32 = 1 says they are numpy float32 scalars -- what the reference
 * feeds the 3-D bank (:91-93) -- so numpy's scalar promotion rules (vnect_config::numpy_promotion) decide the arithmetic.
 * Same timestamp errors as vnect_infer.  The filter state lives on the device; this runs one tiny kernel. */
int vnect_joint_filter(vnect_handle* h, int dim, const double* joints_in, int values_are_f32, double t, double* joints_out);

/* New filters, as constructing a fresh VNectEstimator would (src/estimator.py:46-52). */
int vnect_reset_filters(vnect_handle* h);

/* Layer output of the last vnect_forward / vnect_infer, by reference scope name ("conv1", "pool1",
 * "res2a_branch2a", block outputs "res2a".."res5a", "res5c_branch2a_feat", "res5c_branch2c", ...).
 * Writes N,H,W,C to shape4 and up to capacity floats (dense NHWC, padding channels stripped).
 * Parity/debug aid; not on the hot path. */
int vnect_read_activation(vnect_handle* h, const char* name, float* out, int64_t capacity, int32_t* shape4);

typedef struct vnect_timings {
    int32_t struct_size;
    int32_t frames;            /* profiled frames since the last vnect_reset_timings                              */
    double total_ms;           /* HIP events around the whole frame (pre + conv stack + post), summed over frames  */
    double net_ms;             /* first conv kernel start .. last conv kernel end (device clock), summed           */
    double conv_ms;            /* sum of the conv kernels' own durations (device clock, like rocprofv3), summed    */
    int32_t conv_launches;     /* conv kernel launches per frame                                                   */
    double conv_flops;         /* algorithmic conv FLOPs per frame (2*MAC, live graph)                             */
    double conv_slot_ms;       /* sum over the conv kernels of the slot each one occupies on the stream: its start to the
                                  start of the kernel behind it (own duration + median boundary where another kind of
                                  kernel follows).  rocprofv3's per-kernel durations abut the same way (dispatch ->
                                  completion), so this is the figure its --stats average agrees with.  Summed.          */
    /* ABI v6 (a caller compiled against v5 passes the shorter struct_size and gets the fields above): the shader clock the chip HELD
     * while the conv launches of the profiled frames ran -- workgroup 0 of every conv launch stamps the shader-cycle counter
     * (s_memtime) and the 100 MHz clock (s_memrealtime) at its start and its end; clock [MHz] = 100 * shader_cycles / shader_ticks.
     * What an N > 1 run needs to tell a clock-limited rank (eight GPUs sharing a node's power) from a host-limited one.              */
    double shader_cycles;      /* sum over conv launches and profiled frames of workgroup 0's span in shader cycles                */
    double shader_ticks;       /* the same spans in 100 MHz ticks                                                                  */
} vnect_timings;
/* Profiling replays a twin of the frame graph in which every conv kernel stamps its start and end with the
 * 100 MHz device clock (s_memrealtime); off by default, no cost when off. */
int vnect_set_profiling(vnect_handle* h, int on);
int vnect_get_timings(vnect_handle* h, vnect_timings* out);
int vnect_reset_timings(vnect_handle* h);

/* Per-layer launch plan: fills name (<=63 chars + NUL) and the numbers for layer idx; returns
 * VNECT_E_ARG when idx is past the last layer. */
typedef struct vnect_layer_info {
    char name[64];
    int32_t M, N, K;           /* implicit-GEMM view                                              */
    int32_t tile_m, tile_n, split_k, workgroups;
    double flops;              /* algorithmic                                                     */
    double last_ms;            /* kernel duration in the last profiled frame, device clock (0 if none) */
} vnect_layer_info;
int vnect_get_layer_info(vnect_handle* h, int idx, vnect_layer_info* out);
/* Raw 100 MHz device-clock stamps of layer idx in the last profiled frame (tuning aid): [0] earliest workgroup start,
 * [1..8] latest workgroup ends, [9..13] workgroup 0: start, operands requested, first chunk in LDS, K loop done,
 * stores done; [15] start of the last-dispatched workgroups; [16..18] shader-clock cycles producer wave 0 of workgroup
 * 0 spent waiting for landings / at the barrier / issuing; [19] cycles consumer wave 0 waited at the barrier.  (24 values:
 * the shader-clock stamps of ABI v6 are reported through vnect_timings, not here.) */
int vnect_get_layer_stamps(vnect_handle* h, int idx, uint64_t* out24);

/* Pyramid sharding over RCCL (one scale per rank; BASELINE.json configs[3]).  A handle created with
 * pyramid_nranks == num_scales runs gen_input_batch and the conv stack for scale `pyramid_rank` only (the S
 * images of the batch are independent through the net, src/estimator.py:75-80,100-104); its (46,46,84) maps are
 * all-gathered (ncclAllGather, 710 976 B per rank) and every rank finishes the post-processing, so every rank
 * returns the same joints.  All ranks must be fed the same frame and timestamps.  unique_id is an ncclUniqueId
 * (128 bytes) made by rank 0 with vnect_comm_unique_id and distributed by the host (torch.distributed / files).
 * vnect_comm_init must be called once, after vnect_create and before the first inference.  vnect_forward on such a
 * handle takes ONE image (its scale). */
int vnect_comm_unique_id(void* id128);
int vnect_comm_init(vnect_handle* h, int rank, int nranks, const void* id128);
/* Which RCCL the library resolved, and how: the file ncclAllGather lives in (dladdr) goes to path_out (NUL-terminated, truncated to
 * capacity); *reused_out = 1 when a copy the process had mapped already was reused -- the caller's torch.distributed "nccl" backend
 * loads torch's bundled librccl.so; the reference has no counterpart (its one process owns one TF session,
 * run_estimator_ps.py:35-36, 120-129) -- and 0 when the library opened librccl.so.1 itself.  VNECT_RCCL_LIB overrides the choice.
 * Loads RCCL if it is not loaded yet (VNECT_E_COMM if there is none). */
int vnect_comm_library(char* path_out, int capacity, int32_t* reused_out);
/* The same exchange by plain peer writes over xGMI instead of RCCL (vnect_config::exchange = VNECT_XCHG_P2P; SURVEY 8e asks for
 * both, measured side by side: 711 KB per rank is ~4.6 us of wire time, a collective's launch + sync latency is tens of us).
 * Every rank exports a 128-byte blob describing its exchange block (IPC memory handle + the address inside the exporting
 * process), the host distributes the blobs (torch.distributed all_gather_object / files), and vnect_comm_p2p_init(rank, nranks,
 * blobs[nranks * 128]) maps the peers' blocks (hipIpcOpenMemHandle across processes; hipDeviceEnablePeerAccess when the peer
 * handle lives in this process).  Per frame one kernel stores the rank's maps into every peer's block, publishes a per-rank flag
 * and gathers the peers' slots; a peer that does not show up within ~2 s makes the frame fail with VNECT_E_COMM, never hang. */
int vnect_comm_p2p_export(vnect_handle* h, void* blob128);
int vnect_comm_p2p_init(vnect_handle* h, int rank, int nranks, const void* blobs);

#ifdef __cplusplus
}
#endif
#endif
