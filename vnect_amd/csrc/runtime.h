// runtime.h -- internal header of the host runtime of libvnect_hip.so (not part of the ABI: include/vnect_abi.h is).
// The runtime is four translation units along its seams (round 6; one 2 300-line file before):
//   rt_plan.cpp  weights -> packed device layouts, the launch plan (layers, tiles, fused forms), the activation arena, the resize tables
//   rt_exec.cpp  running frames: launch sequences, hipGraph, lanes (twins), streams, submit / collect, staging, warm start, roctx ranges
//   rt_comm.cpp  the pyramid exchange: RCCL (dlopen'ed) and peer writes, vnect_comm_*
//   rt_abi.cpp   the extern "C" entry points of include/vnect_abi.h (argument checks, device selection, the no-exception guard)
#pragma once
#include <dlfcn.h>
#include <link.h>
#include <unistd.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <stddef.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vnect_abi.h"
#include "hostplan.h"
#include "kernels.h"

namespace vnect {
namespace rt {

struct HostArray {
    std::vector<float> d;
    std::vector<int64_t> shape;
};

struct Tensor {
    std::string name;
    int S = 0, H = 0, W = 0, C = 0, Cs = 0;  // C valid channels, Cs elements per pixel
    int esz = 4;                             // bytes per element: 4 (fp32) or 2 (bf16)
    float* d = nullptr;                      // device buffer (bf16 data when esz == 2)
    size_t elems() const { return (size_t)S * H * W * Cs; }
    size_t bytes() const { return elems() * esz; }
};

enum OpKind { OP_CONV, OP_POOL, OP_BONE };

struct Layer {
    OpKind op = OP_CONV;
    std::string name;
    int in = -1, resid = -1, out = -1, out2 = -1;
    int out3 = -1;  // the chain GEMM's output tensor (the next block's branch2a), or -1
    int out_col0 = 0;  // first channel of `out` this launch writes (a paired launch whose head columns run as a launch of their own)
    ConvArgs a{};
    ReduceArgs r{};
    int BM = 64, BN = 64, KG = 1;  // tile shape; KG = in-workgroup K groups (conv.hip)
    int dy[MAX_TAPS] = {}, dx[MAX_TAPS] = {};  // filter taps [phase*ntaps + tap] (host side; the kernel gets them packed)
    float *w = nullptr, *bias = nullptr, *scale = nullptr, *shift = nullptr;
    float* frag_w = nullptr;  // a 1x1 pair on 64 input channels: the same weights in MFMA fragment order, for the stem's PAIR form
    int Nreal = 0, Kreal = 0;
    double flops = 0;
    float last_ms = 0;
};

constexpr int RING = 4;

}  // namespace rt
}  // namespace vnect

using namespace vnect;
using namespace vnect::rt;

struct vnect_handle {
    vnect_config cfg{};
    int S = 0;      // scales of the pyramid (merge, tables)
    int Snet = 0;   // images this handle pushes through the conv stack: S, or 1 when pyramid-sharded
    std::string err;
    bool finalized = false;
    bool pre_only = false;  // vnect_config::preprocess_only: the input batch buffer and the resize tables, nothing else
    bool x3 = false;    // VNECT_FP32_SPLIT: fp32 tensors; the 64x64-tile layers multiply on the bf16 pipe by three-way splits (conv.hip, X3)
    bool bf16 = false;  // VNECT_BF16: bf16 activations + weights, fp32 accumulate; final maps and post-processing stay fp32/f64
    hipStream_t st = nullptr;
    std::map<std::string, HostArray> weights;
    std::vector<Tensor> tensors;
    std::vector<Layer> layers;
    std::map<std::string, int> tensor_by_name;
    int t_input4 = -1, t_out = -1;
    // The stem as one launch (stem.hip): conv1 + pool1 [+ gen_input_batch].  0: off (the stand-alone layers), 1: from the batch
    // tensor (behind pyramid_kernel; also what vnect_forward uses), 2: from the frame (no pyramid launch, no batch tensor).
    int stem_mode = 0;
    bool stem_pair = false;  // the stem launch also runs res2a_branch2a + res2a_branch1 (stem.hip, PAIR): layer l_pool1 + 1 is skipped
    bool stem_frame_ok = false;  // every tile's rectangle of frame bytes fits the kernel's LDS scratch at the current scales
    ScaleTabs stabs_host{};      // the host's copy of d_stabs (plan::stem_frame_fits reads it)
    int l_conv1 = -1, l_pool1 = -1;  // the two layers a stem launch stands for
    StemArgs stem{};
    float* in3 = nullptr;  // (S,368,368,3) staging for vnect_forward / preprocess read-back
    float* ws = nullptr;
    size_t ws_floats = 0;
    std::vector<void*> dev_allocs;
    char* param_cur = nullptr;     // bump allocator over large blocks for packed weights / biases (param_alloc)
    size_t param_left = 0;
    bool keep_activations = true;  // one private buffer per layer output (vnect_read_activation needs it); false = arena
    size_t arena_bytes = 0;
    std::vector<size_t> arena_off;  // byte offset of every tensor in the arena
    // More lanes (cfg.lanes == 2, 3): a frame submitted while others are in flight runs on a twin -- its own stream, activation
    // arena, split-K workspace, arg-max scratch and graph; weights, tables, resident frames, the result ring and the filter
    // bank are this handle's.  The two frames overlap everywhere except in the joints kernel (the filters are a chain).
    std::vector<vnect_handle*> twins;  // lanes 1 .. cfg.lanes-1 (lane 0 is this handle)
    bool is_twin = false;
    long long lane_seq = -1;           // sequence number of the last frame submitted on this lane
    vnect_handle* last_lane = nullptr;  // lane of the most recently submitted frame
    // pre/post
    uint8_t* frames = nullptr;  // num_frame_slots * max_frame_bytes
    // vnect_infer's way from host memory to slot 0: pinned (page-locked) buffers.  [0], [1] are the caller's capture buffers
    // (vnect_frame_buffer; they only move when the caller asks for a larger one): a frame that lies inside one of them is copied to the
    // device straight from there.  Any other pointer is first copied into [2] by the CPU (grown on demand; vnect_infer is synchronous,
    // so one suffices).  The device copy is asynchronous on the frame's stream: nothing synchronises between it and the frame's first kernel.
    uint8_t* stage[3] = {};
    uint8_t* stage_dev[3] = {};  // the same buffers as the device addresses them (hipHostMallocMapped)
    size_t stage_cap[3] = {};
    size_t pre_frame_cap = 0;   // preprocess_only: bytes of the one, growable frame slot
    struct SlotInfo { int H = 0, W = 0; long long stride = 0; long long last_use = -1; };  // last_use: sequence number of the last frame that reads this slot
    std::vector<SlotInfo> slots;
    FrameParams* d_fp = nullptr;   // crop geometry on the device; re-uploaded only when it differs from fp_dev
    FrameParams* h_fp[RING] = {};  // pinned staging for those uploads
    FrameParams fp_dev{};          // what d_fp holds
    bool fp_dev_valid = false;
    int fp_ring = 0;
    ScaleTabs* d_stabs = nullptr;
    MergeGeo mgeo{};  // the merge's resize geometry: passed to the post kernels by value, they compute table entries themselves
    ArgPartial* d_part = nullptr;
    unsigned* d_ticket = nullptr;  // post_kernel's arrival counter (zero between launches)
    bool post_merged = true;       // merge + arg-max + joints as ONE launch (post_kernel); false: two launches (VNECT_NO_POST_MERGE=1)
    FilterBank* d_fb = nullptr;    // [VNECT_MAX_STREAMS]
    double* h_filt = nullptr;      // pinned, device-mapped: vnect_joint_filter's values in ([0, 64)) and out ([64, 128))
    double* h_filt_dev = nullptr;
    JointsOut* h_out[RING] = {};   // pinned, device-mapped: joints_kernel writes a frame's results straight into its ring slot
    JointsOut* h_out_dev[RING] = {};  // the same slots as the device addresses them
    hipEvent_t done[RING] = {};
    unsigned long long seq_submit = 0, seq_collect = 0;
    // per video stream (vnect_submit_stream; stream 0 is what every other entry point uses): d_fb[stream] on the device, and here
    // the host's copy of the last timestamps, the sequence number of the stream's last frame and the lane it ran on
    bool have2[VNECT_MAX_STREAMS] = {}, have3[VNECT_MAX_STREAMS] = {};
    double last2[VNECT_MAX_STREAMS] = {}, last3[VNECT_MAX_STREAMS] = {};
    long long stream_seq[VNECT_MAX_STREAMS] = {-1, -1, -1, -1};
    vnect_handle* stream_lane[VNECT_MAX_STREAMS] = {};
    int ring_stream[RING] = {};
    // cached squarify table
    int sq_H = -1, sq_W = -1;
    FrameParams sq_cache{};
    // graph
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    hipGraph_t pgraph = nullptr;  // profiling twin: same launches, every conv kernel stamps its start/end
    hipGraphExec_t pgexec = nullptr;
    unsigned long long* d_prof = nullptr;       // [layer][2] device stamps (100 MHz)
    unsigned long long* h_prof = nullptr;       // pinned read-back
    unsigned long long* d_prof_end = nullptr;   // [128 layers][PROF_WGS] per-workgroup end stamps of the profiling twin
    unsigned long long* h_prof_end = nullptr;   // pinned read-back
    // profiling
    bool profiling = false;
    hipEvent_t ev[4] = {};
    vnect_timings tim{};
    double conv_flops = 0;
    int conv_launches = 0;
    // comm
    void* comm = nullptr;
    bool sharded = false;
    float* gather = nullptr;  // (S,46,46,84): all ranks' maps
    // exchange by peer writes (vnect_config::exchange == VNECT_XCHG_P2P; kernels.h: XchgArgs)
    char* xblock = nullptr;            // this rank's exchange block (fine-grained device memory, IPC-exported)
    char* xpeer[VNECT_MAX_SCALES] = {};  // every rank's block as this device addresses it; [rank] == xblock
    bool xopened[VNECT_MAX_SCALES] = {};  // xpeer[r] came from hipIpcOpenMemHandle (close it on destroy)
    bool p2p_ready = false;
    unsigned* xtickets = nullptr;
    int* h_xstatus = nullptr;          // pinned, device-mapped, one word per result-ring slot: a peer's flag did not arrive within the bound
    int* h_xstatus_dev = nullptr;
    unsigned* d_xfail = nullptr;       // device word: sequence number of the last frame whose exchange failed (post_kernel skips its joints stage)
};

namespace vnect {
namespace rt {

// message of the last vnect_create failure on THIS thread (handles are created from several threads / processes); rt_abi.cpp
extern thread_local std::string g_create_error;

inline int fail(vnect_handle* h, int code, const std::string& msg) noexcept
{
    try {
        if (h) h->err = msg;
        else g_create_error = msg;
    } catch (...) {  // out of memory while recording the message: the code still goes back
    }
    return code;
}

// No C++ exception crosses the ABI: every extern "C" body runs inside this guard (std::vector / std::string / std::map / new
// can throw std::bad_alloc or std::length_error on a hostile size).
template <typename F>
int guarded(vnect_handle* const* hp, F&& body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(hp ? *hp : nullptr, VNECT_E_INTERNAL, "host allocation failed");
    } catch (const std::exception& e) {
        return fail(hp ? *hp : nullptr, VNECT_E_INTERNAL, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(hp ? *hp : nullptr, VNECT_E_INTERNAL, "internal error");
    }
}

#define HIPCK(h, expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(h, VNECT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

template <typename T>
int dev_alloc(vnect_handle* h, T** p, size_t count)
{
    void* q = nullptr;
    HIPCK(h, hipMalloc(&q, std::max<size_t>(count * sizeof(T), 16)));
    h->dev_allocs.push_back(q);
    *p = (T*)q;
    return VNECT_OK;
}

// Parameters (packed weights, biases, BN vectors) are carved out of a few large blocks instead of ~150 separate
// allocations: contiguous, 256-byte aligned, and mapped with large page fragments, so a layer's first touch of its
// weights does not start with a page-table walk per 4 KiB.
template <typename T>
int param_alloc(vnect_handle* h, T** p, size_t count)
{
    const size_t need = (std::max<size_t>(count * sizeof(T), 16) + 255) & ~(size_t)255;
    if (h->param_left < need) {
        const size_t block = std::max<size_t>(need, (size_t)32 << 20);
        char* q = nullptr;
        int rc = dev_alloc(h, &q, block);
        if (rc) return rc;
        h->param_cur = q, h->param_left = block;
    }
    *p = (T*)h->param_cur;
    h->param_cur += need, h->param_left -= need;
    return VNECT_OK;
}

template <typename T>
int upload(vnect_handle* h, T** dst, const std::vector<T>& v)
{
    int rc = param_alloc(h, dst, v.size());
    if (rc) return rc;
    HIPCK(h, hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return VNECT_OK;
}

// ---- rt_plan.cpp ---------------------------------------------------------------------------------------------------------------
int build_scale_tables(vnect_handle* h);
int build_up_table(vnect_handle* h);
int squarify_params(vnect_handle* h, int H, int W, FrameParams* fp);
int add_tensor(vnect_handle* h, const std::string& name, int S, int H, int W, int C, int Cs, bool force_f32 = false);
void bind_activations(vnect_handle* h, Layer& L);
void setup_stem(vnect_handle* h);
int finalize_impl(vnect_handle* h);

// ---- rt_exec.cpp ---------------------------------------------------------------------------------------------------------------
int run_network(vnect_handle* h, bool timed, bool stem_done = false);
int sync_geometry(vnect_handle* h, const FrameParams& fp);
int run_pre(vnect_handle* h, const FrameDyn& dyn, bool timed = false, bool want_batch = false);
int run_argmax(vnect_handle* h);
int run_joints(vnect_handle* h, const FrameDyn& dyn, JointsOut* out, int stream = 0);
int run_post(vnect_handle* h, const FrameDyn& dyn, JointsOut* out, int stream = 0);
int check_time(vnect_handle* h, double t2d, double t3d, int s = 0);
void commit_time(vnect_handle* h, double t2d, double t3d, int s = 0);
int reset_filters_impl(vnect_handle* h, int stream = -1);  // -1: every stream
void roctx_load();
int build_graph(vnect_handle* h);
void destroy_twins(vnect_handle* h);
int build_twins(vnect_handle* h);
int enqueue_frame(vnect_handle* h, int slot, double t2d, double t3d, int* ring_out, int stream = 0);
int collect_impl(vnect_handle* h, double* j2, float* j3, int32_t* stream_out = nullptr);
int ensure_stage(vnect_handle* h, int i, size_t bytes);
int stage_frame(vnect_handle* h, int slot, const uint8_t* bgr, int H, int W, int64_t row_stride);
int upload_frame_impl(vnect_handle* h, int slot, const uint8_t* bgr, int H, int W, int64_t row_stride);
int prime(vnect_handle* h);

// ---- rt_comm.cpp ---------------------------------------------------------------------------------------------------------------
int exchange_maps(vnect_handle* h, unsigned long long seq, int ring);
bool comm_ready(const vnect_handle* h);
void comm_destroy(vnect_handle* h);  // the handle's RCCL communicator, if it has one

}  // namespace rt
}  // namespace vnect
