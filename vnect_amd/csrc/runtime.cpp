// runtime.cpp -- host side of libvnect_hip.so: handle, weight packing, launch plan, HIP graph, C ABI.
// See include/vnect_abi.h for the boundary and the reference lines each entry point replaces.
#include <dlfcn.h>
#include <link.h>
#include <unistd.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vnect_abi.h"
#include "hostplan.h"
#include "kernels.h"

using namespace vnect;

namespace {

// message of the last vnect_create failure on THIS thread (handles are created from several threads / processes)
thread_local std::string g_create_error = "";

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

}  // namespace

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

namespace {

int fail(vnect_handle* h, int code, const std::string& msg) noexcept
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

using plan::from_bf16;
using plan::to_bf16;
// packed weights: fp32 as is, or converted to bf16 (the device pointer is typed float* either way)
int upload_weights(vnect_handle* h, float** dst, const std::vector<float>& v)
{
    if (!h->bf16) return upload(h, dst, v);
    std::vector<uint16_t> b(v.size());
    for (size_t i = 0; i < v.size(); i++) b[i] = to_bf16(v[i]);
    uint16_t* p = nullptr;
    int rc = upload(h, &p, b);
    *dst = (float*)p;
    return rc;
}

// Weights of a layer in the layout its launch reads: the split-product form (conv.hip, X3) for 64x64-tile layers of a VNECT_FP32_SPLIT
// handle -- three bf16 planes per row and 32-element K chunk -- else upload_weights.  Sets a.x3.
int upload_layer_weights(vnect_handle* h, Layer& L, const std::vector<float>& wp)
{
    ConvArgs& a = L.a;
    a.x3 = h->x3 && !h->bf16 && L.BM == 64 && L.BN * L.KG == 64 && !a.pixmode && a.K % 32 == 0 && !getenv("VNECT_NO_X3");
    if (!a.x3) return upload_weights(h, &L.w, wp);
    std::vector<uint16_t> pl;
    plan::pack_split3(wp, std::max(a.nphase, 1), a.Npad, a.K, pl);  // (the transposed conv: 4 phases)
    uint16_t* p = nullptr;
    int rc = upload(h, &p, pl);
    L.w = (float*)p;
    return rc;
}

// ---- tables: the arithmetic is in hostplan.h (HIP-free, sanitizer-tested on the CPU box); here they are built and uploaded ----
int build_scale_tables(vnect_handle* h)
{
    ScaleTabs st;
    memset(&st, 0, sizeof st);
    MergeGeo mg;
    memset(&mg, 0, sizeof mg);
    st.S = mg.S = h->S;
    plan::fill_lut(st.lut);
    for (int i = 0; i < h->S; i++) {
        const double s = h->cfg.scales[i];
        if (const char* why = plan::build_scale_tab(s, &st, i)) return fail(h, VNECT_E_ARG, why);
        if (const char* why = plan::build_merge_geo(s, &mg, i)) return fail(h, VNECT_E_ARG, why);
    }
    HIPCK(h, hipMemcpy(h->d_stabs, &st, sizeof st, hipMemcpyHostToDevice));
    h->mgeo = mg;
    h->stabs_host = st;
    // the stem's from-the-frame form: do all tiles' frame rectangles still fit its LDS scratch at these scales?  (lanes share the tables)
    if (h->stem_mode == 2) h->stem_frame_ok = plan::stem_frame_fits(st, h->stem.S, h->stem.scale_base, h->stem.groups, h->stem.row0, h->bf16);
    for (vnect_handle* tw : h->twins) tw->stabs_host = st, tw->mgeo = mg, tw->stem_frame_ok = h->stem_frame_ok;
    return VNECT_OK;
}

int build_up_table(vnect_handle* h)
{
    std::vector<UpTab> u(1);
    // the arg-max kernels walk the rows of the x8 upsample by (segment, phase) and compute the entries themselves: plan::build_up_tab
    // builds the full table with the same formulas and checks that it really has that structure
    if (!plan::build_up_tab(&u[0])) return fail(h, VNECT_E_STATE, "internal: x8 upsample table does not have the segment/phase structure");
    return VNECT_OK;
}

// utils.img_scale_squarify + img_padding geometry for an (H,W) frame
int squarify_params(vnect_handle* h, int H, int W, FrameParams* fp)
{
    if (h->sq_H != H || h->sq_W != W) {
        FrameParams c;
        if (const char* why = plan::squarify(H, W, &c)) return fail(h, VNECT_E_ARG, why);
        h->sq_cache = c, h->sq_H = H, h->sq_W = W;
    }
    *fp = h->sq_cache;
    return VNECT_OK;
}

// ---- network construction --------------------------------------------------------------------------
int add_tensor(vnect_handle* h, const std::string& name, int S, int H, int W, int C, int Cs, bool force_f32 = false)
{
    Tensor t;
    t.name = name, t.S = S, t.H = H, t.W = W, t.C = C, t.Cs = Cs;
    t.esz = (h->bf16 && !force_f32) ? 2 : 4;
    h->tensors.push_back(t);
    h->tensor_by_name[name] = (int)h->tensors.size() - 1;
    return (int)h->tensors.size() - 1;
}

const HostArray* get_w(vnect_handle* h, const std::string& name, std::vector<int64_t> shape)
{
    auto it = h->weights.find(name);
    if (it == h->weights.end()) {
        h->err = "missing weight array " + name;
        return nullptr;
    }
    if (it->second.shape != shape) {
        h->err = "weight " + name + " has the wrong shape";
        return nullptr;
    }
    return &it->second;
}

using plan::round_up;
using plan::same_pad;

// Tile and K-split of a layer: plan::choose_tile (hostplan.h) with the tuning overrides VNECT_FORCE_TILE / VNECT_PLAN.
void choose_tile(Layer& L, long long npix, bool allow96 = false)
{
    (void)npix;
    // (allow96: the transposed conv's three-accumulator shape, fp32 instruction path only -- not on a split-product handle, whose 64x64
    // split-product loop is the faster one for this layer; VNECT_NO_DECONV96=1 restores the 64x64 plan for A/B runs)
    const plan::TileChoice c = plan::choose_tile(L.a.M, L.Nreal, L.a.ntaps, L.a.cpt, L.a.K, L.a.nphase, L.a.bf16 != 0, L.name,
                                                 getenv("VNECT_FORCE_TILE"), getenv("VNECT_PLAN"), allow96 && conv_deconv96_available() && !getenv("VNECT_NO_DECONV96"));
    L.BM = c.BM, L.BN = c.BN, L.KG = c.KG, L.a.ksplit = c.ks;
}

struct ConvSpec {
    std::string scope, out_name;
    int in = -1, resid = -1;
    int k = 1, stride = 1, cout = 0;
    bool relu = false;
};

// tc.layers.conv2d scope -> Layer (weights HWIO + bias); returns the output tensor index or -1
int add_conv(vnect_handle* h, const ConvSpec& sp)
{
    const Tensor tin = h->tensors[sp.in];
    const bool conv1 = sp.k == 7;
    const int cin = tin.C;
    const HostArray* W = get_w(h, sp.scope + "/weights", {sp.k, sp.k, cin, sp.cout});
    const HostArray* B = W ? get_w(h, sp.scope + "/biases", {sp.cout}) : nullptr;
    if (!W || !B) return -1;
    int ho, wo, pt = 0, pl = 0;
    if (sp.k == 1) ho = (tin.H - 1) / sp.stride + 1, wo = (tin.W - 1) / sp.stride + 1;  // VALID
    else same_pad(tin.H, sp.k, sp.stride, &ho, &pt), same_pad(tin.W, sp.k, sp.stride, &wo, &pl);
    Layer L;
    L.op = OP_CONV, L.name = sp.scope, L.in = sp.in, L.resid = sp.resid;
    const bool final_maps = sp.scope == "res5c_branch2c";  // feeds the f64 post-processing: stays fp32
    L.out = add_tensor(h, sp.out_name, tin.S, ho, wo, sp.cout, sp.cout, final_maps);
    ConvArgs& a = L.a;
    a.out_f32 = final_maps;
    a.S = tin.S, a.H = tin.H, a.W = tin.W, a.Cs = tin.Cs;
    a.Ho = ho, a.Wo = wo, a.M = tin.S * ho * wo;
    a.stride = sp.stride;
    a.OH = ho, a.OW = wo, a.os = 1, a.nphase = 1;
    a.ldc = sp.cout, a.ldr = sp.cout;
    a.relu_cols = sp.relu ? sp.cout : 0;
    a.Nvalid = sp.cout;
    L.Nreal = sp.cout;
    const int EPR = h->bf16 ? 64 : 32;  // K-elements per chunk
    a.bf16 = h->bf16;
    int cp;  // channels per tap in the packed K
    if (conv1) {
        // fp32: K = 7 rows x (8 pixels x 4 channels); bf16: K = 4 row pairs x (2 rows x 8 pixels x 4 channels).
        // Pixel 7, channel 3 (and row 7 in bf16) carry zero weights.
        a.pixmode = 1, a.cpt = 1, cp = 32;
        a.ntaps = h->bf16 ? 4 : 7;
        for (int t = 0; t < a.ntaps; t++) L.dy[t] = (int)((h->bf16 ? 2 * t : t) - pt), L.dx[t] = (int)(-pl);
    } else {
        cp = round_up(tin.Cs, EPR);
        if (cp != tin.Cs) {
            h->err = "internal: input channel stride not a multiple of the chunk at " + sp.scope;
            return -1;
        }
        a.pixmode = 0, a.ntaps = sp.k * sp.k, a.cpt = cp / EPR;
        for (int ky = 0; ky < sp.k; ky++)
            for (int kx = 0; kx < sp.k; kx++) L.dy[ky * sp.k + kx] = (int)(ky - pt), L.dx[ky * sp.k + kx] = (int)(kx - pl);
    }
    a.K = a.ntaps * a.cpt * EPR;
    L.Kreal = sp.k * sp.k * cin;
    L.flops = 2.0 * a.M * (double)L.Kreal * sp.cout;
    choose_tile(L, (long long)a.M);
    a.Npad = round_up(sp.cout, L.BN);
    std::vector<float> wp((size_t)a.Npad * a.K, 0.f), bp(a.Npad, 0.f);
    plan::pack_conv(W->d.data(), sp.k, cin, sp.cout, cp, conv1, h->bf16, a.K, 0, wp);
    for (int n = 0; n < sp.cout; n++) bp[n] = B->d[n];
    if (upload_layer_weights(h, L, wp) || upload(h, &L.bias, bp)) return -1;
    h->layers.push_back(L);
    return L.out;
}

// Two convs of ONE input with one kernel size and stride as ONE launch: weights concatenated along N ([a | b]), two output
// tensors, ReLU per column block.  Used for (i) branch2a (ReLU) + branch1 (none), the two 1x1 convs at the head of a projection
// block (vnect_model.py:32-35,64-67,106-109,168-175), and (ii) res2b_branch2b + res2c_branch2b, two 3x3 convs that both read
// res2b_branch2a in the reference's wiring (vnect_model.py:50,56).  Returns the first tensor, *second gets the other one.
//
// Head split (round 5): a pair whose 64x64 tiles need one round over the CUs more than their matrix work does -- res5a_branch2a_new +
// res5a_branch1_new at three scales: 600 tiles = 2.34 per CU = THREE block K loops per SIMD -- runs the first `head` channels of layer a
// as a launch of their own in a K-group shape (64x32x2: half a K loop per SIMD) and the rest as the pair: 500 tiles = two rounds, 2.5 K
// loops in all; plan::pair_head_cols weighs that against the extra launch.  Both launches write the same tensors (the pair from channel
// `head` on: Layer::out_col0); the pair's channels keep their tiles and K order (bit-identical to the single launch), the head's are
// summed by two K groups like every 64x32x2 layer (equal to fp32 rounding).  VNECT_NO_HEAD_SPLIT=1: A/B runs and the parity test.
int add_conv_pair(vnect_handle* h, const std::string& sa, int cout_a, const std::string& sb, int cout_b, int in,
                  int stride, int* second, int k = 1, bool relu_b = false)
{
    const Tensor tin = h->tensors[in];
    const int cin = tin.C;
    const HostArray* Wa = get_w(h, sa + "/weights", {k, k, cin, cout_a});
    const HostArray* Ba = Wa ? get_w(h, sa + "/biases", {cout_a}) : nullptr;
    const HostArray* Wb = Ba ? get_w(h, sb + "/weights", {k, k, cin, cout_b}) : nullptr;
    const HostArray* Bb = Wb ? get_w(h, sb + "/biases", {cout_b}) : nullptr;
    if (!Bb) return -1;
    const int EPR = h->bf16 ? 64 : 32;
    if (cout_a % 64 || tin.Cs % EPR || (k != 1 && (k != 3 || stride != 1))) {
        h->err = "internal: paired conv needs a 64-aligned split, whole K chunks and 1x1 or 3x3 stride 1";
        return -1;
    }
    int ho = (tin.H - 1) / stride + 1, wo = (tin.W - 1) / stride + 1, pt = 0, pl = 0;  // 1x1: VALID
    if (k == 3) same_pad(tin.H, 3, 1, &ho, &pt), same_pad(tin.W, 3, 1, &wo, &pl);
    Layer L;
    L.op = OP_CONV, L.name = sa + "+" + (k == 1 ? sb.substr(sb.find('_') + 1) : sb), L.in = in;
    L.out = add_tensor(h, sa, tin.S, ho, wo, cout_a, cout_a);
    L.out2 = add_tensor(h, sb, tin.S, ho, wo, cout_b, cout_b);
    ConvArgs& a = L.a;
    a.S = tin.S, a.H = tin.H, a.W = tin.W, a.Cs = tin.Cs;
    a.Ho = ho, a.Wo = wo, a.M = tin.S * ho * wo, a.stride = stride;
    a.OH = ho, a.OW = wo, a.os = 1, a.nphase = 1;
    a.ldc = cout_a, a.ldc2 = cout_b, a.split_n = cout_a, a.ldr = 0;
    a.relu_cols = relu_b ? cout_a + cout_b : cout_a;
    a.ntaps = k * k, a.cpt = tin.Cs / EPR, a.K = k * k * tin.Cs;
    for (int ky = 0; ky < k; ky++)
        for (int kx = 0; kx < k; kx++) L.dy[ky * k + kx] = ky - pt, L.dx[ky * k + kx] = kx - pl;
    a.bf16 = h->bf16;
    L.Nreal = cout_a + cout_b, L.Kreal = k * k * cin;
    a.Nvalid = L.Nreal;
    L.flops = 2.0 * a.M * (double)L.Kreal * L.Nreal;
    choose_tile(L, (long long)a.M);
    if (L.BN != 64 || L.a.ksplit != 1 || L.KG != 1) L.BM = 64, L.BN = 64, L.KG = 1, L.a.ksplit = 1;  // the column split relies on 64-wide tiles, no slabs
    a.Npad = round_up(L.Nreal, 64);
    std::vector<float> wp((size_t)a.Npad * a.K, 0.f), bp(a.Npad, 0.f);
    plan::pack_conv(Wa->d.data(), k, cin, cout_a, tin.Cs, false, h->bf16, a.K, 0, wp);
    plan::pack_conv(Wb->d.data(), k, cin, cout_b, tin.Cs, false, h->bf16, a.K, cout_a, wp);
    for (int n = 0; n < cout_a; n++) bp[n] = Ba->d[n];
    for (int n = 0; n < cout_b; n++) bp[cout_a + n] = Bb->d[n];
    const int head = (k == 1 && !relu_b && !h->x3 && !getenv("VNECT_NO_HEAD_SPLIT") && !getenv("VNECT_FORCE_TILE"))
                         ? plan::pair_head_cols(a.M, cout_a, cout_b, a.K, h->bf16) : 0;
    if (head) {
        Layer Hd;  // channels [0, head) of layer a: an ordinary 1x1 launch into the same tensor
        Hd.op = OP_CONV, Hd.name = sa + "[:" + std::to_string(head) + "]", Hd.in = in, Hd.out = L.out;
        ConvArgs& q = Hd.a;
        q = a;
        q.ldc2 = 0, q.split_n = 0, q.relu_cols = head, q.Nvalid = head;
        Hd.dy[0] = Hd.dx[0] = 0;
        Hd.Nreal = head, Hd.Kreal = L.Kreal, Hd.flops = 2.0 * a.M * (double)L.Kreal * head;
        choose_tile(Hd, (long long)a.M);
        q.Npad = round_up(head, Hd.BN);
        std::vector<float> wh(wp.begin(), wp.begin() + (size_t)head * a.K), bh(bp.begin(), bp.begin() + head);
        wh.resize((size_t)q.Npad * a.K, 0.f), bh.resize(q.Npad, 0.f);
        if (upload_layer_weights(h, Hd, wh) || upload(h, &Hd.bias, bh)) return -1;
        h->layers.push_back(Hd);
        // ... and the pair keeps the rest
        wp.erase(wp.begin(), wp.begin() + (size_t)head * a.K), bp.erase(bp.begin(), bp.begin() + head);
        L.out_col0 = head, L.Nreal -= head, a.Nvalid = L.Nreal, a.Npad -= head;
        a.split_n = cout_a - head, a.relu_cols = cout_a - head;
        L.flops = 2.0 * a.M * (double)L.Kreal * L.Nreal;
    }
    if (upload_layer_weights(h, L, wp) || upload(h, &L.bias, bp)) return -1;
    if (k == 1 && stride == 1 && cin == 64 && tin.Cs == 64 && cout_a == 64 && cout_b == 256) {
        // res2a_branch2a + res2a_branch1 read pool1: the stem can run them on its pooled tile (setup_stem) and wants the weights in
        // fragment order, both layers side by side ([k][n] with n over the 64 + 256 outputs)
        std::vector<float> cat((size_t)64 * 320), fw;
        for (int kk = 0; kk < 64; kk++) {
            for (int n = 0; n < 64; n++) cat[(size_t)kk * 320 + n] = Wa->d[(size_t)kk * 64 + n];
            for (int n = 0; n < 256; n++) cat[(size_t)kk * 320 + 64 + n] = Wb->d[(size_t)kk * 256 + n];
        }
        plan::pack_tail(cat.data(), 64, 320, h->bf16, fw);
        if (upload_weights(h, &L.frag_w, fw)) return -1;
    }
    h->layers.push_back(L);
    *second = L.out2;
    return L.out;
}

// A 3x3 conv whose 64 output channels feed a 1x1 conv (+ shortcut + ReLU) of the same pixels -- res2*_branch2b -> res2*_branch2c,
// vnect_model.py:38-41,50-53,56-59 -- as ONE launch: the 3x3 layer's tile stays in LDS and the 1x1 layer is a second GEMM
// inside the workgroup (conv.hip, TAIL).  Possible where a workgroup owns ALL of the 3x3 layer's channels for its rows (N = 64:
// the 92x92 stage) and has one tile (so the ring is free behind the K loop): ceil(M / 64) <= 512 workgroups.  Results are
// bit-identical to the two stand-alone launches.  Returns the block output tensor or -1; *fits = false if the shape does not
// admit the fusion (the caller then builds the two layers).
//
// The WIDE form (round 3) does the same for a 3x3 layer with 128 channels -- res3*_branch2b -> res3*_branch2c and the head's
// res5c_branch2b -> res5c_branch2c (vnect_model.py:62-103,211-217): the 3x3 layer runs on 32 x 128 tiles (as many workgroups as the
// 64 x 64 plan has tiles, the same MFMA work per wave), a workgroup owns all 128 channels of its 32 pixels, the 1x1 layer (K = 128,
// up to 512 outputs) is conv.hip's tail_wide.  One workgroup per CU (100-KB ring): ceil(M / 32) <= 256, i.e. up to three scales at
// 46x46.  Not on a split-product handle (its 3x3 layers keep the 64 x 64 split-product loop).  VNECT_NO_WIDE_TAIL=1: A/B runs.
int add_conv_tail(vnect_handle* h, const std::string& sb, const std::string& sc, int in, int resid, const std::string& out_name,
                  int mid, int cout, bool* fits, bool relu2 = true, const std::string& chain_scope = "", int* chain_out = nullptr)
{
    const Tensor tin = h->tensors[in];
    const int EPR = h->bf16 ? 64 : 32;
    const long long pixels = (long long)tin.S * tin.H * tin.W;
    const bool narrow = mid == 64 && cout == 256 && (pixels + 63) / 64 <= 512;
    // ... and at least half a chip of them in fp32: one scale at 46x46 is 67 workgroups, each alone with the whole 3x3 K loop where the
    // stand-alone layer splits K over the idle CUs -- measured, single-scale handle (a pyramid rank): 0.670 ms per frame with the wide
    // tails against 0.619 without; two scales (133): level; bf16: level at one scale, +5 % at two (tools/one_scale_ab.py).
    const long long wide_wgs = (pixels + 31) / 32;
    const bool wide = mid == 128 && cout <= 512 && wide_wgs <= 256 && (h->bf16 || wide_wgs >= 128 || getenv("VNECT_FORCE_WIDE_TAIL")) && !h->x3 &&
                      !getenv("VNECT_NO_WIDE_TAIL") && !getenv("VNECT_FORCE_TILE");
    *fits = (narrow || wide) && tin.Cs % EPR == 0 && !h->keep_activations && !getenv("VNECT_NO_TAIL");
    if (!*fits) return -1;
    const int cin = tin.C;
    const HostArray* Wb = get_w(h, sb + "/weights", {3, 3, cin, mid});
    const HostArray* Bb = Wb ? get_w(h, sb + "/biases", {mid}) : nullptr;
    const HostArray* Wc = Bb ? get_w(h, sc + "/weights", {1, 1, mid, cout}) : nullptr;
    const HostArray* Bc = Wc ? get_w(h, sc + "/biases", {cout}) : nullptr;
    if (!Bc) return -1;
    int ho, wo, pt = 0, pl = 0;
    same_pad(tin.H, 3, 1, &ho, &pt), same_pad(tin.W, 3, 1, &wo, &pl);
    Layer L;
    L.op = OP_CONV, L.name = sb + ">" + sc, L.in = in, L.resid = resid;
    const bool final_maps = sc == "res5c_branch2c";  // feeds the f64 post-processing: stays fp32
    L.out = add_tensor(h, out_name, tin.S, ho, wo, cout, cout, final_maps);
    ConvArgs& a = L.a;
    a.out_f32 = final_maps;
    a.S = tin.S, a.H = tin.H, a.W = tin.W, a.Cs = tin.Cs;
    a.Ho = ho, a.Wo = wo, a.M = tin.S * ho * wo, a.stride = 1;
    a.OH = ho, a.OW = wo, a.os = 1, a.nphase = 1;
    a.ldc = cout, a.ldr = cout, a.relu_cols = relu2 ? cout : 0, a.Nvalid = cout;  // the TAIL's output, shortcut and ReLU
    a.bf16 = h->bf16;
    a.ntaps = 9, a.cpt = tin.Cs / EPR, a.K = 9 * tin.Cs;
    for (int ky = 0; ky < 3; ky++)
        for (int kx = 0; kx < 3; kx++) L.dy[ky * 3 + kx] = ky - pt, L.dx[ky * 3 + kx] = kx - pl;
    L.Nreal = mid, L.Kreal = 9 * cin;
    L.flops = 2.0 * a.M * ((double)L.Kreal * mid + (double)mid * cout);
    L.BM = wide ? 32 : 64, L.BN = mid, L.KG = 1, a.ksplit = 1;
    a.Npad = mid;
    std::vector<float> wp((size_t)mid * a.K, 0.f), bp(mid, 0.f), w2, b2(round_up(cout, 32), 0.f);
    plan::pack_conv(Wb->d.data(), 3, cin, mid, tin.Cs, false, h->bf16, a.K, 0, wp);
    for (int n = 0; n < mid; n++) bp[n] = Bb->d[n];
    plan::pack_tail(Wc->d.data(), mid, cout, h->bf16, w2);
    for (int n = 0; n < cout; n++) b2[n] = Bc->d[n];
    float *dw2 = nullptr, *db2 = nullptr;
    if (upload_layer_weights(h, L, wp) || upload(h, &L.bias, bp) || upload_weights(h, &dw2, w2) || upload(h, &db2, b2)) return -1;
    a.tail_w = dw2, a.tail_bias = db2, a.tail_n = cout;
    // Chain GEMM (conv.hip: chain_gemm): the NEXT block's branch2a (1x1, 512 -> 128, ReLU) on the output tile while it is still in LDS --
    // one launch fewer per identity block of the 46x46 stage.  bf16 only: there a launch is mostly fixed cost (+2.9 % frames/s, A/B in
    // one call); in fp32 the chained layer is MFMA-bound either way (7.6 us of matrix work on the same four SIMDs) and the chain's own
    // overhead exceeds the launch it saves (-1.5 %).  VNECT_NO_CHAIN=1 / VNECT_FORCE_CHAIN=1: A/B runs and the fp32 form's parity test.
    if (chain_out) *chain_out = -1;
    // The 64-wide tail chains too (res2a -> res2b_branch2a, 256 -> 64; conv.hip: chain_narrow), in bf16 only: its output tile would not
    // fit the two-workgroups-per-CU ring in fp32.
    const bool chain_wide = wide && cout == 512 && (h->bf16 || getenv("VNECT_FORCE_CHAIN"));
    const bool chain_narrow = narrow && !wide && cout == 256 && h->bf16 && resid >= 0;
    if ((chain_wide || chain_narrow) && relu2 && !chain_scope.empty() && chain_out && !getenv("VNECT_NO_CHAIN")) {
        const int cn = chain_wide ? 128 : 64;
        const HostArray* Wn = get_w(h, chain_scope + "/weights", {1, 1, cout, cn});
        const HostArray* Bn = Wn ? get_w(h, chain_scope + "/biases", {cn}) : nullptr;
        if (!Bn) return -1;
        std::vector<float> w3, b3(Bn->d.begin(), Bn->d.end());
        plan::pack_tail(Wn->d.data(), cout, cn, h->bf16, w3);
        float *dw3 = nullptr, *db3 = nullptr;
        if (upload_weights(h, &dw3, w3) || upload(h, &db3, b3)) return -1;
        L.out3 = add_tensor(h, chain_scope, tin.S, ho, wo, cn, cn);
        a.chain_w = dw3, a.chain_bias = db3, a.chain_n = cn, a.chain_ld = cn;
        L.name += ">" + chain_scope;
        L.flops += 2.0 * a.M * (double)cout * cn;
        *chain_out = L.out3;
    }
    h->layers.push_back(L);
    return L.out;
}

// point a conv layer's arguments at h's activation buffers and workspace (weights are whatever L already holds)
void bind_activations(vnect_handle* h, Layer& L)
{
    ConvArgs& a = L.a;
    a.in = h->tensors[L.in].d;
    a.out = (float*)((char*)h->tensors[L.out].d + (size_t)L.out_col0 * h->tensors[L.out].esz);
    a.out2 = L.out2 >= 0 ? h->tensors[L.out2].d : nullptr;
    a.chain_out = L.out3 >= 0 ? h->tensors[L.out3].d : nullptr;
    a.resid = L.resid >= 0 ? h->tensors[L.resid].d : nullptr;
    a.w = L.w, a.bias = L.bias, a.scale = L.scale, a.shift = L.shift, a.ws = h->ws;
    L.r.ws = h->ws, L.r.resid = a.resid, L.r.out = a.out;
}

// The fused stem (stem.hip) stands for layers l_conv1 + l_pool1 (and, from the frame, for pyramid_kernel).  Default: from the frame
// on handles whose layers share the arena; handles with per-layer read-back keep the stand-alone layers (their "conv1" activation
// must exist) unless VNECT_FORCE_STEM says otherwise (the parity test reads pool1 from both forms).  VNECT_NO_STEM=1 restores the
// three launches, VNECT_STEM=batch keeps pyramid_kernel and fuses conv1 + pool1 only (A/B runs).
void setup_stem(vnect_handle* h)
{
    h->stem_mode = 0, h->stem_pair = false;
    if (h->l_conv1 < 0 || h->l_pool1 != h->l_conv1 + 1) return;
    const Layer& C = h->layers[h->l_conv1];
    const Tensor& tin = h->tensors[h->t_input4];
    const Tensor& tp = h->tensors[h->layers[h->l_pool1].out];
    if (tin.H != BOX || tin.W != BOX || tp.H != 92 || tp.W != 92 || tp.Cs != 64 || C.a.Npad != 64 || C.a.ksplit != 1) return;
    const char* force = getenv("VNECT_FORCE_STEM");
    const char* mode = getenv("VNECT_STEM");
    if (getenv("VNECT_NO_STEM")) return;
    if (h->keep_activations && !force) return;
    if (force) mode = force;
    h->stem_mode = (mode && !strcmp(mode, "batch")) ? 1 : 2;
    StemArgs& a = h->stem;
    memset(&a, 0, sizeof a);
    a.batch = tin.d, a.w = C.w, a.bias = C.bias, a.out = tp.d;
    a.fp = h->d_fp, a.tabs = h->d_stabs;
    a.S = h->Snet, a.scale_base = h->sharded ? h->cfg.pyramid_rank : 0, a.bf16 = h->bf16;
    // row groups of 4 and 5 pooled rows (hostplan.h)
    a.groups = plan::stem_groups(a.S, a.row0);
    // (a lane's tables are lane 0's: build_twin copies stabs_host before calling this)
    h->stem_frame_ok = h->stem_mode == 2 && plan::stem_frame_fits(h->stabs_host, a.S, a.scale_base, a.groups, a.row0, h->bf16);
    // PAIR form: the launch behind pool1 is res2a_branch2a + res2a_branch1 (1x1 on pool1's 64 channels) and nothing else reads pool1:
    // the stem runs it on the pooled tile and pool1 is never written.  Not with per-layer read-back (pool1 must exist there).
    // VNECT_NO_STEM_PAIR=1: A/B runs.
    h->stem_pair = false;
    const size_t lp = (size_t)h->l_pool1 + 1;
    if (!h->keep_activations && !getenv("VNECT_NO_STEM_PAIR") && lp < h->layers.size()) {
        const Layer& P = h->layers[lp];
        if (P.op == OP_CONV && P.frag_w && P.in == h->layers[h->l_pool1].out && P.out >= 0 && P.out2 >= 0 && P.a.Npad == 320 &&
            P.a.split_n == 64 && P.a.relu_cols == 64 && P.a.ldc == 64 && P.a.ldc2 == 256 && P.a.M == a.S * 92 * 92) {
            a.pair_w = P.frag_w, a.pair_bias = P.bias, a.pair_out_a = h->tensors[P.out].d, a.pair_out_b = h->tensors[P.out2].d;
            h->stem_pair = true;
        }
    }
}

int finalize_impl(vnect_handle* h)
{
    const int S = h->Snet;
    h->tensors.clear(), h->layers.clear(), h->tensor_by_name.clear();
    h->t_input4 = add_tensor(h, "input", S, BOX, BOX, 3, 4);
    auto conv = [&](const std::string& scope, int in, int k, int stride, int cout, bool relu, int resid = -1,
                    const std::string& out_name = "") {
        ConvSpec sp;
        sp.scope = scope, sp.out_name = out_name.empty() ? scope : out_name;
        sp.in = in, sp.resid = resid, sp.k = k, sp.stride = stride, sp.cout = cout, sp.relu = relu;
        return add_conv(h, sp);
    };
#define NEED(x)                                 \
    do {                                        \
        if ((x) < 0) return VNECT_E_ARG;        \
    } while (0)
    // vnect_model.py:27-29
    int conv1 = conv("conv1", h->t_input4, 7, 2, 64, true);
    NEED(conv1);
    int pool1;
    {
        const Tensor t = h->tensors[conv1];
        int ho, wo, p;
        same_pad(t.H, 3, 2, &ho, &p), same_pad(t.W, 3, 2, &wo, &p);
        Layer L;
        L.op = OP_POOL, L.name = "pool1", L.in = conv1;
        pool1 = L.out = add_tensor(h, "pool1", S, ho, wo, 64, 64);
        h->l_conv1 = (int)h->layers.size() - 1;
        h->layers.push_back(L);
        h->l_pool1 = (int)h->layers.size() - 1;
    }
    // bottleneck blocks (vnect_model.py:31-165); block output tensors are named resNx
    // branch2b (3x3) -> branch2c (1x1, + shortcut s, ReLU): one launch where the tail GEMM fits (add_conv_tail), else two
    // (`next`: the identity block behind this one -- where the fused launch takes the wide form, that block's branch2a rides along
    // as its chain GEMM and `chained` holds its output tensor for ident() to pick up)
    int chained = -1;
    std::string chained_for;
    auto b_then_c = [&](const std::string& p, int a, int mid, int out, int s, const std::string& next = "") {
        if (a < 0) return -1;
        bool fits = false;
        int co = -1;
        const int o = add_conv_tail(h, p + "_branch2b", p + "_branch2c", a, s, p, mid, out, &fits, true, next.empty() ? "" : next + "_branch2a", &co);
        if (fits && co >= 0) chained = co, chained_for = next;
        if (fits) return o;
        const int b = conv(p + "_branch2b", a, 3, 1, mid, true);
        return b < 0 ? -1 : conv(p + "_branch2c", b, 1, 1, out, true, s, p);
    };
    auto proj = [&](const std::string& p, int x, int mid, int out, int stride, const std::string& next = "") {
        int s = -1;
        int a = add_conv_pair(h, p + "_branch2a", mid, p + "_branch1", out, x, stride, &s);
        return b_then_c(p, a, mid, out, s, next);
    };
    auto ident = [&](const std::string& p, int x, int mid, int out, const std::string& next = "") {
        int a = chained_for == p ? chained : conv(p + "_branch2a", x, 1, 1, mid, true);
        chained_for.clear();
        return b_then_c(p, a, mid, out, x, next);
    };
    // 92x92 stage: each block's 3x3 layer has 64 channels, i.e. one 64-wide tile column, so its 1x1 successor can run as a tail
    // GEMM of the same workgroups (add_conv_tail): 3 launches and 3 x 13 MB of intermediate traffic fewer.  Where the shape does
    // not admit it (more than 512 tiles: four or more scales; per-layer read-back requested) the stand-alone layers run, and in
    // the reference's wiring res2b_branch2b / res2c_branch2b -- both read res2b_branch2a -- share one dual-output launch.
    int r;
    int res2_chained = -1;  // the next block's branch2a where the tail launch of this one has produced it (chain GEMM, bf16)
    {
        int s = -1;
        int a = add_conv_pair(h, "res2a_branch2a", 64, "res2a_branch1", 256, pool1, 1, &s);
        NEED(a);
        bool fits = false;
        r = add_conv_tail(h, "res2a_branch2b", "res2a_branch2c", a, s, "res2a", 64, 256, &fits, true, "res2b_branch2a", &res2_chained);
        if (!fits) {
            int b = conv("res2a_branch2b", a, 3, 1, 64, true);
            NEED(b);
            r = conv("res2a_branch2c", b, 1, 1, 256, true, s, "res2a");
        }
        NEED(r);
    }
    if (h->cfg.paper_res2c) {
        for (const char* p : {"res2b", "res2c"}) {
            const std::string P = p;
            int a = res2_chained >= 0 ? res2_chained : conv(P + "_branch2a", r, 1, 1, 64, true);
            res2_chained = -1;
            NEED(a);
            bool fits = false;
            int o = add_conv_tail(h, P + "_branch2b", P + "_branch2c", a, r, P, 64, 256, &fits, true, P == "res2b" ? "res2c_branch2a" : "", &res2_chained);
            if (!fits) {
                int b = conv(P + "_branch2b", a, 3, 1, 64, true);
                NEED(b);
                o = conv(P + "_branch2c", b, 1, 1, 256, true, r, P);
            }
            NEED(o);
            r = o;
        }
    } else {
        // vnect_model.py:50-57: res2c_branch2b consumes res2b_branch2a (`:56`), res2c_branch2a is dead and pruned
        if (!get_w(h, "res2c_branch2a/weights", {1, 1, 256, 64})) return VNECT_E_ARG;  // schema completeness, like the reference's load_weights
        const int x = r;
        int a = res2_chained >= 0 ? res2_chained : conv("res2b_branch2a", x, 1, 1, 64, true);
        res2_chained = -1;
        NEED(a);
        bool fits = false;
        int r2b = add_conv_tail(h, "res2b_branch2b", "res2b_branch2c", a, x, "res2b", 64, 256, &fits);
        if (fits) {
            NEED(r2b);
            r = add_conv_tail(h, "res2c_branch2b", "res2c_branch2c", a, r2b, "res2c", 64, 256, &fits);
            NEED(r);
        } else {
            int b2 = -1;
            int b1 = add_conv_pair(h, "res2b_branch2b", 64, "res2c_branch2b", 64, a, 1, &b2, 3, true);
            NEED(b1);
            r2b = conv("res2b_branch2c", b1, 1, 1, 256, true, x, "res2b");
            NEED(r2b);
            r = conv("res2c_branch2c", b2, 1, 1, 256, true, r2b, "res2c");
            NEED(r);
        }
    }
    r = proj("res3a", r, 128, 512, 2, "res3b");
    NEED(r);
    {
        const char* blocks[] = {"res3b", "res3c", "res3d", ""};
        for (int i = 0; i < 3; i++) {
            r = ident(blocks[i], r, 128, 512, blocks[i + 1]);
            NEED(r);
        }
    }
    r = proj("res4a", r, 256, 1024, 2);
    NEED(r);
    for (const char* p : {"res4b", "res4c", "res4d", "res4e", "res4f"}) {
        r = ident(p, r, 256, 1024);
        NEED(r);
    }
    // res5a / res5b (vnect_model.py:167-185)
    {
        int s = -1;
        int a = add_conv_pair(h, "res5a_branch2a_new", 512, "res5a_branch1_new", 1024, r, 1, &s);
        NEED(a);
        int b = conv("res5a_branch2b_new", a, 3, 1, 512, true);
        NEED(b);
        r = conv("res5a_branch2c_new", b, 1, 1, 1024, true, s, "res5a");
        NEED(r);
        a = conv("res5b_branch2a_new", r, 1, 1, 256, true);
        NEED(a);
        b = conv("res5b_branch2b_new", a, 3, 1, 128, true);
        NEED(b);
        r = conv("res5b_branch2c_new", b, 1, 1, 256, true);
        NEED(r);
    }
    // Transposed convs + BN + ReLU + deltas, one 4-phase launch (vnect_model.py:188-209).
    // out[2i-1+ky, 2j-1+kx, oc] += in[i,j,ic] * W[ky,kx,oc,ic]; phase (py,px) = (oy&1, ox&1):
    //   py = 0: ky = 1 reads row i', ky = 3 reads row i'-1;  py = 1: ky = 0 reads row i'+1, ky = 2 reads row i'.
    int feat;
    {
        const Tensor tin = h->tensors[r];
        const HostArray* W1 = get_w(h, "res5c_branch1a/kernel", {4, 4, 63, 256});
        const HostArray* W2 = get_w(h, "res5c_branch2a/kernel", {4, 4, 128, 256});
        const HostArray* ga = get_w(h, "bn5c_branch2a/gamma", {128});
        const HostArray* be = get_w(h, "bn5c_branch2a/beta", {128});
        const HostArray* mu = get_w(h, "bn5c_branch2a/moving_mean", {128});
        const HostArray* va = get_w(h, "bn5c_branch2a/moving_variance", {128});
        if (!W1 || !W2 || !ga || !be || !mu || !va) return VNECT_E_ARG;
        Layer L;
        L.op = OP_CONV, L.name = "res5c_deconv", L.in = r;
        const int featCs = h->bf16 ? 256 : 224;  // 212 channels padded to a whole number of K chunks
        feat = L.out = add_tensor(h, "res5c_branch2a_feat", S, 2 * tin.H, 2 * tin.W, 212, featCs);
        ConvArgs& a = L.a;
        a.S = S, a.H = tin.H, a.W = tin.W, a.Cs = tin.Cs;
        a.Ho = tin.H, a.Wo = tin.W, a.M = S * tin.H * tin.W;
        a.stride = 1, a.OH = 2 * tin.H, a.OW = 2 * tin.W, a.os = 2, a.nphase = 4;
        a.ntaps = 4, a.cpt = tin.Cs / (h->bf16 ? 64 : 32), a.K = 4 * tin.Cs;
        a.bf16 = h->bf16;
        a.ldc = featCs, a.ldr = 0, a.relu_cols = 128, a.Nvalid = 191;
        L.Nreal = 191, L.Kreal = 4 * 256;
        L.flops = 2.0 * (double)S * 46 * 46 * 4 * 256 * 191;
        choose_tile(L, (long long)S * 46 * 46, !h->x3);
        a.Npad = round_up(191, L.BN);
        a.w_phase_stride = (long long)a.Npad * a.K;
        std::vector<float> wp, bp, sc, sh;
        plan::pack_deconv(W1->d.data(), W2->d.data(), a.Npad, a.K, wp, L.dy, L.dx);
        // FusedBatchNorm inference (contrib batch_norm default epsilon 0.001): (x - mean) * (gamma * rsqrt(var + eps)) + beta
        plan::fold_bn(ga->d.data(), be->d.data(), mu->d.data(), va->d.data(), 128, a.Npad, bp, sc, sh);
        if (upload_layer_weights(h, L, wp) || upload(h, &L.bias, bp) || upload(h, &L.scale, sc) || upload(h, &L.shift, sh))
            return VNECT_E_HIP;
        // bone-length features (vnect_model.py:198-209): inside this launch (conv.hip, FUSE = 2) where every workgroup has one tile,
        // i.e. up to 5 scales; as a launch of their own otherwise, and when per-layer read-back is requested
        const bool shape96 = L.BM == 64 && L.BN == 96 && L.KG == 2;
        const long long deconv_items = (long long)((a.M + L.BM - 1) / L.BM) * (a.Npad / L.BN) * 4;
        const bool fuse_bone = ((L.BM == 64 && L.BN == 64 && L.KG == 1 && deconv_items <= 512) || (shape96 && deconv_items <= 256)) && a.ksplit == 1 &&
                               !h->keep_activations && !getenv("VNECT_NO_BONE_FUSE");
        a.bone = fuse_bone;
        if (fuse_bone) L.name = "res5c_deconv+bone_length";
        h->layers.push_back(L);
        if (!fuse_bone) {
            Layer Bn;
            Bn.op = OP_BONE, Bn.name = "res5c_bone_length", Bn.in = feat, Bn.out = feat;
            h->layers.push_back(Bn);
        }
    }
    // head (vnect_model.py:211-217)
    {
        const HostArray* Wk = get_w(h, "res5c_branch2c/kernel", {1, 1, 128, 84});
        if (!Wk) return VNECT_E_ARG;
        // tf.layers.conv2d without bias == the tc.layers form with zero biases
        HostArray z;
        z.d.assign(84, 0.f), z.shape = {84};
        h->weights["res5c_branch2c/weights"] = *Wk;
        h->weights["res5c_branch2c/biases"] = z;
        bool fits = false;
        h->t_out = add_conv_tail(h, "res5c_branch2b", "res5c_branch2c", feat, -1, "res5c_branch2c", 128, 84, &fits, false);
        if (!fits) {
            const int hd = conv("res5c_branch2b", feat, 3, 1, 128, true);
            h->t_out = hd < 0 ? -1 : conv("res5c_branch2c", hd, 1, 1, 84, false);
        }
        h->weights.erase("res5c_branch2c/weights"), h->weights.erase("res5c_branch2c/biases");
        NEED(h->t_out);
    }
#undef NEED
    // buffers
    // + 64 pixels of slack per tensor: the streaming conv kernel's epilogue reads shortcut rows and writes output rows of
    // its last 64-row tile without a per-row bound check (rows >= M land in the slack and are never read)
    auto padded = [](const Tensor& t) { return (t.bytes() + (size_t)64 * t.Cs * t.esz + 255) & ~(size_t)255; };
    if (!h->keep_activations) {
        // Activation arena: a tensor lives from the layer that writes it to the last layer that reads it, and tensors with
        // disjoint lifetimes share addresses (first fit over the live intervals).  The per-frame working set is then the
        // peak live set (~0.1 GB at S = 3) instead of one buffer per layer output (~0.35 GB), so weights + activations stay
        // inside the 256 MiB Infinity Cache from frame to frame.
        const int nt = (int)h->tensors.size(), nl = (int)h->layers.size();
        std::vector<int> first(nt, nl + 1), last(nt, -2);
        auto touch = [&](int t, int l) {
            if (t < 0) return;
            first[t] = std::min(first[t], l), last[t] = std::max(last[t], l);
        };
        touch(h->t_input4, -1);  // written by the pre-processing
        for (int l = 0; l < nl; l++) {
            const Layer& L = h->layers[l];
            touch(L.in, l), touch(L.resid, l), touch(L.out, l), touch(L.out2, l), touch(L.out3, l);
            // the stem may run this pair itself (setup_stem, PAIR form) and then writes its outputs while it still reads the batch tensor:
            // they must not share addresses with anything alive from conv1 on
            if (L.frag_w && l == h->l_pool1 + 1) touch(L.out, h->l_conv1), touch(L.out2, h->l_conv1);
        }
        touch(h->t_out, nl);  // read by the post-processing
        std::vector<size_t> need(nt), off;
        for (int t = 0; t < nt; t++) {
            if (last[t] < first[t]) first[t] = -1, last[t] = nl;  // never touched by a layer: keep it private
            need[t] = padded(h->tensors[t]);
        }
        const size_t total = plan::arena_first_fit(first, last, need, off);
        char* base = nullptr;
        int rc = dev_alloc(h, &base, total);
        if (rc) return rc;
        HIPCK(h, hipMemset(base, 0, total));
        for (int t = 0; t < nt; t++) h->tensors[t].d = (float*)(base + off[t]);
        h->arena_bytes = total, h->arena_off = off;
    } else {
        for (Tensor& t : h->tensors) {
            char* p = nullptr;
            int rc = dev_alloc(h, &p, padded(t));
            if (rc) return rc;
            t.d = (float*)p;
            HIPCK(h, hipMemset(t.d, 0, padded(t)));
        }
    }
    size_t ws = 0;
    for (Layer& L : h->layers)
        if (L.op == OP_CONV && L.a.ksplit > 1)
            ws = std::max(ws, (size_t)L.a.ksplit * ((size_t)L.a.S * L.a.OH * L.a.OW + 64) * L.a.Npad);
    h->ws_floats = ws;
    if (ws) {
        int rc = dev_alloc(h, &h->ws, ws);
        if (rc) return rc;
    }
    HIPCK(h, hipDeviceSynchronize());
    h->conv_flops = 0, h->conv_launches = 0;
    for (Layer& L : h->layers) {
        if (L.op != OP_CONV) continue;
        ConvArgs& a = L.a;
        bind_activations(h, L);
        {   // tap byte offsets for the buffer-addressed loads (kernels.h)
            const int esz = a.bf16 ? 2 : 4, nt = a.nphase * a.ntaps;
            int lo = 0;
            for (int t = 0; t < nt; t++) lo = std::min(lo, (L.dy[t] * a.W + L.dx[t]) * a.Cs * esz);
            a.tap_bias = -lo;
            a.tapgrid = 0;
            if (a.nphase == 1 && a.ntaps == 1 && L.dy[0] == 0 && L.dx[0] == 0 && !a.pixmode) a.tapgrid = 1;
            if (a.nphase == 1 && a.ntaps == 9 && !a.pixmode) {
                bool ok = true;
                for (int t = 0; t < 9; t++) ok = ok && L.dy[t] == t / 3 - 1 && L.dx[t] == t % 3 - 1;
                if (ok) a.tapgrid = 3;
            }
            a.dy_pack = a.dx_pack = 0;
            for (int t = 0; t < nt; t++) {
                if (L.dy[t] < -8 || L.dy[t] > 7 || L.dx[t] < -8 || L.dx[t] > 7) {
                    h->err = "internal: filter tap outside the packed range";
                    return VNECT_E_ARG;
                }
                a.dy_pack |= (unsigned long long)(L.dy[t] + 8) << (4 * t), a.dx_pack |= (unsigned long long)(L.dx[t] + 8) << (4 * t);
            }
        }
        if (a.ksplit > 1) {
            ReduceArgs& q = L.r;
            q.bias = L.bias, q.scale = L.scale, q.shift = L.shift;
            a.slab_pix = (long long)a.S * a.OH * a.OW + 64, q.slab_pix = a.slab_pix;
            q.npix = (long long)a.S * a.OH * a.OW, q.Npad = a.Npad, q.Nvalid = a.Nvalid, q.ldc = a.ldc, q.ldr = a.ldr;
            q.ksplit = a.ksplit, q.relu_cols = a.relu_cols;
            q.bf16 = a.bf16, q.out_f32 = a.out_f32;
        }
        h->conv_flops += L.flops;
        h->conv_launches += 1;
    }
    setup_stem(h);
    if (h->stem_pair) h->conv_launches -= 1;
    return VNECT_OK;
}

// ---- launch sequences -------------------------------------------------------------------------------
// `stem_done`: the caller has launched the stem from the frame already (enqueue_frame: it takes the frame's arguments by value)
int run_network(vnect_handle* h, bool timed, bool stem_done = false)
{
    for (Layer& L : h->layers) {
        const int li = (int)(&L - h->layers.data());
        if (h->stem_mode && (li == h->l_conv1 || li == h->l_pool1)) {
            if (li == h->l_conv1 && !stem_done) {  // from the batch tensor (stem_mode 1, or vnect_forward on a stem_mode 2 handle)
                StemArgs a = h->stem;
                a.from_frame = 0;
                a.prof = timed ? h->d_prof + PROF_SLOTS * li : nullptr;
                a.prof_end = timed ? h->d_prof_end + (size_t)PROF_WGS * li : nullptr;
                HIPCK(h, launch_stem(a, h->st));
            }
            continue;
        }
        if (h->stem_mode && h->stem_pair && li == h->l_pool1 + 1) continue;  // ran inside the stem launch
        if (L.op == OP_CONV) {
            ConvArgs a = L.a;
            a.prof = timed ? h->d_prof + PROF_SLOTS * (&L - h->layers.data()) : nullptr;
            a.prof_end = timed ? h->d_prof_end + (size_t)PROF_WGS * (&L - h->layers.data()) : nullptr;
            HIPCK(h, launch_conv(a, L.BM, L.BN, L.KG, h->st));
            if (L.a.ksplit > 1) HIPCK(h, launch_reduce(L.r, h->st));
        } else if (L.op == OP_POOL) {
            const Tensor &i = h->tensors[L.in], &o = h->tensors[L.out];
            HIPCK(h, launch_maxpool(i.d, o.d, i.S, i.H, i.W, i.Cs, o.H, o.W, h->bf16, h->st));
        } else {
            const Tensor& t = h->tensors[L.out];
            HIPCK(h, launch_bone(t.d, (long long)t.S * t.H * t.W, t.Cs, h->bf16, h->st));
        }
    }
    return VNECT_OK;
}

// Crop geometry -> d_fp, only if it differs from what the device holds (a stream of equally sized crops never uploads).
// In-stream, so frames still in flight keep the geometry they were launched with.
int sync_geometry(vnect_handle* h, const FrameParams& fp)
{
    if (h->fp_dev_valid && memcmp(&h->fp_dev, &fp, sizeof fp) == 0) return VNECT_OK;
    const int r = h->fp_ring = (h->fp_ring + 1) % RING;  // a staging slot of its own: earlier copies may still be queued
    *h->h_fp[r] = fp;
    HIPCK(h, hipMemcpyAsync(h->d_fp, h->h_fp[r], sizeof(FrameParams), hipMemcpyHostToDevice, h->st));
    h->fp_dev = fp, h->fp_dev_valid = true;
    return VNECT_OK;
}

int run_pre(vnect_handle* h, const FrameDyn& dyn, bool timed = false, bool want_batch = false)
{
    if (h->stem_mode == 2 && !want_batch) {
        StemArgs a = h->stem;
        a.prof = timed ? h->d_prof + PROF_SLOTS * h->l_conv1 : nullptr;
        a.prof_end = timed ? h->d_prof_end + (size_t)PROF_WGS * h->l_conv1 : nullptr;
        // gen_input_batch + conv1 + pool1 in ONE launch, the batch tensor never written: for frames whose squarify step is a copy (long
        // side == 368) at scales whose rectangles fit the kernel's scratch.  Any other frame: pyramid_kernel, then the stem from the
        // batch tensor -- both in front of the graph, which starts at res2a either way.  Same results bit for bit.
        if (h->stem_frame_ok && h->fp_dev_valid && h->fp_dev.sq.copy) {
            a.from_frame = 1, a.dyn = dyn;
            HIPCK(h, launch_stem(a, h->st));
            return VNECT_OK;
        }
        HIPCK(h, launch_pyramid(h->d_fp, dyn, h->d_stabs, h->tensors[h->t_input4].d, h->Snet,
                                h->sharded ? h->cfg.pyramid_rank : 0, h->bf16, h->st));
        a.from_frame = 0;
        HIPCK(h, launch_stem(a, h->st));
        return VNECT_OK;
    }
    HIPCK(h, launch_pyramid(h->d_fp, dyn, h->d_stabs, h->tensors[h->t_input4].d, h->Snet,
                            h->sharded ? h->cfg.pyramid_rank : 0, h->bf16, h->st));
    return VNECT_OK;
}

// multi-scale merge + arg-max (graph-capturable: no per-frame arguments)
int run_argmax(vnect_handle* h)
{
    const float* maps = h->sharded ? h->gather : h->tensors[h->t_out].d;
    HIPCK(h, launch_argmax(maps, h->mgeo, h->d_part, h->st));
    return VNECT_OK;
}

// filters + read-off; results go straight to `out` (a device-mapped pinned host slot)
int run_joints(vnect_handle* h, const FrameDyn& dyn, JointsOut* out, int stream = 0)
{
    const float* maps = h->sharded ? h->gather : h->tensors[h->t_out].d;
    HIPCK(h, launch_joints(h->d_part, maps, h->mgeo, h->d_fb + stream, h->d_fp, dyn, h->cfg.numpy_promotion, out, h->st));
    return VNECT_OK;
}

// both in one launch (post.hip: post_kernel): takes the frame's arguments by value, so it runs behind the graph, not inside it
int run_post(vnect_handle* h, const FrameDyn& dyn, JointsOut* out, int stream = 0)
{
    const float* maps = h->sharded ? h->gather : h->tensors[h->t_out].d;
    HIPCK(h, launch_post(maps, h->mgeo, h->d_part, h->d_ticket, h->d_fb + stream, h->d_fp, dyn, h->cfg.numpy_promotion, out, h->st));
    return VNECT_OK;
}

// OneEuroFilter.py:65-66: `if self.__lasttime and timestamp: self.__freq = 1.0 / (timestamp - self.__lasttime)`.
//   t == last  -> ZeroDivisionError (VNECT_E_TIMESTAMP);
//   t <  last  -> freq < 0, so alpha = 1 / (1 + tau * freq) leaves (0, 1] and LowPassFilter.__setAlpha raises ValueError
//                 (OneEuroFilter.py:19-23) -> VNECT_E_TIMEORDER.
// Nothing is committed here: the reference would leave half-updated filters behind its exception, this path rejects the call
// before any state changes, and the host-side copy of the last timestamps moves only after the frame has been enqueued.
int check_time(vnect_handle* h, double t2d, double t3d, int s = 0)
{
    if (h->have2[s] && h->last2[s] != 0.0 && t2d != 0.0) {
        if (t2d == h->last2[s]) return fail(h, VNECT_E_TIMESTAMP, "t2d equals the previous 2-D filter timestamp");
        if (t2d < h->last2[s]) return fail(h, VNECT_E_TIMEORDER, "t2d is earlier than the previous 2-D filter timestamp");
    }
    if (h->have3[s] && h->last3[s] != 0.0 && t3d != 0.0) {
        if (t3d == h->last3[s]) return fail(h, VNECT_E_TIMESTAMP, "t3d equals the previous 3-D filter timestamp");
        if (t3d < h->last3[s]) return fail(h, VNECT_E_TIMEORDER, "t3d is earlier than the previous 3-D filter timestamp");
    }
    return VNECT_OK;
}
// `self.__lasttime = timestamp` runs on every call, also with timestamp 0.0 / None
void commit_time(vnect_handle* h, double t2d, double t3d, int s = 0)
{
    h->have2[s] = h->have3[s] = true, h->last2[s] = t2d, h->last3[s] = t3d;
}

int reset_filters_impl(vnect_handle* h, int stream = -1)  // -1: every stream
{
    std::vector<FilterBank> fb(1);
    memset(fb.data(), 0, sizeof(FilterBank));
    for (int j = 0; j < NJ; j++) {
        for (int k = 0; k < 2; k++) {  // filter_config_2d, estimator.py:34-39
            Filt& f = fb[0].f2[j][k];
            f.freq = 30, f.mincutoff = 1.7, f.beta = 0.3, f.dcutoff = 0.4;
        }
        for (int k = 0; k < 3; k++) {  // filter_config_3d, estimator.py:40-45
            Filt& f = fb[0].f3[j][k];
            f.freq = 30, f.mincutoff = 0.8, f.beta = 0.4, f.dcutoff = 0.4;
        }
    }
    for (int s = 0; s < VNECT_MAX_STREAMS; s++) {
        if (stream >= 0 && s != stream) continue;
        HIPCK(h, hipMemcpyAsync(h->d_fb + s, fb.data(), sizeof(FilterBank), hipMemcpyHostToDevice, h->st));
        h->have2[s] = h->have3[s] = false;
    }
    HIPCK(h, hipStreamSynchronize(h->st));
    return VNECT_OK;
}

// ---- roctx ranges (SURVEY 5 aux: tracing).  Opt-in with VNECT_ROCTX=1; libroctx64 is dlopen'ed, so nothing links it. -----
// The ranges bracket the ENQUEUE of each stage of a frame on the host thread (pre-processing, conv stack + merge/arg-max,
// filters + read-off); rocprofv3 --marker-trace shows them above the kernel rows of the same stream.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool tried = false;
} g_roctx;
void roctx_load()
{
    if (g_roctx.tried) return;
    g_roctx.tried = true;
    const char* e = getenv("VNECT_ROCTX");
    if (!e || !atoi(e)) return;
    void* lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);  // what rocprofv3 --marker-trace listens to
    if (!lib) lib = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return;
    g_roctx.push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
    g_roctx.pop = (int (*)())dlsym(lib, "roctxRangePop");
    if (!g_roctx.push || !g_roctx.pop) g_roctx.push = nullptr, g_roctx.pop = nullptr;
}
struct RoctxRange {
    explicit RoctxRange(const char* name) { if (g_roctx.push) g_roctx.push(name); }
    ~RoctxRange() { if (g_roctx.pop) g_roctx.pop(); }
};

// ---- RCCL, opened lazily so single-GPU use never loads it -----------------------------------------------
typedef struct { char internal[128]; } nccl_uid;
void* g_rccl = nullptr;
int (*p_ncclGetUniqueId)(nccl_uid*) = nullptr;
int (*p_ncclCommInitRank)(void**, int, nccl_uid, int) = nullptr;
int (*p_ncclAllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
int (*p_ncclCommDestroy)(void*) = nullptr;
const char* (*p_ncclGetErrorString)(int) = nullptr;

// Which copy?  A process that also runs torch.distributed's "nccl" backend (bench.py --pyramid, parallel.PyramidJob) has torch's
// bundled torch/lib/librccl.so mapped already; /opt/rocm/lib/librccl.so.1 is ANOTHER build with the same SONAME.  Two copies in one
// process would each keep their own bootstrap threads, proxy state and IPC caches on the same device, and with RTLD_GLOBAL the
// second one's internal symbols could bind into the first.  So: (1) VNECT_RCCL_LIB names a file explicitly; (2) a copy this
// process has mapped already (dl_iterate_phdr: any object whose file name starts with "librccl.so") is REUSED (RTLD_NOLOAD: a
// reference to that very mapping); (3) only then is librccl.so.1 / librccl.so opened through the ordinary search (this library's
// RUNPATH is the ROCm it was built with).  Always RTLD_LOCAL: only the five entry points below are looked up, by dlsym.
std::mutex g_rccl_mu;
bool g_rccl_reused = false;
std::string g_rccl_path;
int rccl_find_mapped(struct dl_phdr_info* info, size_t, void* out)
{
    const char* n = info->dlpi_name;
    if (!n || !*n) return 0;
    const char* b = strrchr(n, '/');
    b = b ? b + 1 : n;
    if (strncmp(b, "librccl.so", 10) != 0) return 0;
    *(std::string*)out = n;
    return 1;
}
bool load_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl) return true;
    g_rccl_reused = false;
    const char* forced = getenv("VNECT_RCCL_LIB");
    if (forced && *forced) {
        g_rccl = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!g_rccl) return false;  // an explicit choice that cannot be honoured is an error, not a reason to pick another copy
    }
    if (!g_rccl) {
        std::string mapped;
        dl_iterate_phdr(rccl_find_mapped, &mapped);
        if (!mapped.empty()) g_rccl = dlopen(mapped.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (!g_rccl) g_rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);  // by SONAME
        g_rccl_reused = g_rccl != nullptr;
    }
    if (!g_rccl) g_rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!g_rccl) g_rccl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!g_rccl) return false;
    p_ncclGetUniqueId = (int (*)(nccl_uid*))dlsym(g_rccl, "ncclGetUniqueId");
    p_ncclCommInitRank = (int (*)(void**, int, nccl_uid, int))dlsym(g_rccl, "ncclCommInitRank");
    p_ncclAllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(g_rccl, "ncclAllGather");
    p_ncclCommDestroy = (int (*)(void*))dlsym(g_rccl, "ncclCommDestroy");
    p_ncclGetErrorString = (const char* (*)(int))dlsym(g_rccl, "ncclGetErrorString");
    if (p_ncclGetUniqueId && p_ncclCommInitRank && p_ncclAllGather && p_ncclCommDestroy) {
        Dl_info di{};  // the file the entry point really lives in (what bench.py reports)
        g_rccl_path = dladdr((void*)p_ncclAllGather, &di) && di.dli_fname ? di.dli_fname : "?";
        return true;
    }
    g_rccl = nullptr;
    return false;
}

// rank r's (46,46,84) maps -> slot r of the (S,46,46,84) gather buffer on every rank (the one exchange of SURVEY 8e),
// by ncclAllGather or by peer writes (kernels.h: XchgArgs).  `seq` numbers the frame (the p2p flag value).
int exchange_maps(vnect_handle* h, unsigned long long seq, int ring)
{
    const Tensor& t = h->tensors[h->t_out];
    if (h->cfg.exchange == VNECT_XCHG_P2P) {
        if (!h->p2p_ready) return fail(h, VNECT_E_STATE, "pyramid-sharded handle (p2p): call vnect_comm_p2p_init before inference");
        XchgArgs a{};
        a.src = t.d, a.gather = h->gather, a.tickets = h->xtickets, a.status = h->h_xstatus_dev + ring, a.dfail = h->d_xfail;
        for (int r = 0; r < h->S; r++) a.block[r] = h->xpeer[r];
        a.rank = h->cfg.pyramid_rank, a.nranks = h->S, a.parity = (int)(seq & 1), a.seq = (unsigned)(seq + 1);
        const char* lim = getenv("VNECT_XCHG_SPINS");  // polls (~1 us each) before a missing peer fails the frame; default ~2 s
        a.spin_limit = lim && atoi(lim) > 0 ? (unsigned)atoi(lim) : 2000000u;
        HIPCK(h, launch_exchange(a, h->st));
        return VNECT_OK;
    }
    if (!h->comm) return fail(h, VNECT_E_STATE, "pyramid-sharded handle: call vnect_comm_init before inference");
    const int rc = p_ncclAllGather(t.d, h->gather, (size_t)HM * HM * MAPC, 7 /* ncclFloat32 */, h->comm, h->st);
    if (rc != 0)
        return fail(h, VNECT_E_COMM, std::string("ncclAllGather: ") + (p_ncclGetErrorString ? p_ncclGetErrorString(rc) : "error"));
    return VNECT_OK;
}
bool comm_ready(const vnect_handle* h) { return h->cfg.exchange == VNECT_XCHG_P2P ? h->p2p_ready : h->comm != nullptr; }

// The part of a frame without per-frame arguments (what the hipGraph holds): conv stack, merge + arg-max.  The pyramid kernel
// before it and the joints kernel after it take the frame's arguments by value and are launched around the graph.  A
// pyramid-sharded handle's graph ends with the conv stack: the exchange (whose flag value / parity change every frame) and the
// merge + arg-max launch behind it are eager (two captured parts around the exchange -- the
// second part is the single arg-max launch, which gains nothing from a graph of its own).
int run_frame_kernels(vnect_handle* h, bool timed)
{
    const size_t pbytes = h->layers.size() * PROF_SLOTS * sizeof(unsigned long long);
    int rc = run_network(h, timed, h->stem_mode == 2);  // stem_mode 2: run_pre has launched the stem in front of this
    if (rc) return rc;
    if (!h->sharded && !h->post_merged && (rc = run_argmax(h))) return rc;
    if (timed) {
        HIPCK(h, hipMemcpyAsync(h->h_prof, h->d_prof, pbytes, hipMemcpyDeviceToHost, h->st));
        HIPCK(h, hipMemcpyAsync(h->h_prof_end, h->d_prof_end, h->layers.size() * PROF_WGS * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->st));
    }
    return VNECT_OK;
}

int build_graph(vnect_handle* h)
{
    if (h->gexec) hipGraphExecDestroy(h->gexec), h->gexec = nullptr;
    if (h->graph) hipGraphDestroy(h->graph), h->graph = nullptr;
    if (h->pgexec) hipGraphExecDestroy(h->pgexec), h->pgexec = nullptr;
    if (h->pgraph) hipGraphDestroy(h->pgraph), h->pgraph = nullptr;
    if (!h->cfg.use_graph) return VNECT_OK;
    HIPCK(h, hipStreamBeginCapture(h->st, hipStreamCaptureModeThreadLocal));
    int rc = run_frame_kernels(h, false);
    hipError_t e = hipStreamEndCapture(h->st, &h->graph);
    if (rc) return rc;
    HIPCK(h, e);
    HIPCK(h, hipGraphInstantiate(&h->gexec, h->graph, nullptr, nullptr, 0));
    if (h->is_twin) return VNECT_OK;  // profiled frames always run on the first lane
    // profiling twin: identical launches, but every conv kernel stamps {min start, max end} (s_memrealtime)
    HIPCK(h, hipStreamBeginCapture(h->st, hipStreamCaptureModeThreadLocal));
    rc = run_frame_kernels(h, true);
    e = hipStreamEndCapture(h->st, &h->pgraph);
    if (rc) return rc;
    HIPCK(h, e);
    HIPCK(h, hipGraphInstantiate(&h->pgexec, h->pgraph, nullptr, nullptr, 0));
    return VNECT_OK;
}

void destroy_twins(vnect_handle* h)
{
    for (vnect_handle* t : h->twins) {
        if (t->st) hipStreamSynchronize(t->st);
        if (t->gexec) hipGraphExecDestroy(t->gexec);
        if (t->graph) hipGraphDestroy(t->graph);
        for (int i = 0; i < RING; i++)
            if (t->h_fp[i]) hipHostFree(t->h_fp[i]);
        for (void* p : t->dev_allocs) hipFree(p);
        if (t->st) hipStreamDestroy(t->st);
        delete t;
    }
    h->twins.clear();
    h->last_lane = nullptr;
}

// One more lane of the frame pipeline (vnect_config::lanes): same layers and weights, its own stream, activation arena,
// workspace, arg-max scratch, geometry block and graph.
int build_twin(vnect_handle* h)
{
    vnect_handle* t = new vnect_handle();
    h->twins.push_back(t);
    t->is_twin = true;
    t->cfg = h->cfg, t->S = h->S, t->Snet = h->Snet, t->bf16 = h->bf16, t->keep_activations = false;
    HIPCK(h, hipStreamCreateWithFlags(&t->st, hipStreamNonBlocking));
    // shared: read-only tables and frames; the filter bank (its users are chained by events)
    t->frames = h->frames, t->d_stabs = h->d_stabs, t->mgeo = h->mgeo, t->d_fb = h->d_fb;
    t->slots = h->slots;
    int rc;
    if ((rc = dev_alloc(t, &t->d_fp, 1))) return fail(h, rc, t->err);
    if ((rc = dev_alloc(t, &t->d_part, (size_t)NJ * ARG_SLABS_MAX))) return fail(h, rc, t->err);
    if ((rc = dev_alloc(t, &t->d_ticket, 4))) return fail(h, rc, t->err);
    HIPCK(h, hipMemset(t->d_ticket, 0, 4 * sizeof(unsigned)));
    t->post_merged = h->post_merged;
    for (int i = 0; i < RING; i++) HIPCK(h, hipHostMalloc((void**)&t->h_fp[i], sizeof(FrameParams), hipHostMallocDefault));
    t->tensors = h->tensors, t->layers = h->layers, t->tensor_by_name = h->tensor_by_name;
    t->t_input4 = h->t_input4, t->t_out = h->t_out;
    char* base = nullptr;
    if ((rc = dev_alloc(t, &base, h->arena_bytes))) return fail(h, rc, t->err);
    HIPCK(h, hipMemset(base, 0, h->arena_bytes));
    for (size_t i = 0; i < t->tensors.size(); i++) t->tensors[i].d = (float*)(base + h->arena_off[i]);
    t->ws_floats = h->ws_floats;
    if (t->ws_floats && (rc = dev_alloc(t, &t->ws, t->ws_floats))) return fail(h, rc, t->err);
    for (Layer& L : t->layers)
        if (L.op == OP_CONV) bind_activations(t, L);
    t->l_conv1 = h->l_conv1, t->l_pool1 = h->l_pool1;
    t->stabs_host = h->stabs_host;
    setup_stem(t);  // same plan as lane 0, this lane's arena and geometry block
    t->finalized = true;
    if ((rc = build_graph(t))) return fail(h, rc, t->err);
    HIPCK(h, hipStreamSynchronize(t->st));
    return VNECT_OK;
}

int build_twins(vnect_handle* h)
{
    destroy_twins(h);
    if (h->cfg.lanes < 2 || h->sharded || h->keep_activations) return VNECT_OK;
    for (int i = 1; i < h->cfg.lanes; i++) {
        int rc = build_twin(h);
        if (rc) return rc;
    }
    return VNECT_OK;
}

// enqueue one frame from a resident slot; results land in h_out[ring]
int enqueue_frame(vnect_handle* h, int slot, double t2d, double t3d, int* ring_out, int stream = 0)
{
    if (stream < 0 || stream >= VNECT_MAX_STREAMS) return fail(h, VNECT_E_ARG, "stream out of range");
    if (stream != 0 && h->sharded) return fail(h, VNECT_E_ARG, "a pyramid-sharded handle serves one stream");
    if (slot < 0 || slot >= (int)h->slots.size() || h->slots[slot].H == 0)
        return fail(h, VNECT_E_ARG, "frame slot empty or out of range");
    const unsigned long long max_in_flight = h->twins.empty() ? 2 : h->twins.size() + 1;  // one lane: two frames queue on its stream
    if (h->seq_submit - h->seq_collect >= max_in_flight) return fail(h, VNECT_E_STATE, "too many frames in flight: collect one first");
    if (h->sharded && !comm_ready(h))  // refuse before any filter / timestamp state changes
        return fail(h, VNECT_E_STATE, "pyramid-sharded handle: call vnect_comm_init / vnect_comm_p2p_init before inference");
    const auto& si = h->slots[slot];
    FrameParams fp;
    int rc = squarify_params(h, si.H, si.W, &fp);
    if (rc) return rc;
    rc = check_time(h, t2d, t3d, stream);
    if (rc) return rc;
    const int ring = (int)(h->seq_submit % RING);
    FrameDyn dyn{};
    dyn.t2d = t2d, dyn.t3d = t3d;
    dyn.row_stride = si.stride;
    dyn.frame = h->frames + (size_t)slot * h->cfg.max_frame_bytes;
    const bool timed = h->profiling;
    // Lane: the first whose last frame has been collected (lane 0 when nothing is in flight).  Frames on different lanes
    // overlap -- the idle CUs between one frame's launches are the other frames' -- and only the post-processing launch of a frame
    // (post_kernel: merge + arg-max + joints; with VNECT_NO_POST_MERGE=1 only the joints kernel) waits for the SAME video's previous
    // frame.  Measured (round 3, A/B in one call): 1 388-1 390 frames/s three deep with the merged launch, 1 388-1 389 with the two
    // launches -- ordering the 17-us merged launch instead of the 11-us joints kernel costs nothing measurable.
    vnect_handle* L = h;
    if (!h->twins.empty() && !timed && h->lane_seq >= (long long)h->seq_collect)
        for (vnect_handle* t : h->twins)
            if (t->lane_seq < (long long)h->seq_collect) {
                L = t;
                break;
            }
    if ((rc = sync_geometry(L, fp))) return fail(h, rc, L->err);
    if (timed) HIPCK(h, hipEventRecord(h->ev[0], L->st));
    {
        RoctxRange r("vnect:gen_input_batch");
        if ((rc = run_pre(L, dyn, timed))) return fail(h, rc, L->err);
    }
    RoctxRange r_net("vnect:conv_stack+merge+argmax");
    // use_graph 2 (auto): a frame submitted while nothing is in flight -- the synchronous pattern -- is launched eagerly (median
    // 6.6 us shorter than pyramid + graph replay + joints, A/B in one call, both precisions; the host has nothing else to do
    // meanwhile), a frame submitted behind others replays the graph (the host must stay ahead of two or three lanes)
    const bool replay = L->gexec && (h->cfg.use_graph == 1 || h->seq_submit != h->seq_collect);
    if (replay && !timed) {
        HIPCK(h, hipGraphLaunch(L->gexec, L->st));
    } else if (L->pgexec && timed) {
        HIPCK(h, hipGraphLaunch(L->pgexec, L->st));
    } else {
        rc = run_frame_kernels(L, timed);
        if (rc) return fail(h, rc, L->err);
    }
    if (L->sharded) {  // the one exchange of the pyramid path, then the merge + arg-max over everybody's maps
        if (g_roctx.pop) g_roctx.pop(), g_roctx.push("vnect:exchange+merge+argmax");
        if ((rc = exchange_maps(L, h->seq_submit, ring))) return fail(h, rc, L->err);
        dyn.xfail = L->d_xfail, dyn.xseq = (unsigned)(h->seq_submit + 1);  // post_kernel skips the joints stage of a frame whose exchange failed
        if (!L->post_merged && (rc = run_argmax(L))) return fail(h, rc, L->err);
    }
    if (g_roctx.pop) g_roctx.pop(), g_roctx.push(L->post_merged ? "vnect:merge+argmax+filters+readoff" : "vnect:filters+readoff");  // r_net's pop now closes this range
    // the filters are a chain WITHIN a video stream: this frame's post-processing waits for the stream's previous frame if that one
    // ran on another lane and may still be in flight (frames of other streams are no concern of it)
    if (h->stream_seq[stream] >= (long long)h->seq_collect && h->stream_lane[stream] && h->stream_lane[stream] != L)
        HIPCK(h, hipStreamWaitEvent(L->st, h->done[h->stream_seq[stream] % RING], 0));
    // writes the ring slot in pinned host memory
    if ((rc = L->post_merged ? run_post(L, dyn, h->h_out_dev[ring], stream) : run_joints(L, dyn, h->h_out_dev[ring], stream))) return fail(h, rc, L->err);
    if (timed) HIPCK(h, hipEventRecord(h->ev[3], L->st));
    HIPCK(h, hipEventRecord(h->done[ring], L->st));
    commit_time(h, t2d, t3d, stream);  // only now: every launch of the frame has been accepted
    h->stream_seq[stream] = (long long)h->seq_submit, h->stream_lane[stream] = L, h->ring_stream[ring] = stream;
    h->last_lane = L;
    L->lane_seq = (long long)h->seq_submit;
    h->slots[slot].last_use = (long long)h->seq_submit;
    h->seq_submit++;
    *ring_out = ring;
    return VNECT_OK;
}

int collect_impl(vnect_handle* h, double* j2, float* j3, int32_t* stream_out = nullptr)
{
    if (h->seq_collect == h->seq_submit) return fail(h, VNECT_E_STATE, "nothing in flight");
    const int ring = (int)(h->seq_collect % RING);
    // the caller is about to consume the joints: poll (a frame is ~1 ms) before falling back to a blocking wait, whose
    // wake-up alone costs tens of microseconds of idle GPU per frame
    hipError_t q = hipErrorNotReady;
    for (int spin = 0; spin < 200000 && (q = hipEventQuery(h->done[ring])) == hipErrorNotReady; spin++) {}
    if (q == hipErrorNotReady) q = hipEventSynchronize(h->done[ring]);
    HIPCK(h, q);
    h->seq_collect++;
    if (h->h_xstatus && h->h_xstatus[ring]) {  // one word per ring slot: the error lands on the frame it belongs to
        h->h_xstatus[ring] = 0;
        // The frame's joints stage was skipped on the device (post_kernel: the filter banks did not advance on stale maps), but the
        // ranks are out of step now and the host's timestamps have moved: VNECT_E_COMM means tear the job down and reconnect.
        return fail(h, VNECT_E_COMM, "pyramid exchange: a peer's maps did not arrive within the bound (ranks out of step?); "
                                     "destroy the handles of every rank and reconnect");
    }
    if (j2) memcpy(j2, h->h_out[ring]->j2d, sizeof(double) * NJ * 2);
    if (j3) memcpy(j3, h->h_out[ring]->j3d, sizeof(float) * NJ * 3);
    if (stream_out) *stream_out = h->ring_stream[ring];
    if (h->profiling) {
        float frame_ms = 0;
        hipEventElapsedTime(&frame_ms, h->ev[0], h->ev[3]);
        unsigned long long first = ~0ull, last = 0;
        double conv_ms = 0;
        for (size_t i = 0; i < h->layers.size(); i++) {
            Layer& L = h->layers[i];
            unsigned long long* p = h->h_prof + PROF_SLOTS * i;
            const unsigned long long t0 = p[0];
            unsigned long long t1 = 0;
            if (L.op == OP_CONV) {  // latest workgroup end of this launch (slots past the grid stay 0)
                const unsigned long long* e = h->h_prof_end + (size_t)PROF_WGS * i;
                for (int k = 0; k < PROF_WGS; k++) t1 = std::max(t1, e[k]);
                for (int k = 1; k <= 8; k++) p[k] = t1;  // vnect_get_layer_stamps keeps its layout
            }
            L.last_ms = 0;
            if (L.op != OP_CONV || t1 <= t0) continue;
            L.last_ms = (float)((double)(t1 - t0) * 1e-5);  // 100 MHz ticks -> ms
            conv_ms += L.last_ms;
            first = std::min(first, t0), last = std::max(last, t1);
        }
        // slot of a conv kernel on the stream: its start to the next conv kernel's start when that one follows directly,
        // else its own duration + the median boundary of the direct pairs (pool / reduce / bone / arg-max follow it)
        std::vector<double> gaps;
        std::vector<size_t> convs;
        for (size_t i = 0; i < h->layers.size(); i++)
            if (h->layers[i].op == OP_CONV && h->layers[i].last_ms > 0) convs.push_back(i);
        auto t_start = [&](size_t i) { return h->h_prof[PROF_SLOTS * i]; };
        auto t_end = [&](size_t i) {
            unsigned long long e = 0;
            for (int k = 1; k <= 8; k++) e = std::max(e, h->h_prof[PROF_SLOTS * i + k]);
            return e;
        };
        auto direct = [&](size_t a, size_t b) {  // no kernel in between (the stem stands for conv1 AND pool1)
            return (b == a + 1 || (h->stem_mode && (int)a == h->l_conv1 && b == a + (h->stem_pair ? 3 : 2))) && h->layers[a].a.ksplit == 1;
        };
        for (size_t c = 0; c + 1 < convs.size(); c++)
            if (direct(convs[c], convs[c + 1]) && t_start(convs[c + 1]) > t_end(convs[c]))
                gaps.push_back((double)(t_start(convs[c + 1]) - t_end(convs[c])) * 1e-5);
        std::sort(gaps.begin(), gaps.end());
        const double med_gap = gaps.empty() ? 0.0 : gaps[gaps.size() / 2];
        double slot_ms = 0;
        for (size_t c = 0; c < convs.size(); c++) {
            const size_t i = convs[c];
            if (c + 1 < convs.size() && direct(i, convs[c + 1]) && t_start(convs[c + 1]) > t_start(i))
                slot_ms += (double)(t_start(convs[c + 1]) - t_start(i)) * 1e-5;
            else
                slot_ms += h->layers[i].last_ms + med_gap;
        }
        h->tim.conv_slot_ms += slot_ms;
        h->tim.frames++;
        h->tim.total_ms += frame_ms;                                     // HIP events around the whole frame
        h->tim.net_ms += last > first ? (double)(last - first) * 1e-5 : 0;  // first conv start .. last conv end
        h->tim.conv_ms += conv_ms;                                       // sum of conv kernel durations
    }
    return VNECT_OK;
}

// pinned staging buffer i with room for `bytes` (grows in 1-MiB steps; a grown buffer moves, so nothing may be in flight)
int ensure_stage(vnect_handle* h, int i, size_t bytes)
{
    if (h->stage_cap[i] >= bytes) return VNECT_OK;
    HIPCK(h, hipStreamSynchronize(h->st));
    if (h->stage[i]) HIPCK(h, hipHostFree(h->stage[i]));
    h->stage[i] = nullptr, h->stage_cap[i] = 0;
    const size_t cap = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    HIPCK(h, hipHostMalloc((void**)&h->stage[i], cap, hipHostMallocMapped));
    HIPCK(h, hipHostGetDevicePointer((void**)&h->stage_dev[i], h->stage[i], 0));
    h->stage_cap[i] = cap;
    return VNECT_OK;
}

// vnect_infer: the frame goes to slot `slot` through pinned memory, asynchronously on the handle's stream (the caller runs the frame
// on that stream next).  The copy is a kernel reading the pinned buffer over PCIe (post.hip: frame_copy_kernel; VNECT_INFER_DMA=1: the
// copy engine instead, hipMemcpyAsync).  VNECT_INFER_SYNC_COPY=1: the round-4 form (a synchronous pageable hipMemcpy2D), for A/B runs.
int stage_frame(vnect_handle* h, int slot, const uint8_t* bgr, int H, int W, int64_t row_stride)
{
    if (!bgr || slot < 0 || slot >= (int)h->slots.size()) return fail(h, VNECT_E_ARG, "bad frame slot");
    if (H < 1 || W < 1 || row_stride < (int64_t)W * 3) return fail(h, VNECT_E_ARG, "bad frame geometry");
    if ((size_t)H * W * 3 > (size_t)h->cfg.max_frame_bytes) return fail(h, VNECT_E_ARG, "frame larger than max_frame_bytes");
    const size_t row = (size_t)W * 3, span = (size_t)(H - 1) * (size_t)row_stride + row;
    const uint8_t *src = nullptr, *src_dev = nullptr;
    size_t src_stride = (size_t)row_stride;
    for (int i = 0; i < 2 && !src; i++)  // already in pinned memory (a crop of a frame the caller captured into vnect_frame_buffer)?
        if (h->stage[i] && bgr >= h->stage[i] && bgr + span <= h->stage[i] + h->stage_cap[i]) src = bgr, src_dev = h->stage_dev[i] + (bgr - h->stage[i]);
    if (!src) {
        const int i = 2;
        int rc = ensure_stage(h, i, (size_t)H * row);
        if (rc) return rc;
        if ((size_t)row_stride == row) memcpy(h->stage[i], bgr, (size_t)H * row);
        else
            for (int y = 0; y < H; y++) memcpy(h->stage[i] + (size_t)y * row, bgr + (size_t)y * (size_t)row_stride, row);
        src = h->stage[i], src_dev = h->stage_dev[i], src_stride = row;
    }
    uint8_t* dst = h->frames + (size_t)slot * h->cfg.max_frame_bytes;
    static const bool dma = getenv("VNECT_INFER_DMA") != nullptr;
    if (!dma) HIPCK(h, launch_frame_copy(src_dev, dst, H, (int)row, (long long)src_stride, h->st));
    else if (src_stride == row) HIPCK(h, hipMemcpyAsync(dst, src, (size_t)H * row, hipMemcpyHostToDevice, h->st));
    else HIPCK(h, hipMemcpy2DAsync(dst, row, src, src_stride, row, H, hipMemcpyHostToDevice, h->st));
    h->slots[slot].H = H, h->slots[slot].W = W, h->slots[slot].stride = (long long)row;
    return VNECT_OK;
}

int upload_frame_impl(vnect_handle* h, int slot, const uint8_t* bgr, int H, int W, int64_t row_stride)
{
    if (!bgr || slot < 0 || slot >= (int)h->slots.size()) return fail(h, VNECT_E_ARG, "bad frame slot");
    if (H < 1 || W < 1 || row_stride < (int64_t)W * 3) return fail(h, VNECT_E_ARG, "bad frame geometry");
    if ((size_t)H * W * 3 > (size_t)h->cfg.max_frame_bytes) return fail(h, VNECT_E_ARG, "frame larger than max_frame_bytes");
    if (h->pre_only && (size_t)H * W * 3 > h->pre_frame_cap) {  // the one slot of a pre-processing-only handle grows with its frames
        HIPCK(h, hipStreamSynchronize(h->st));
        if (h->frames) {
            HIPCK(h, hipFree(h->frames));
            h->dev_allocs.erase(std::find(h->dev_allocs.begin(), h->dev_allocs.end(), (void*)h->frames));
            h->frames = nullptr, h->pre_frame_cap = 0;
        }
        int rc = dev_alloc(h, &h->frames, (size_t)H * W * 3);
        if (rc) return rc;
        h->pre_frame_cap = (size_t)H * W * 3;
    }
    uint8_t* dst = h->frames + (size_t)slot * h->cfg.max_frame_bytes;
    // a frame still being read by an in-flight inference must not be overwritten: wait for that inference only (frames in
    // other slots keep running, so a pipelined caller uploads frame k+1 while frames k and k-1 compute)
    const long long q = h->slots[slot].last_use;
    if (q >= (long long)h->seq_collect) HIPCK(h, hipEventSynchronize(h->done[q % RING]));
    HIPCK(h, hipMemcpy2D(dst, (size_t)W * 3, bgr, (size_t)row_stride, (size_t)W * 3, H, hipMemcpyHostToDevice));
    h->slots[slot].H = H, h->slots[slot].W = W, h->slots[slot].stride = (long long)W * 3;
    return VNECT_OK;
}

// Warm start: a tracking loop wants its FIRST frames at steady-state speed, but the first ~25 frames behind vnect_finalize run 1.5-3 %
// slower (shader clocks ramp up from idle, instruction and translation caches are cold -- measured).  So finalize runs
// the launch plan a few times on a grey 368 x 368 frame in an empty slot (every lane once more, three in flight), then restores the state a fresh
// handle has: empty slot, new filters, no timestamps.  VNECT_PRIME_FRAMES overrides the count (0 = off).  ~25 ms at start-up.
int prime(vnect_handle* h)
{
    const int n_env = getenv("VNECT_PRIME_FRAMES") ? atoi(getenv("VNECT_PRIME_FRAMES")) : 24;  // (read per call: a test flips it inside one process)
    if (n_env <= 0 || h->sharded || (size_t)BOX * BOX * 3 > (size_t)h->cfg.max_frame_bytes) return VNECT_OK;
    int ps = -1;  // an EMPTY frame slot (a caller may have uploaded frames before vnect_finalize: those are not touched)
    for (size_t i = 0; i < h->slots.size() && ps < 0; i++)
        if (h->slots[i].H == 0) ps = (int)i;
    if (ps < 0) return VNECT_OK;
    HIPCK(h, hipMemsetAsync(h->frames + (size_t)ps * h->cfg.max_frame_bytes, 128, (size_t)BOX * BOX * 3, h->st));
    HIPCK(h, hipStreamSynchronize(h->st));
    h->slots[ps].H = BOX, h->slots[ps].W = BOX, h->slots[ps].stride = (long long)BOX * 3;
    int rc = VNECT_OK, ring = 0;
    double t = 1.0;
    // failure injection for the test of this path (tests/test_gpu_surface.py): VNECT_PRIME_INJECT=hip makes the second grey frame fail
    // like a launch error, =state like a benign refusal
    const char* inject = getenv("VNECT_PRIME_INJECT");
    for (int i = 0; i < n_env && !rc; i++, t += 1.0) {
        if (inject && i == 1) {
            rc = fail(h, !strcmp(inject, "hip") ? VNECT_E_HIP : VNECT_E_STATE, std::string("injected warm-start failure (") + inject + ")");
            break;
        }
        rc = enqueue_frame(h, ps, t, t, &ring);
        if (!rc) rc = collect_impl(h, nullptr, nullptr);
    }
    for (int rep = 0; rep < 2 && !rc && !h->twins.empty(); rep++) {  // the other lanes: as many frames in flight as there are lanes
        const int depth = (int)h->twins.size() + 1;
        for (int i = 0; i < depth && !rc; i++, t += 1.0) rc = enqueue_frame(h, ps, t, t, &ring);
        for (int i = 0; i < depth && !rc; i++) rc = collect_impl(h, nullptr, nullptr);
    }
    // A fresh handle's state comes back UNCONDITIONALLY: whatever is still in flight is drained and dropped, the slot is emptied, the
    // filter banks are rebuilt.  What the failure means for vnect_finalize depends on its kind (advisor, round 4):
    //  * VNECT_E_HIP / VNECT_E_INTERNAL / VNECT_E_COMM -- a launch was refused or the device faulted on the launch plan this handle
    //    will run for every real frame: that is a broken plan or a broken device, and vnect_finalize returns the code with the reason;
    //  * anything else (a refused argument or state of the grey frame itself) -- the warm start is an optimisation: skipped, noted in
    //    vnect_last_error, vnect_finalize succeeds.
    const std::string why = rc ? h->err : std::string();
    if (rc) {
        (void)hipStreamSynchronize(h->st);
        for (vnect_handle* tw : h->twins) (void)hipStreamSynchronize(tw->st);
        (void)hipGetLastError();
        h->seq_collect = h->seq_submit;
    }
    h->slots[ps] = vnect_handle::SlotInfo();
    const int rf = reset_filters_impl(h);  // (sets h->err itself when it fails)
    for (int s = 0; s < VNECT_MAX_STREAMS; s++) h->stream_seq[s] = -1, h->stream_lane[s] = nullptr;
    h->fp_dev_valid = false;  // (the next frame uploads its own geometry)
    for (vnect_handle* tw : h->twins) tw->fp_dev_valid = false;
    const bool serious = rc == VNECT_E_HIP || rc == VNECT_E_INTERNAL || rc == VNECT_E_COMM;
    if (rf) {  // the filter banks could not be rebuilt: the handle must not be used; keep both reasons
        if (rc) h->err += "; behind a failed warm start: " + why;
        return rf;
    }
    if (serious) {
        h->err = "warm start failed -- the launch plan or the device is broken: " + why;
        return rc;
    }
    if (rc) h->err = "warm start skipped (not an error of vnect_finalize): " + why;
    return VNECT_OK;
}

}  // namespace

// =====================================================================================================
extern "C" {

int vnect_abi_version(void) { return VNECT_ABI_VERSION; }

const char* vnect_last_error(vnect_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int vnect_create(const vnect_config* cfg, vnect_handle** out)
{
    if (out) *out = nullptr;  // the guard below reports on *out once the handle exists
    return guarded(out, [&]() -> int {
        if (!cfg || !out || cfg->struct_size != (int32_t)sizeof(vnect_config))
            return fail(nullptr, VNECT_E_ARG, "vnect_create: bad config (struct_size mismatch)");
        if (cfg->num_scales < 1 || cfg->num_scales > VNECT_MAX_SCALES)
            return fail(nullptr, VNECT_E_ARG, "vnect_create: num_scales out of range");
        if (cfg->precision != VNECT_FP32 && cfg->precision != VNECT_BF16 && cfg->precision != VNECT_FP32_SPLIT)
            return fail(nullptr, VNECT_E_ARG, "vnect_create: precision must be VNECT_FP32, VNECT_BF16 or VNECT_FP32_SPLIT");
        if (cfg->lanes < 0 || cfg->lanes > RING - 1)
            return fail(nullptr, VNECT_E_ARG, "vnect_create: lanes must be 0 .. 3");
        if (cfg->exchange != VNECT_XCHG_RCCL && cfg->exchange != VNECT_XCHG_P2P)
            return fail(nullptr, VNECT_E_ARG, "vnect_create: exchange must be VNECT_XCHG_RCCL or VNECT_XCHG_P2P");
        if (cfg->preprocess_only && cfg->pyramid_nranks > 0)
            return fail(nullptr, VNECT_E_ARG, "vnect_create: preprocess_only and pyramid sharding exclude each other");
        roctx_load();
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || cfg->device < 0 || cfg->device >= ndev)
            return fail(nullptr, VNECT_E_NODEVICE, "vnect_create: no HIP device " + std::to_string(cfg->device));
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess || !strstr(prop.gcnArchName, "gfx950"))
            return fail(nullptr, VNECT_E_NODEVICE, "vnect_create: device is not gfx950 (MI355X); kernels are built for gfx950 only");
        const bool sharded = cfg->pyramid_nranks > 1 || (cfg->pyramid_nranks == 1 && cfg->num_scales == 1);
        if (sharded && (cfg->pyramid_nranks != cfg->num_scales || cfg->pyramid_rank < 0 || cfg->pyramid_rank >= cfg->pyramid_nranks))
            return fail(nullptr, VNECT_E_ARG, "vnect_create: pyramid sharding needs pyramid_nranks == num_scales and 0 <= pyramid_rank < nranks");
        vnect_handle* h = new vnect_handle();
        h->cfg = *cfg;
        h->S = cfg->num_scales;
        h->Snet = sharded ? 1 : cfg->num_scales;
        h->bf16 = cfg->precision == VNECT_BF16;
        h->x3 = cfg->precision == VNECT_FP32_SPLIT;
        h->sharded = sharded;
        h->keep_activations = cfg->keep_activations != 0;
        const bool pre = cfg->preprocess_only != 0;
        h->pre_only = pre;
        if (h->cfg.max_frame_bytes <= 0) h->cfg.max_frame_bytes = 4096 * 4096 * 3;
        if (h->cfg.max_frame_bytes < INT32_MAX - 16) h->cfg.max_frame_bytes = (h->cfg.max_frame_bytes + 15) & ~15;  // slots start 16-byte aligned (the stem reads frame rows as aligned dwords)
        if (h->cfg.num_frame_slots <= 0) h->cfg.num_frame_slots = 4;
        // gen_input_batch alone (the static method creates and destroys such a handle per call): ONE frame slot that grows with the
        // frames it is given (upload_frame_impl), no gather buffer, no filter bank, no profiling buffers -- the resize tables and the
        // (S,368,368) batch + its read-back staging are all it owns (~20 MB at S = 3)
        if (pre) h->cfg.num_frame_slots = 1;
        *out = h;  // returned even on failure below so the caller can read the message, then destroy
        HIPCK(h, hipSetDevice(cfg->device));
        HIPCK(h, hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
        HIPCK(h, conv_setup());
        HIPCK(h, stem_setup());
        h->slots.resize(h->cfg.num_frame_slots);
        int rc;
        if (!pre && (rc = dev_alloc(h, &h->frames, (size_t)h->cfg.num_frame_slots * h->cfg.max_frame_bytes + 16))) return rc;  // + slack: a dword read may run 3 bytes past a frame's last pixel
        if ((rc = dev_alloc(h, &h->d_fp, 1))) return rc;
        if ((rc = dev_alloc(h, &h->d_stabs, 1))) return rc;
        if ((rc = dev_alloc(h, &h->d_part, (size_t)NJ * ARG_SLABS_MAX))) return rc;
        if ((rc = dev_alloc(h, &h->d_ticket, 4))) return rc;
        HIPCK(h, hipMemset(h->d_ticket, 0, 4 * sizeof(unsigned)));
        h->post_merged = getenv("VNECT_NO_POST_MERGE") == nullptr;
        if (!pre && (rc = dev_alloc(h, &h->d_fb, VNECT_MAX_STREAMS))) return rc;
        if ((rc = dev_alloc(h, &h->in3, (size_t)(pre ? h->Snet : VNECT_MAX_SCALES) * BOX * BOX * 3))) return rc;
        if (!pre && (rc = dev_alloc(h, &h->gather, (size_t)VNECT_MAX_SCALES * HM * HM * MAPC))) return rc;
        for (int i = 0; i < RING; i++) {
            HIPCK(h, hipHostMalloc((void**)&h->h_fp[i], sizeof(FrameParams), hipHostMallocDefault));
            HIPCK(h, hipHostMalloc((void**)&h->h_out[i], sizeof(JointsOut), hipHostMallocMapped | hipHostMallocCoherent));
            HIPCK(h, hipHostGetDevicePointer((void**)&h->h_out_dev[i], h->h_out[i], 0));
            HIPCK(h, hipEventCreateWithFlags(&h->done[i], hipEventDisableTiming));
        }
        if (sharded && cfg->exchange == VNECT_XCHG_P2P) {
            // fine-grained device memory: peers' stores and the system-scope loads of this device bypass its L2, so a slot is
            // never served from a line cached two frames ago.  The protocol DEPENDS on it (exchange_kernel polls flags and reads slots
            // that a peer writes over xGMI): no silent fall-back to coarse-grained memory, whose stale flags / slots would
            // time out or -- worse -- merge old maps.
            void* q = nullptr;
            hipError_t e = hipExtMallocWithFlags(&q, XCHG_BYTES, hipDeviceMallocFinegrained);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return fail(h, VNECT_E_COMM, std::string("exchange = VNECT_XCHG_P2P needs fine-grained device memory "
                                                         "(hipExtMallocWithFlags(hipDeviceMallocFinegrained): ") +
                                                 hipGetErrorString(e) + "); use VNECT_XCHG_RCCL");
            }
            h->dev_allocs.push_back(q);
            h->xblock = (char*)q;
            HIPCK(h, hipMemset(h->xblock, 0, XCHG_BYTES));
            if ((rc = dev_alloc(h, &h->xtickets, 8))) return rc;
            HIPCK(h, hipMemset(h->xtickets, 0, 8 * sizeof(unsigned)));
            HIPCK(h, hipHostMalloc((void**)&h->h_xstatus, RING * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
            for (int i = 0; i < RING; i++) h->h_xstatus[i] = 0;
            if ((rc = dev_alloc(h, &h->d_xfail, 4))) return rc;
            HIPCK(h, hipMemset(h->d_xfail, 0, 4 * sizeof(unsigned)));
            HIPCK(h, hipHostGetDevicePointer((void**)&h->h_xstatus_dev, h->h_xstatus, 0));
            h->xpeer[cfg->pyramid_rank] = h->xblock;
        }
        HIPCK(h, hipHostMalloc((void**)&h->h_filt, 128 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        HIPCK(h, hipHostGetDevicePointer((void**)&h->h_filt_dev, h->h_filt, 0));
        for (auto& e : h->ev) HIPCK(h, hipEventCreate(&e));
        if (!pre) {
            if ((rc = dev_alloc(h, &h->d_prof, PROF_SLOTS * 128))) return rc;
            if ((rc = dev_alloc(h, &h->d_prof_end, (size_t)PROF_WGS * 128))) return rc;
            HIPCK(h, hipMemset(h->d_prof_end, 0, (size_t)PROF_WGS * 128 * sizeof(unsigned long long)));
            HIPCK(h, hipHostMalloc((void**)&h->h_prof_end, (size_t)PROF_WGS * 128 * sizeof(unsigned long long), hipHostMallocDefault));
            HIPCK(h, hipHostMalloc((void**)&h->h_prof, PROF_SLOTS * 128 * sizeof(unsigned long long), hipHostMallocDefault));
            for (int i = 0; i < PROF_SLOTS * 128; i++) h->h_prof[i] = 0;
        }
        if ((rc = build_scale_tables(h))) return rc;
        if ((rc = build_up_table(h))) return rc;
        if (!pre && (rc = reset_filters_impl(h))) return rc;
        h->tim.struct_size = sizeof(vnect_timings);
        if (pre) {
            // gen_input_batch alone (the reference's static method needs no session either, estimator.py:70-81): the input
            // batch buffer and the tables made above; no weights, no launch plan, vnect_finalize is refused
            h->t_input4 = add_tensor(h, "input", h->Snet, BOX, BOX, 3, 4);
            Tensor& t = h->tensors[h->t_input4];
            char* p = nullptr;
            if ((rc = dev_alloc(h, &p, t.bytes() + 256))) return rc;
            t.d = (float*)p;
        }
        return VNECT_OK;
    });
}

void vnect_destroy(vnect_handle* h)
{
    if (!h) return;
    hipSetDevice(h->cfg.device);
    if (h->st) hipStreamSynchronize(h->st);
    destroy_twins(h);
    if (h->comm && p_ncclCommDestroy) p_ncclCommDestroy(h->comm);
    if (h->gexec) hipGraphExecDestroy(h->gexec);
    if (h->graph) hipGraphDestroy(h->graph);
    if (h->pgexec) hipGraphExecDestroy(h->pgexec);
    if (h->pgraph) hipGraphDestroy(h->pgraph);
    for (int i = 0; i < RING; i++) {
        if (h->h_fp[i]) hipHostFree(h->h_fp[i]);
        if (h->h_out[i]) hipHostFree(h->h_out[i]);
        if (h->done[i]) hipEventDestroy(h->done[i]);
    }
    for (auto& e : h->ev)
        if (e) hipEventDestroy(e);
    for (int r = 0; r < VNECT_MAX_SCALES; r++)
        if (h->xopened[r] && h->xpeer[r]) hipIpcCloseMemHandle(h->xpeer[r]);
    if (h->h_xstatus) hipHostFree(h->h_xstatus);
    for (int i = 0; i < 3; i++)
        if (h->stage[i]) hipHostFree(h->stage[i]);
    if (h->h_filt) hipHostFree(h->h_filt);
    if (h->h_prof) hipHostFree(h->h_prof);
    if (h->h_prof_end) hipHostFree(h->h_prof_end);
    for (void* p : h->dev_allocs) hipFree(p);
    if (h->st) hipStreamDestroy(h->st);
    delete h;
}

int vnect_set_weight(vnect_handle* h, const char* name, const float* data, const int64_t* shape, int ndim)
{
    return guarded(&h, [&]() -> int {
        if (!h) return VNECT_E_ARG;
        if (!name || !data || !shape || ndim < 1 || ndim > 4) return fail(h, VNECT_E_ARG, "vnect_set_weight: bad argument");
        if (h->finalized) return fail(h, VNECT_E_STATE, "vnect_set_weight after vnect_finalize");
        HostArray a;
        size_t n = 1;
        for (int i = 0; i < ndim; i++) {
            if (shape[i] < 1) return fail(h, VNECT_E_ARG, "vnect_set_weight: bad shape");
            a.shape.push_back(shape[i]);
            n *= (size_t)shape[i];
        }
        a.d.assign(data, data + n);
        h->weights[name] = std::move(a);
        return VNECT_OK;
    });
}

int vnect_finalize(vnect_handle* h)
{
    return guarded(&h, [&]() -> int {
        if (!h) return VNECT_E_ARG;
        if (h->finalized) return fail(h, VNECT_E_STATE, "already finalized");
        if (h->pre_only) return fail(h, VNECT_E_STATE, "vnect_finalize on a preprocess_only handle");
        HIPCK(h, hipSetDevice(h->cfg.device));
        int rc = finalize_impl(h);
        if (rc) {
            if (h->err.empty()) h->err = "finalize failed";
            return rc;
        }
        rc = build_graph(h);
        if (rc) return rc;
        rc = build_twins(h);
        if (rc) return rc;
        HIPCK(h, hipStreamSynchronize(h->st));
        h->finalized = true;
        h->weights.clear();
        return prime(h);
    });
}

int vnect_set_scales(vnect_handle* h, const double* scales, int n)
{
    return guarded(&h, [&]() -> int {
        if (!h || !scales) return VNECT_E_ARG;
        if (n != h->S) return fail(h, VNECT_E_ARG, "vnect_set_scales: the number of scales is fixed at create time");
        if (h->seq_submit != h->seq_collect) return fail(h, VNECT_E_STATE, "frames in flight");
        HIPCK(h, hipSetDevice(h->cfg.device));
        HIPCK(h, hipStreamSynchronize(h->st));
        for (vnect_handle* t : h->twins) HIPCK(h, hipStreamSynchronize(t->st));
        double old[VNECT_MAX_SCALES];
        memcpy(old, h->cfg.scales, sizeof old);
        for (int i = 0; i < n; i++) h->cfg.scales[i] = scales[i];
        int rc = build_scale_tables(h);
        if (rc) {
            memcpy(h->cfg.scales, old, sizeof old);
            build_scale_tables(h);
        }
        // the two-launch form of the post-processing (VNECT_NO_POST_MERGE=1) has its arg-max launch -- and with it the merge geometry,
        // a by-value kernel argument -- inside the captured graph: capture again (the default form launches post_kernel eagerly)
        if (h->finalized && !h->sharded && !h->post_merged && h->gexec) {
            int rg = build_graph(h);
            for (vnect_handle* t : h->twins)
                if (!rg && (rg = build_graph(t))) fail(h, rg, t->err);
            if (rg) return rg;
        }
        return rc;
    });
}

int vnect_forward(vnect_handle* h, const float* batch, int num_images, float* out)
{
    return guarded(&h, [&]() -> int {
        if (!h || !batch || !out) return VNECT_E_ARG;
        if (!h->finalized) return fail(h, VNECT_E_STATE, "vnect_forward before vnect_finalize");
        if (h->seq_submit != h->seq_collect) return fail(h, VNECT_E_STATE, "frames in flight");
        if (num_images != h->Snet)
            return fail(h, VNECT_E_ARG, "vnect_forward: num_images must equal num_scales (1 on a pyramid-sharded handle)");
        HIPCK(h, hipSetDevice(h->cfg.device));
        const long long npix = (long long)h->Snet * BOX * BOX;
        HIPCK(h, hipMemcpyAsync(h->in3, batch, npix * 3 * sizeof(float), hipMemcpyHostToDevice, h->st));
        HIPCK(h, launch_pad3to4(h->in3, h->tensors[h->t_input4].d, npix, h->bf16, h->st));
        int rc = run_network(h, false);
        if (rc) return rc;
        const Tensor& t = h->tensors[h->t_out];
        HIPCK(h, hipMemcpyAsync(out, t.d, t.bytes(), hipMemcpyDeviceToHost, h->st));
        HIPCK(h, hipStreamSynchronize(h->st));
        return VNECT_OK;
    });
}

int vnect_preprocess(vnect_handle* h, const uint8_t* bgr, int H, int W, int64_t row_stride, float* batch_out,
                     double* scaler, int32_t* offset_x, int32_t* offset_y)
{
    return guarded(&h, [&]() -> int {
        if (!h || !bgr) return VNECT_E_ARG;
        if (!h->finalized && !h->pre_only) return fail(h, VNECT_E_STATE, "vnect_preprocess before vnect_finalize");
        if (h->seq_submit != h->seq_collect) return fail(h, VNECT_E_STATE, "frames in flight");
        HIPCK(h, hipSetDevice(h->cfg.device));
        int rc = upload_frame_impl(h, 0, bgr, H, W, row_stride);
        if (rc) return rc;
        FrameParams fp;
        if ((rc = squarify_params(h, H, W, &fp))) return rc;
        FrameDyn dyn{};
        dyn.row_stride = h->slots[0].stride, dyn.frame = h->frames;
        if ((rc = sync_geometry(h, fp))) return rc;
        if ((rc = run_pre(h, dyn, false, true))) return rc;
        if (batch_out) {
            const long long npix = (long long)h->Snet * BOX * BOX;
            HIPCK(h, launch_strip4to3(h->tensors[h->t_input4].d, h->in3, npix, h->bf16, h->st));
            HIPCK(h, hipMemcpyAsync(batch_out, h->in3, npix * 3 * sizeof(float), hipMemcpyDeviceToHost, h->st));
        }
        HIPCK(h, hipStreamSynchronize(h->st));
        if (scaler) *scaler = fp.scaler;
        if (offset_x) *offset_x = fp.offx;
        if (offset_y) *offset_y = fp.offy;
        return VNECT_OK;
    });
}

int vnect_postprocess(vnect_handle* h, const float* maps, double t2d, double t3d, double scaler, int32_t offset_x,
                      int32_t offset_y, double* j2, float* j3)
{
    return guarded(&h, [&]() -> int {
        if (!h || !maps || !j2 || !j3) return VNECT_E_ARG;
        if (!h->finalized) return fail(h, VNECT_E_STATE, "vnect_postprocess before vnect_finalize");
        if (h->seq_submit != h->seq_collect) return fail(h, VNECT_E_STATE, "frames in flight");
        if (!(scaler > 0)) return fail(h, VNECT_E_ARG, "scaler must be positive");
        HIPCK(h, hipSetDevice(h->cfg.device));
        int rc = check_time(h, t2d, t3d);
        if (rc) return rc;
        float* dst = h->sharded ? h->gather : h->tensors[h->t_out].d;
        HIPCK(h, hipMemcpyAsync(dst, maps, (size_t)h->S * HM * HM * MAPC * sizeof(float), hipMemcpyHostToDevice, h->st));
        FrameParams fp;
        memset(&fp, 0, sizeof fp);
        fp.scaler = scaler, fp.offx = offset_x, fp.offy = offset_y;
        FrameDyn dyn{};
        dyn.t2d = t2d, dyn.t3d = t3d;
        if ((rc = sync_geometry(h, fp))) return rc;
        if (h->post_merged) {
            if ((rc = run_post(h, dyn, h->h_out_dev[0]))) return rc;
        } else {
            if ((rc = run_argmax(h))) return rc;
            if ((rc = run_joints(h, dyn, h->h_out_dev[0]))) return rc;
        }
        commit_time(h, t2d, t3d);
        HIPCK(h, hipStreamSynchronize(h->st));
        memcpy(j2, h->h_out[0]->j2d, sizeof(double) * NJ * 2);
        memcpy(j3, h->h_out[0]->j3d, sizeof(float) * NJ * 3);
        return VNECT_OK;
    });
}

int vnect_frame_buffer(vnect_handle* h, int index, int64_t min_bytes, uint8_t** ptr_out)
{
    return guarded(&h, [&]() -> int {
        if (!h || !ptr_out || index < 0 || index > 1 || min_bytes < 1) return h ? fail(h, VNECT_E_ARG, "vnect_frame_buffer: bad argument") : VNECT_E_ARG;
        if (h->pre_only) return fail(h, VNECT_E_STATE, "vnect_frame_buffer on a preprocess_only handle");
        if (min_bytes > (int64_t)h->cfg.max_frame_bytes) return fail(h, VNECT_E_ARG, "vnect_frame_buffer: larger than max_frame_bytes");
        if (h->seq_submit != h->seq_collect) return fail(h, VNECT_E_STATE, "frames in flight");
        HIPCK(h, hipSetDevice(h->cfg.device));
        int rc = ensure_stage(h, index, (size_t)min_bytes);
        if (rc) return rc;
        *ptr_out = h->stage[index];
        return VNECT_OK;
    });
}

int vnect_upload_frame(vnect_handle* h, int slot, const uint8_t* bgr, int H, int W, int64_t row_stride)
{
    return guarded(&h, [&]() -> int {
        if (!h) return VNECT_E_ARG;
        HIPCK(h, hipSetDevice(h->cfg.device));
        return upload_frame_impl(h, slot, bgr, H, W, row_stride);
    });
}

int vnect_submit_resident(vnect_handle* h, int slot, double t2d, double t3d)
{
    return guarded(&h, [&]() -> int {
        if (!h) return VNECT_E_ARG;
        if (!h->finalized) return fail(h, VNECT_E_STATE, "inference before vnect_finalize");
        HIPCK(h, hipSetDevice(h->cfg.device));
        int ring;
        return enqueue_frame(h, slot, t2d, t3d, &ring);
    });
}

int vnect_submit_stream(vnect_handle* h, int stream, int slot, double t2d, double t3d)
{
    return guarded(&h, [&]() -> int {
        if (!h) return VNECT_E_ARG;
        if (!h->finalized) return fail(h, VNECT_E_STATE, "inference before vnect_finalize");
        HIPCK(h, hipSetDevice(h->cfg.device));
        int ring;
        return enqueue_frame(h, slot, t2d, t3d, &ring, stream);
    });
}

int vnect_collect_stream(vnect_handle* h, int32_t* stream_out, double* j2, float* j3)
{
    return guarded(&h, [&]() -> int {
        if (!h) return VNECT_E_ARG;
        HIPCK(h, hipSetDevice(h->cfg.device));
        return collect_impl(h, j2, j3, stream_out);
    });
}

int vnect_reset_filters_stream(vnect_handle* h, int stream)
{
    return guarded(&h, [&]() -> int {
        if (!h) return VNECT_E_ARG;
        if (stream < 0 || stream >= VNECT_MAX_STREAMS) return fail(h, VNECT_E_ARG, "stream out of range");
        if (h->seq_submit != h->seq_collect) return fail(h, VNECT_E_STATE, "frames in flight");
        if (h->pre_only) return fail(h, VNECT_E_STATE, "vnect_reset_filters_stream on a preprocess_only handle (it has no filter bank)");
        HIPCK(h, hipSetDevice(h->cfg.device));
        return reset_filters_impl(h, stream);
    });
}

int vnect_collect(vnect_handle* h, double* j2, float* j3)
{
    return guarded(&h, [&]() -> int {
        if (!h) return VNECT_E_ARG;
        HIPCK(h, hipSetDevice(h->cfg.device));
        return collect_impl(h, j2, j3);
    });
}

int vnect_infer_resident(vnect_handle* h, int slot, double t2d, double t3d, double* j2, float* j3)
{
    return guarded(&h, [&]() -> int {
        if (!h || !j2 || !j3) return VNECT_E_ARG;
        if (!h->finalized) return fail(h, VNECT_E_STATE, "inference before vnect_finalize");
        if (h->seq_submit != h->seq_collect) return fail(h, VNECT_E_STATE, "frames in flight");
        int rc = vnect_submit_resident(h, slot, t2d, t3d);
        if (rc) return rc;
        return collect_impl(h, j2, j3);
    });
}

int vnect_infer(vnect_handle* h, const uint8_t* bgr, int H, int W, int64_t row_stride, double t2d, double t3d,
                double* j2, float* j3)
{
    return guarded(&h, [&]() -> int {
        if (!h || !bgr || !j2 || !j3) return VNECT_E_ARG;
        if (!h->finalized) return fail(h, VNECT_E_STATE, "vnect_infer before vnect_finalize");
        if (h->seq_submit != h->seq_collect) return fail(h, VNECT_E_STATE, "frames in flight");
        HIPCK(h, hipSetDevice(h->cfg.device));
        // nothing is in flight, so the frame will run on the first lane's stream (enqueue_frame), the stream stage_frame copies on
        static const bool sync_copy = getenv("VNECT_INFER_SYNC_COPY") != nullptr;
        int rc = sync_copy ? upload_frame_impl(h, 0, bgr, H, W, row_stride) : stage_frame(h, 0, bgr, H, W, row_stride);
        if (rc) return rc;
        return vnect_infer_resident(h, 0, t2d, t3d, j2, j3);
    });
}

int vnect_joint_filter(vnect_handle* h, int dim, const double* joints_in, int values_are_f32, double t, double* joints_out)
{
    return guarded(&h, [&]() -> int {
        if (!h || !joints_in || !joints_out || (dim != 2 && dim != 3)) return h ? fail(h, VNECT_E_ARG, "vnect_joint_filter: bad argument") : VNECT_E_ARG;
        if (h->pre_only) return fail(h, VNECT_E_STATE, "vnect_joint_filter on a preprocess_only handle (it has no filter bank)");
        if (h->seq_submit != h->seq_collect) return fail(h, VNECT_E_STATE, "frames in flight");
        HIPCK(h, hipSetDevice(h->cfg.device));
        // the timestamp rules of check_time, for the one bank this call advances
        const bool have = dim == 2 ? h->have2[0] : h->have3[0];   // (stream 0: the bank vnect_infer advances)
        const double last = dim == 2 ? h->last2[0] : h->last3[0];
        if (have && last != 0.0 && t != 0.0) {
            if (t == last) return fail(h, VNECT_E_TIMESTAMP, "timestamp equals the previous one of this filter bank");
            if (t < last) return fail(h, VNECT_E_TIMEORDER, "timestamp is earlier than the previous one of this filter bank");
        }
        const int n = NJ * dim;
        memcpy(h->h_filt, joints_in, sizeof(double) * n);
        HIPCK(h, launch_filter(h->d_fb, dim, values_are_f32 != 0, h->cfg.numpy_promotion, t, h->h_filt_dev, h->h_filt_dev + 64, h->st));
        HIPCK(h, hipStreamSynchronize(h->st));
        if (dim == 2) h->have2[0] = true, h->last2[0] = t;
        else h->have3[0] = true, h->last3[0] = t;
        memcpy(joints_out, h->h_filt + 64, sizeof(double) * n);
        return VNECT_OK;
    });
}

int vnect_reset_filters(vnect_handle* h)
{
    return guarded(&h, [&]() -> int {
        if (!h) return VNECT_E_ARG;
        if (h->seq_submit != h->seq_collect) return fail(h, VNECT_E_STATE, "frames in flight");
        if (h->pre_only) return fail(h, VNECT_E_STATE, "vnect_reset_filters on a preprocess_only handle (it has no filter bank)");
        HIPCK(h, hipSetDevice(h->cfg.device));
        return reset_filters_impl(h);
    });
}

int vnect_read_activation(vnect_handle* h, const char* name, float* out, int64_t capacity, int32_t* shape4)
{
    return guarded(&h, [&]() -> int {
        if (!h || !name || !shape4) return VNECT_E_ARG;
        if (!h->finalized) return fail(h, VNECT_E_STATE, "not finalized");
        auto it = h->tensor_by_name.find(name);
        if (it == h->tensor_by_name.end()) return fail(h, VNECT_E_ARG, std::string("no activation named ") + name);
        const Tensor& t = h->tensors[it->second];
        shape4[0] = t.S, shape4[1] = t.H, shape4[2] = t.W, shape4[3] = t.C;
        if (!out) return VNECT_OK;
        if (!h->keep_activations && it->second != h->t_out)
            return fail(h, VNECT_E_STATE, "vnect_read_activation: inner layers share an arena; create the handle with keep_activations = 1");
        const size_t npix = (size_t)t.S * t.H * t.W;
        if ((int64_t)(npix * t.C) > capacity) return fail(h, VNECT_E_ARG, "vnect_read_activation: capacity too small");
        HIPCK(h, hipSetDevice(h->cfg.device));
        HIPCK(h, hipStreamSynchronize(h->st));
        if (t.esz == 4) {
            HIPCK(h, hipMemcpy2D(out, (size_t)t.C * sizeof(float), t.d, (size_t)t.Cs * sizeof(float), (size_t)t.C * sizeof(float),
                                 npix, hipMemcpyDeviceToHost));
        } else {  // bf16 activations: fetch raw, widen on the host
            std::vector<uint16_t> raw(npix * t.C);
            HIPCK(h, hipMemcpy2D(raw.data(), (size_t)t.C * 2, t.d, (size_t)t.Cs * 2, (size_t)t.C * 2, npix, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < raw.size(); i++) out[i] = from_bf16(raw[i]);
        }
        return VNECT_OK;
    });
}

int vnect_get_layer_stamps(vnect_handle* h, int idx, uint64_t* out24)
{
    return guarded(&h, [&]() -> int {
        if (!h || !out24 || idx < 0 || idx >= (int)h->layers.size()) return h ? fail(h, VNECT_E_ARG, "bad layer index") : VNECT_E_ARG;
        for (int k = 0; k < PROF_SLOTS; k++) out24[k] = h->h_prof[PROF_SLOTS * idx + k];
        return VNECT_OK;
    });
}

int vnect_set_profiling(vnect_handle* h, int on)
{
    return guarded(&h, [&]() -> int {
        if (!h) return VNECT_E_ARG;
        h->profiling = on != 0;
        return VNECT_OK;
    });
}

int vnect_get_timings(vnect_handle* h, vnect_timings* out)
{
    return guarded(&h, [&]() -> int {
        if (!h || !out || out->struct_size != (int32_t)sizeof(vnect_timings)) return VNECT_E_ARG;
        *out = h->tim;
        out->struct_size = sizeof(vnect_timings);
        out->conv_launches = h->conv_launches;
        out->conv_flops = h->conv_flops;
        return VNECT_OK;
    });
}

int vnect_reset_timings(vnect_handle* h)
{
    return guarded(&h, [&]() -> int {
        if (!h) return VNECT_E_ARG;
        memset(&h->tim, 0, sizeof h->tim);
        h->tim.struct_size = sizeof(vnect_timings);
        return VNECT_OK;
    });
}

int vnect_get_layer_info(vnect_handle* h, int idx, vnect_layer_info* out)
{
    return guarded(&h, [&]() -> int {
        if (!h || !out || idx < 0 || idx >= (int)h->layers.size()) return VNECT_E_ARG;
        const Layer& L = h->layers[idx];
        memset(out, 0, sizeof *out);
        snprintf(out->name, sizeof out->name, "%s", L.name.c_str());
        if (L.op == OP_CONV) {
            out->M = L.a.M * L.a.nphase, out->N = L.Nreal, out->K = L.Kreal;
            out->tile_m = L.BM, out->tile_n = L.BN, out->split_k = L.a.ksplit;
            out->workgroups = ((L.a.M + L.BM - 1) / L.BM) * (L.a.Npad / L.BN) * L.a.nphase * L.a.ksplit;
            out->flops = L.flops;
        }
        out->last_ms = L.last_ms;
        return VNECT_OK;
    });
}

int vnect_comm_unique_id(void* id128)
{
    return guarded(nullptr, [&]() -> int {
        if (!id128) return VNECT_E_ARG;
        if (!load_rccl()) return fail(nullptr, VNECT_E_COMM, "librccl.so not available");
        nccl_uid u;
        if (p_ncclGetUniqueId(&u) != 0) return fail(nullptr, VNECT_E_COMM, "ncclGetUniqueId failed");
        memcpy(id128, &u, sizeof u);
        return VNECT_OK;
    });
}

int vnect_comm_init(vnect_handle* h, int rank, int nranks, const void* id128)
{
    return guarded(&h, [&]() -> int {
        if (!h || !id128) return VNECT_E_ARG;
        if (!h->sharded) return fail(h, VNECT_E_STATE, "vnect_comm_init: handle was not created with pyramid_nranks");
        if (h->comm) return fail(h, VNECT_E_STATE, "vnect_comm_init: communicator already initialised");
        if (nranks != h->cfg.pyramid_nranks || rank != h->cfg.pyramid_rank)
            return fail(h, VNECT_E_ARG, "vnect_comm_init: rank / nranks differ from the handle's pyramid configuration");
        if (!load_rccl()) return fail(h, VNECT_E_COMM, "librccl.so not available");
        HIPCK(h, hipSetDevice(h->cfg.device));
        nccl_uid u;
        memcpy(&u, id128, sizeof u);
        const int rc = p_ncclCommInitRank(&h->comm, nranks, u, rank);
        if (rc != 0) {
            h->comm = nullptr;
            return fail(h, VNECT_E_COMM, std::string("ncclCommInitRank: ") + (p_ncclGetErrorString ? p_ncclGetErrorString(rc) : "error"));
        }
        return VNECT_OK;
    });
}

int vnect_comm_library(char* path_out, int capacity, int32_t* reused_out)
{
    return guarded(nullptr, [&]() -> int {
        if (!path_out || capacity < 2) return VNECT_E_ARG;
        if (!load_rccl()) return fail(nullptr, VNECT_E_COMM, "librccl.so not available");
        snprintf(path_out, (size_t)capacity, "%s", g_rccl_path.c_str());
        if (reused_out) *reused_out = g_rccl_reused ? 1 : 0;
        return VNECT_OK;
    });
}

/* blob layout (128 bytes): [0,64) hipIpcMemHandle_t of the exchange block, [64,72) its address in the exporting process,
 * [72,80) that process's pid, [80,84) its device ordinal */
int vnect_comm_p2p_export(vnect_handle* h, void* blob128)
{
    return guarded(&h, [&]() -> int {
        if (!h || !blob128) return VNECT_E_ARG;
        if (!h->sharded || h->cfg.exchange != VNECT_XCHG_P2P) return fail(h, VNECT_E_STATE, "vnect_comm_p2p_export: handle was not created with exchange = VNECT_XCHG_P2P");
        HIPCK(h, hipSetDevice(h->cfg.device));
        char* b = (char*)blob128;
        memset(b, 0, 128);
        hipIpcMemHandle_t ipc;
        static_assert(sizeof(ipc) <= 64, "blob layout");
        hipError_t e = hipIpcGetMemHandle(&ipc, h->xblock);
        if (e == hipSuccess) memcpy(b, &ipc, sizeof ipc);
        else (void)hipGetLastError();  // still usable inside this process (the raw address below)
        const unsigned long long addr = (unsigned long long)(uintptr_t)h->xblock, pid = (unsigned long long)getpid();
        const int dev = h->cfg.device, has_ipc = e == hipSuccess;
        memcpy(b + 64, &addr, 8), memcpy(b + 72, &pid, 8), memcpy(b + 80, &dev, 4), memcpy(b + 84, &has_ipc, 4);
        return VNECT_OK;
    });
}

int vnect_comm_p2p_init(vnect_handle* h, int rank, int nranks, const void* blobs)
{
    return guarded(&h, [&]() -> int {
        if (!h || !blobs) return VNECT_E_ARG;
        if (!h->sharded || h->cfg.exchange != VNECT_XCHG_P2P) return fail(h, VNECT_E_STATE, "vnect_comm_p2p_init: handle was not created with exchange = VNECT_XCHG_P2P");
        if (h->p2p_ready) return fail(h, VNECT_E_STATE, "vnect_comm_p2p_init: peers already connected");
        if (nranks != h->cfg.pyramid_nranks || rank != h->cfg.pyramid_rank)
            return fail(h, VNECT_E_ARG, "vnect_comm_p2p_init: rank / nranks differ from the handle's pyramid configuration");
        HIPCK(h, hipSetDevice(h->cfg.device));
        for (int r = 0; r < nranks; r++) {
            if (r == rank) continue;
            const char* b = (const char*)blobs + (size_t)r * 128;
            unsigned long long addr, pid;
            int dev, has_ipc;
            memcpy(&addr, b + 64, 8), memcpy(&pid, b + 72, 8), memcpy(&dev, b + 80, 4), memcpy(&has_ipc, b + 84, 4);
            if (pid == (unsigned long long)getpid()) {
                // a peer handle of this very process (several GPUs driven by one process, or the one-GPU test): its address is
                // valid here; make the other device reachable
                if (dev != h->cfg.device) {
                    hipError_t e = hipDeviceEnablePeerAccess(dev, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                        return fail(h, VNECT_E_COMM, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                    (void)hipGetLastError();
                }
                h->xpeer[r] = (char*)(uintptr_t)addr;
            } else {
                if (!has_ipc) return fail(h, VNECT_E_COMM, "vnect_comm_p2p_init: peer could not export its exchange block (hipIpcGetMemHandle failed there)");
                hipIpcMemHandle_t ipc;
                memcpy(&ipc, b, sizeof ipc);
                void* q = nullptr;
                hipError_t e = hipIpcOpenMemHandle(&q, ipc, hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess) return fail(h, VNECT_E_COMM, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e));
                h->xpeer[r] = (char*)q, h->xopened[r] = true;
            }
        }
        h->p2p_ready = true;
        return VNECT_OK;
    });
}

}  // extern "C"
