// rt_exec.cpp -- running frames on the plan: launch sequences, the hipGraph and its profiling twin, lanes, video streams, submit /
// collect, host -> device staging, the warm start.  One call of VNectEstimator.__call__ (/root/reference/src/estimator.py:97-142)
// is enqueue_frame + collect_impl.
#include "runtime.h"

namespace vnect {
namespace rt {

// ---- launch sequences -------------------------------------------------------------------------------
// `stem_done`: the caller has launched the stem from the frame already (enqueue_frame: it takes the frame's arguments by value)
int run_network(vnect_handle* h, bool timed, bool stem_done)
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

int run_pre(vnect_handle* h, const FrameDyn& dyn, bool timed, bool want_batch)
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
int run_joints(vnect_handle* h, const FrameDyn& dyn, JointsOut* out, int stream)
{
    const float* maps = h->sharded ? h->gather : h->tensors[h->t_out].d;
    HIPCK(h, launch_joints(h->d_part, maps, h->mgeo, h->d_fb + stream, h->d_fp, dyn, h->cfg.numpy_promotion, out, h->st));
    return VNECT_OK;
}

// both in one launch (post.hip: post_kernel): takes the frame's arguments by value, so it runs behind the graph, not inside it
int run_post(vnect_handle* h, const FrameDyn& dyn, JointsOut* out, int stream)
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
int check_time(vnect_handle* h, double t2d, double t3d, int s)
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
void commit_time(vnect_handle* h, double t2d, double t3d, int s)
{
    h->have2[s] = h->have3[s] = true, h->last2[s] = t2d, h->last3[s] = t3d;
}

int reset_filters_impl(vnect_handle* h, int stream)  // -1: every stream
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
static struct Roctx {
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

// The part of a frame without per-frame arguments (what the hipGraph holds): conv stack, merge + arg-max.  The pyramid kernel
// before it and the joints kernel after it take the frame's arguments by value and are launched around the graph.  A
// pyramid-sharded handle's graph ends with the conv stack: the exchange (whose flag value / parity change every frame) and the
// merge + arg-max launch behind it are eager (two captured parts around the exchange -- the
// second part is the single arg-max launch, which gains nothing from a graph of its own).
static int run_frame_kernels(vnect_handle* h, bool timed)
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
static int build_twin(vnect_handle* h)
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
int enqueue_frame(vnect_handle* h, int slot, double t2d, double t3d, int* ring_out, int stream)
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

int collect_impl(vnect_handle* h, double* j2, float* j3, int32_t* stream_out)
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
            // workgroup 0's span in both clocks ([24], [25] shader cycles; [0], [26] 100 MHz): the clock held during this launch
            if (p[25] > p[24] && p[26] > p[0] && p[26] - p[0] < 100000000ull) h->tim.shader_cycles += (double)(p[25] - p[24]), h->tim.shader_ticks += (double)(p[26] - p[0]);
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
    const uint8_t *src = nullptr, *src_dev = nullptr, *src_end = nullptr;  // src_end: end of the pinned buffer, as the device addresses it
    size_t src_stride = (size_t)row_stride;
    for (int i = 0; i < 2 && !src; i++)  // already in pinned memory (a crop of a frame the caller captured into vnect_frame_buffer)?
        if (h->stage[i] && bgr >= h->stage[i] && bgr + span <= h->stage[i] + h->stage_cap[i])
            src = bgr, src_dev = h->stage_dev[i] + (bgr - h->stage[i]), src_end = h->stage_dev[i] + h->stage_cap[i];
    if (!src) {
        const int i = 2;
        int rc = ensure_stage(h, i, (size_t)H * row);
        if (rc) return rc;
        if ((size_t)row_stride == row) memcpy(h->stage[i], bgr, (size_t)H * row);
        else
            for (int y = 0; y < H; y++) memcpy(h->stage[i] + (size_t)y * row, bgr + (size_t)y * (size_t)row_stride, row);
        src = h->stage[i], src_dev = h->stage_dev[i], src_stride = row, src_end = h->stage_dev[i] + h->stage_cap[i];
    }
    uint8_t* dst = h->frames + (size_t)slot * h->cfg.max_frame_bytes;
    static const bool dma = getenv("VNECT_INFER_DMA") != nullptr;
    if (!dma) HIPCK(h, launch_frame_copy(src_dev, dst, H, (int)row, (long long)src_stride, src_end, h->st));
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
// handle has: empty slot, new filters, no timestamps.  VNECT_PRIME_FRAMES overrides the count (0 = off).
// Round 6: the count is a MINIMUM -- the grey frames go on until VNECT_PRIME_MS (default 40) milliseconds have passed, at most 400 frames:
// 24 bf16 frames are 8 ms, and the driver-length bench (5 warm-up + 20 timed frames directly behind vnect_finalize) showed every frame
// of such a window 0.5-2 % slower than the one before it (profiles/r06_short_run_tail.txt): the clock ramp takes tens of milliseconds.
int prime(vnect_handle* h)
{
    const int n_env = getenv("VNECT_PRIME_FRAMES") ? atoi(getenv("VNECT_PRIME_FRAMES")) : 24;  // (read per call: a test flips it inside one process)
    const double min_ms = getenv("VNECT_PRIME_MS") ? atof(getenv("VNECT_PRIME_MS")) : 40.0;
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
    // like a launch error, =state like a benign refusal.  Compiled ONLY into the test build (`make testhooks`, -DVNECT_TEST_HOOKS=1 ->
    // libvnect_hip_testhooks.so; advisor, round 5): the shipped library has no hook on its initialisation path and ignores the variable.
#if defined(VNECT_TEST_HOOKS) && VNECT_TEST_HOOKS
    const char* inject = getenv("VNECT_PRIME_INJECT");
#else
    const char* inject = nullptr;
#endif
    timespec ts0;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    auto elapsed_ms = [&] {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)(ts.tv_sec - ts0.tv_sec) * 1e3 + (double)(ts.tv_nsec - ts0.tv_nsec) * 1e-6;
    };
    for (int i = 0; (i < n_env || (i < 400 && elapsed_ms() < min_ms)) && !rc; i++, t += 1.0) {
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


}  // namespace rt
}  // namespace vnect
