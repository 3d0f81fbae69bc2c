// rt_abi.cpp -- the extern "C" entry points of include/vnect_abi.h (which cites the reference lines each one replaces): argument and
// state checks, device selection, and the guard that lets no C++ exception cross the boundary.  The work is in rt_plan / rt_exec / rt_comm.
#include "runtime.h"

namespace vnect {
namespace rt {
thread_local std::string g_create_error = "";
}  // namespace rt
}  // namespace vnect

// =====================================================================================================
extern "C" {

int vnect_abi_version(void) { return VNECT_ABI_VERSION; }

#ifndef VNECT_BUILD_FLAGS
#define VNECT_BUILD_FLAGS "?"
#endif
#ifndef VNECT_BUILD_VARIANT
#define VNECT_BUILD_VARIANT ""
#endif
#ifndef VNECT_TEST_HOOKS
#define VNECT_TEST_HOOKS 0
#endif
const char* vnect_build_info(void)
{
    static const std::string info = [] {
        std::string s = "abi=" + std::to_string(VNECT_ABI_VERSION) + "; compiler=" + __VERSION__ + "; flags=" VNECT_BUILD_FLAGS + "; variant=" VNECT_BUILD_VARIANT +
                        "; test_hooks=" + std::to_string((int)VNECT_TEST_HOOKS) + "; conv: " + conv_build_probes() + "; post: " + post_build_probes() +
                        "; probes_off=" + ((conv_probes_off() && post_probes_off()) ? "1" : "0");
        return s;
    }();
    return info.c_str();
}

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
    comm_destroy(h);
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
            for (size_t i = 0; i < raw.size(); i++) out[i] = plan::from_bf16(raw[i]);
        }
        return VNECT_OK;
    });
}

int vnect_get_layer_stamps(vnect_handle* h, int idx, uint64_t* out24)
{
    return guarded(&h, [&]() -> int {
        if (!h || !out24 || idx < 0 || idx >= (int)h->layers.size()) return h ? fail(h, VNECT_E_ARG, "bad layer index") : VNECT_E_ARG;
        for (int k = 0; k < 24; k++) out24[k] = h->h_prof[PROF_SLOTS * idx + k];
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
        // ABI v5 callers pass the struct without the shader-clock fields: they get exactly what they got before
        constexpr int32_t V5_SIZE = (int32_t)offsetof(vnect_timings, shader_cycles);
        if (!h || !out || (out->struct_size != (int32_t)sizeof(vnect_timings) && out->struct_size != V5_SIZE)) return VNECT_E_ARG;
        const int32_t sz = out->struct_size;
        vnect_timings t = h->tim;
        t.struct_size = sz;
        t.conv_launches = h->conv_launches;
        t.conv_flops = h->conv_flops;
        memcpy(out, &t, (size_t)sz);
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

}  // extern "C"
