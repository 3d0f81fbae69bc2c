// rt_comm.cpp -- the one exchange of the pyramid-sharded path (the S images of a frame meet only in the merge,
// /root/reference/src/estimator.py:100-129): ncclAllGather over a dlopen'ed RCCL, or peer writes into IPC-mapped blocks.
#include "runtime.h"

namespace vnect {
namespace rt {

// ---- RCCL, opened lazily so single-GPU use never loads it -----------------------------------------------
typedef struct { char internal[128]; } nccl_uid;
static void* g_rccl = nullptr;
static int (*p_ncclGetUniqueId)(nccl_uid*) = nullptr;
static int (*p_ncclCommInitRank)(void**, int, nccl_uid, int) = nullptr;
static int (*p_ncclAllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
static int (*p_ncclCommDestroy)(void*) = nullptr;
static const char* (*p_ncclGetErrorString)(int) = nullptr;

// Which copy?  A process that also runs torch.distributed's "nccl" backend (bench.py --pyramid, parallel.PyramidJob) has torch's
// bundled torch/lib/librccl.so mapped already; /opt/rocm/lib/librccl.so.1 is ANOTHER build with the same SONAME.  Two copies in one
// process would each keep their own bootstrap threads, proxy state and IPC caches on the same device, and with RTLD_GLOBAL the
// second one's internal symbols could bind into the first.  So: (1) VNECT_RCCL_LIB names a file explicitly; (2) a copy this
// process has mapped already (dl_iterate_phdr: any object whose file name starts with "librccl.so") is REUSED (RTLD_NOLOAD: a
// reference to that very mapping); (3) only then is librccl.so.1 / librccl.so opened through the ordinary search (this library's
// RUNPATH is the ROCm it was built with).  Always RTLD_LOCAL: only the five entry points below are looked up, by dlsym.
static std::mutex g_rccl_mu;
static bool g_rccl_reused = false;
static std::string g_rccl_path;
static int rccl_find_mapped(struct dl_phdr_info* info, size_t, void* out)
{
    const char* n = info->dlpi_name;
    if (!n || !*n) return 0;
    const char* b = strrchr(n, '/');
    b = b ? b + 1 : n;
    if (strncmp(b, "librccl.so", 10) != 0) return 0;
    *(std::string*)out = n;
    return 1;
}
static bool load_rccl()
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

void comm_destroy(vnect_handle* h)
{
    if (h->comm && p_ncclCommDestroy) p_ncclCommDestroy(h->comm);
    h->comm = nullptr;
}

}  // namespace rt
}  // namespace vnect

using namespace vnect;
using namespace vnect::rt;

extern "C" {

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
