// Probe (round 6): what bounds vnect_infer's host-to-device copy?  406 272 bytes (one 368 x 368 BGR frame) from
// device-mapped PINNED host memory to device memory, as a kernel on a stream (post.hip: frame_copy_kernel) in several shapes, beside
// hipMemcpyAsync (the copy engine) and an empty launch (the floor).  Per-launch time = N back-to-back launches between two events.
// If no shape moves the figure, the bound is the link's read path (outstanding reads x payload / latency), not the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o tools/pcie_copy_probe tools/pcie_copy_probe.hip && tools/pcie_copy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            exit(1);                                                             \
        }                                                                        \
    } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void empty_kernel() {}
// U units of 16 bytes per thread, grid-strided (consecutive threads touch consecutive units in every pass)
template <int U, bool NT>
__global__ void copy16(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int n16)
{
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    u32x4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        const int i = i0 + k * stride;
        if (i < n16) v[k] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
        const int i = i0 + k * stride;
        if (i < n16) dst[i] = v[k];
    }
}
__global__ void copy4(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int n4)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    const size_t bytes = 368 * 368 * 3;
    const int n16 = (int)(bytes / 16), n4 = (int)(bytes / 4);
    unsigned char *h = nullptr, *hd = nullptr, *d = nullptr;
    CK(hipHostMalloc((void**)&h, 1 << 20, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&hd, h, 0));
    CK(hipMalloc((void**)&d, 1 << 20));
    for (size_t i = 0; i < bytes; i++) h[i] = (unsigned char)(i * 7 + 3);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<unsigned char> back(bytes);
    auto timeit = [&](const char* name, auto launch, bool check) {
        CK(hipMemsetAsync(d, 0, bytes, st));
        for (int i = 0; i < 20; i++) launch();
        CK(hipStreamSynchronize(st));
        bool ok = true;
        if (check) {
            CK(hipMemcpy(back.data(), d, bytes, hipMemcpyDeviceToHost));
            ok = memcmp(back.data(), h, bytes) == 0;
        }
        std::vector<double> us;
        for (int rep = 0; rep < 7; rep++) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; i++) launch();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            us.push_back(ms * 1e3 / reps);
        }
        std::sort(us.begin(), us.end());
        printf("  %-62s %7.2f us per launch (min %.2f max %.2f)%s  %5.1f GB/s\n", name, us[3], us[0], us[6], check ? (ok ? "  bytes ok" : "  BYTES WRONG") : "          ",
               check ? bytes / (us[3] * 1e-6) / 1e9 : 0.0);
        return us[3];
    };
    printf("406 272 bytes, pinned (mapped) host memory -> device memory, %d back-to-back launches per sample:\n", reps);
    timeit("empty kernel (launch floor)", [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st); }, false);
    timeit("hipMemcpyAsync (copy engine)", [&] { CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st)); }, true);
    timeit("16 B per thread, 256-thread blocks (100 blocks) = the product's form", [&] { hipLaunchKernelGGL((copy16<1, false>), dim3((n16 + 255) / 256), dim3(256), 0, st, (const u32x4*)hd, (u32x4*)d, n16); }, true);
    timeit("16 B per thread, 64-thread blocks (397 blocks)", [&] { hipLaunchKernelGGL((copy16<1, false>), dim3((n16 + 63) / 64), dim3(64), 0, st, (const u32x4*)hd, (u32x4*)d, n16); }, true);
    timeit("16 B per thread, 1024-thread blocks (25 blocks)", [&] { hipLaunchKernelGGL((copy16<1, false>), dim3((n16 + 1023) / 1024), dim3(1024), 0, st, (const u32x4*)hd, (u32x4*)d, n16); }, true);
    timeit("2 x 16 B per thread in flight, 256-thread blocks (50 blocks)", [&] { hipLaunchKernelGGL((copy16<2, false>), dim3((n16 / 2 + 255) / 256), dim3(256), 0, st, (const u32x4*)hd, (u32x4*)d, n16); }, true);
    timeit("4 x 16 B per thread in flight, 256-thread blocks (25 blocks)", [&] { hipLaunchKernelGGL((copy16<4, false>), dim3((n16 / 4 + 255) / 256), dim3(256), 0, st, (const u32x4*)hd, (u32x4*)d, n16); }, true);
    timeit("8 x 16 B per thread in flight, 256-thread blocks (13 blocks)", [&] { hipLaunchKernelGGL((copy16<8, false>), dim3((n16 / 8 + 255) / 256), dim3(256), 0, st, (const u32x4*)hd, (u32x4*)d, n16); }, true);
    timeit("16 B per thread, non-temporal loads, 256-thread blocks", [&] { hipLaunchKernelGGL((copy16<1, true>), dim3((n16 + 255) / 256), dim3(256), 0, st, (const u32x4*)hd, (u32x4*)d, n16); }, true);
    timeit("4 B per thread, 256-thread blocks (397 blocks)", [&] { hipLaunchKernelGGL(copy4, dim3((n4 + 255) / 256), dim3(256), 0, st, (const unsigned*)hd, (unsigned*)d, n4); }, true);
    // half the frame by the copy engine on a second stream while a kernel copies the other half (joined by an event)
    hipStream_t st2;
    CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    hipEvent_t ev, ev0;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev0, hipEventDisableTiming));
    timeit("halves: copy engine (2nd stream) + kernel, joined by an event", [&] {
        const size_t half = (bytes / 2) & ~(size_t)15;
        CK(hipEventRecord(ev0, st));
        CK(hipStreamWaitEvent(st2, ev0, 0));
        CK(hipMemcpyAsync(d + half, h + half, bytes - half, hipMemcpyHostToDevice, st2));
        CK(hipEventRecord(ev, st2));
        hipLaunchKernelGGL((copy16<1, false>), dim3((int)((half / 16 + 255) / 256)), dim3(256), 0, st, (const u32x4*)hd, (u32x4*)d, (int)(half / 16));
        CK(hipStreamWaitEvent(st, ev, 0));
    }, true);
    return 0;
}
