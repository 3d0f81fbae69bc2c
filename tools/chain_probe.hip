// chain_probe.hip -- can consecutive layers overlap their launch / fill / drain when they are chained by per-row-block counters
// instead of kernel boundaries?  (the fixed cost per launch -- boundary 1.7 us, setup 0.75 us, first chunk
// 1.2 us, drain -- is ~19 % of a frame.)  Standalone: no product code.
//
// A "layer" is a grid of NB workgroups (512 threads, 72 KB of LDS: two per CU, like conv_stream_kernel<64,64,1,5>); workgroup b
//   1. waits until counter_in[b] says its 64 input rows (128 floats each) are complete  (mode 1; mode 0: stream order only)
//   2. fetches them by buffer-addressed LDS-DMA (optionally with sc1), 3. spins `work` batches of 16 MFMAs,
//   4. writes rows + 1.0 with write-through stores, drains, 5. adds 1 to counter_out[b] (agent scope).
// Chain: layer l reads buffer l % 3, writes (l + 1) % 3.  mode 0: all layers on ONE stream.  mode 1: layers alternate between TWO
// streams with no event between them, so layer l + 1 is dispatched while layer l runs and its workgroups wait on the counters
// (at most two layers are in flight and NB <= 256 each, so everything in flight is resident: nobody waits for an undispatched
// workgroup).  Every frame starts from fresh values and the result is checked exactly: a stale read shows as a wrong sum.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/chain_probe tools/chain_probe.hip ; run on the GPU box: tools/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ROWS = 64, C = 128;  // one workgroup's block: 64 rows x 512 B = 32 KB

__global__ __launch_bounds__(512, 4) void layer_kernel(const float* in, float* out, int* cnt_in, int* cnt_out, int need, int work,
                                                        int sc1_loads, int* err, unsigned long long* stamps)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (stamps && tid == 0 && b == 0) stamps[0] = __builtin_amdgcn_s_memrealtime();
    __shared__ int ok;
    if (cnt_in) {
        if (tid == 0) {
            int seen = 0;
            for (int spin = 0; spin < (1 << 18); spin++) {  // bounded
                seen = __hip_atomic_load(cnt_in + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seen >= need) break;
                __builtin_amdgcn_s_sleep(2);
            }
            ok = seen >= need;
            if (!ok) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    // LDS-DMA: 512 lanes x 16 B = 8 KB per instruction, 4 instructions for the block
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (size_t)b * ROWS * C), 0, ROWS * C * 4, 0x00020000);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float* dst = lds + i * 2048 + wave * 256;  // the hardware adds lane * 16 B
        if (sc1_loads) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)dst, 16, (i * 2048 + tid * 4) * 4, 0, 0, 16);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)dst, 16, (i * 2048 + tid * 4) * 4, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // stand-in for the K loop: consumer waves 0-3 run `work` x 16 dependent MFMAs (1024 cycles per batch at full rate)
    f32x16 acc = {};
    if (wave < 4) {
        const float x = lds[tid], y = lds[tid + 64];
        for (int w = 0; w < work; w++) {
#pragma unroll
            for (int k = 0; k < 16; k++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
        }
    }
    float sink = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) sink += acc[r];
    const float zero = sink == 12345.678f ? 1.f : 0.f;  // keeps the MFMAs alive, contributes nothing
    float* o = out + (size_t)b * ROWS * C;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int idx = i * 512 + tid;
        __hip_atomic_store(o + idx, lds[idx] + 1.0f + zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: write-through
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(cnt_out + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stamps) stamps[1 + b] = __builtin_amdgcn_s_memrealtime();
    }
}

int main(int argc, char** argv)
{
    const int NB = argc > 1 ? atoi(argv[1]) : 200;     // workgroups per layer (<= 256)
    const int L = argc > 2 ? atoi(argv[2]) : 16;       // layers per frame
    const int frames = argc > 3 ? atoi(argv[3]) : 200;
    if (NB > 256 || NB < 1 || L < 2 || L > 64) return printf("bad arguments\n"), 1;
    const size_t n = (size_t)NB * ROWS * C;
    float* buf[3];
    for (auto& p : buf) CK(hipMalloc(&p, n * 4));
    int *cnt, *err;
    CK(hipMalloc(&cnt, (size_t)(L + 1) * 256 * 4));
    CK(hipMalloc(&err, 4));
    CK(hipMemset(err, 0, 4));
    CK(hipMemset(cnt, 0, (size_t)(L + 1) * 256 * 4));
    hipStream_t st[2];
    CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void*)layer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    std::vector<float> host(n), init(n);
    hipEvent_t e0, e1, ej, ej2;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ej2, hipEventDisableTiming));
    printf("NB=%d workgroups per layer, %d layers per frame, %d frames\n", NB, L, frames);
    printf("%-34s %6s %10s %10s %8s %8s\n", "mode", "work", "us/frame", "us/layer", "wrong", "timeouts");
    for (int work : {8, 16, 32}) {
        hipGraphExec_t graphs[3] = {nullptr, nullptr, nullptr};
        for (int mode = 0; mode < 3; mode++) {  // 0: one stream; 1: two streams + counters, plain LDS-DMA; 2: the same with sc1 LDS-DMA
            long long wrong = 0;
            float ms_total = 0;
            for (int f = -5; f < frames; f++) {
                for (size_t i = 0; i < n; i++) init[i] = (float)((f + 7) * 3 + (int)(i % 5));
                CK(hipMemcpy(buf[0], init.data(), n * 4, hipMemcpyHostToDevice));
                CK(hipMemset(cnt, 0, (size_t)(L + 1) * 256 * 4));
                CK(hipDeviceSynchronize());
                if (!graphs[mode]) {  // one graph per mode: the host's launch cost stays out of the measurement
                    hipGraph_t g;
                    CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal));
                    if (mode != 0) {
                        CK(hipEventRecord(ej, st[0]));
                        CK(hipStreamWaitEvent(st[1], ej, 0));  // fork
                    }
                    for (int l = 0; l < L; l++) {
                        hipStream_t s = mode == 0 ? st[0] : st[l & 1];
                        int* ci = mode == 0 || l == 0 ? nullptr : cnt + l * 256;
                        hipLaunchKernelGGL(layer_kernel, dim3(NB), dim3(512), 72 * 1024, s, (const float*)buf[l % 3], buf[(l + 1) % 3], ci,
                                           cnt + (l + 1) * 256, 1, work, mode == 2 ? 1 : 0, err, (unsigned long long*)nullptr);
                    }
                    if (mode != 0) {
                        CK(hipEventRecord(ej2, st[1]));
                        CK(hipStreamWaitEvent(st[0], ej2, 0));  // join
                    }
                    CK(hipStreamEndCapture(st[0], &g));
                    CK(hipGraphInstantiate(&graphs[mode], g, nullptr, nullptr, 0));
                }
                CK(hipEventRecord(e0, st[0]));
                CK(hipGraphLaunch(graphs[mode], st[0]));
                CK(hipEventRecord(e1, st[0]));
                CK(hipEventSynchronize(e1));
                CK(hipDeviceSynchronize());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (f >= 0) ms_total += ms;
                CK(hipMemcpy(host.data(), buf[L % 3], n * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < n; i++) wrong += host[i] != init[i] + (float)L;
            }
            int herr = 0;
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            CK(hipMemset(err, 0, 4));
            const char* names[3] = {"one stream (kernel boundaries)", "two streams + counters", "two streams + counters, sc1 DMA"};
            printf("%-34s %6d %10.2f %10.2f %8lld %8d\n", names[mode], work, ms_total / frames * 1e3, ms_total / frames * 1e3 / L, wrong, herr);
            fflush(stdout);
        }
    }
    return 0;
}
