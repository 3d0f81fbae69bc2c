// Microbenchmark: does kernarg PRELOAD (the first dwords of the argument segment delivered in SGPRs at wave launch, hipcc -mllvm
// -amdgpu-kernarg-preload-count=N) shorten a short dependent kernel, or does the wave launch simply wait for the same fetch?
// (Question behind it, DESIGN.md section 8 item 4: ~0.4 us of every conv launch's cold start is the scalar loads of its argument block.)
// A chain of dependent launches of a 512-workgroup kernel whose every wave needs all 14 argument dwords before it can do anything; the
// same source is built twice (tools/kernarg_preload and tools/kernarg_preload_on) and both binaries run in ONE gpurun call.
// build: hipcc --offload-arch=gfx950 -O3 tools/kernarg_preload.hip -o tools/kernarg_preload
//        hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=14 tools/kernarg_preload.hip -o tools/kernarg_preload_on
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(512) void k(float* out, int a0, int a1, int a2, int a3, int a4, int a5, int a6, int a7, int a8, int a9, int a10, int a11)
{
    // every wave uses every argument at once: the sum decides the one store
    const int s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11;
    if (threadIdx.x == (unsigned)(s & 511)) out[blockIdx.x] += 1.f;
}

int main()
{
    float* out;
    hipMalloc(&out, 512 * 4);
    hipMemset(out, 0, 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int N = 4000;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, out, i, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%d dependent launches of a 512 x 512 kernel: %.3f us per launch\n", N, ms * 1e3 / N);
    }
    return 0;
}
