// What does a do-nothing launch cost inside a dependent chain (HIP graph of 64 kernel nodes), as a function of
// grid size, dynamic LDS request and kernel-argument size?  (The GEMM launches of the step never take less than
// ~4.5 us; an empty <<<1,64>>> node takes ~1.7 us.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

struct Big { float* p; int pad[500]; };  // ~2 KB of kernel arguments

__global__ __launch_bounds__(256) void k_small(float* p, int v) {
    extern __shared__ float sm[];
    if (v == 12345) p[threadIdx.x] = sm[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_big(Big b, int v) {
    extern __shared__ float sm[];
    if (b.pad[v & 255] == 12345) b.p[threadIdx.x] = sm[threadIdx.x];  // reads one argument word
}
__global__ __launch_bounds__(256) void k_touch(float* p, int v) {  // every wave does one dependent global round trip
    extern __shared__ float sm[];
    const float x = p[(blockIdx.x * 256 + threadIdx.x) & 65535];
    if (x == 12345.f || v == 12345) p[threadIdx.x] = x + sm[0];
}

template <typename F>
static float run(F launch, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 64; ++i) launch();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1000.f / 640.f;
}

int main() {
    hipStream_t s; hipStreamCreate(&s);
    float* p; hipMalloc(&p, 1 << 20); hipMemset(p, 0, 1 << 20);
    hipFuncSetAttribute((const void*)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    Big b{}; b.p = p;
    for (int grid : {1, 256, 1024}) {
        for (size_t lds : {(size_t)0, (size_t)70 * 1024, (size_t)82 * 1024}) {
            const float a = run([&] { hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), lds, s, p, 0); }, s);
            const float c = run([&] { hipLaunchKernelGGL(k_big, dim3(grid), dim3(256), lds, s, b, 0); }, s);
            const float t = run([&] { hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), lds, s, p, 0); }, s);
            printf("grid %4d lds %6zu: empty %.2f us, 2KB-args %.2f us, one global round trip %.2f us\n", grid, lds, a, c, t);
        }
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
