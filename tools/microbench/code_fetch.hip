// Does cold instruction fetch set the launch floor?  A kernel that executes N straight-line FMAs once (code size
// ~8 N bytes, executed front to back) against the same work as a loop (tiny code), 64-node dependent graph.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int N, bool LOOP>
__global__ __launch_bounds__(256) void k(float* p, float a, float b) {
    float x = a + threadIdx.x;
    if (LOOP) {
#pragma unroll 1
        for (int i = 0; i < N; ++i) x = __builtin_fmaf(x, a, b + (float)(i & 3));
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) x = __builtin_fmaf(x, a, b + (float)(i & 3));
    }
    if (x == 12345.678f) p[threadIdx.x] = x;
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
    return ms * 1000.f / 640.f;
}

template <int N>
void both(float* p, hipStream_t s) {
    const float l = run([&] { hipLaunchKernelGGL((k<N, true>), dim3(256), dim3(256), 0, s, p, 1.0001f, 0.5f); }, s);
    const float u = run([&] { hipLaunchKernelGGL((k<N, false>), dim3(256), dim3(256), 0, s, p, 1.0001f, 0.5f); }, s);
    printf("N=%5d (~%3d KB unrolled): loop %.2f us, straight-line %.2f us\n", N, N * 8 / 1024, l, u);
}

int main() {
    hipStream_t s; hipStreamCreate(&s);
    float* p; hipMalloc(&p, 1 << 20);
    both<256>(p, s); both<1024>(p, s); both<4096>(p, s); both<8192>(p, s);
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
