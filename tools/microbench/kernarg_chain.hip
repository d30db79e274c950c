// What does a chain of DEPENDENT scalar loads from the kernel-argument segment cost a launch?  (The grouped GEMM kernel
// reads its problem descriptor through ~8 serialized s_load round trips before its first vector load.)  256 workgroups,
// every wave walks NCH dependent words in NCH different 64-byte lines of a 2 KB by-value argument, then (optionally) one
// dependent global load; 64 launches in a HIP graph, each launch with its OWN argument values.
// Build twice: plain, and with -mllvm -amdgpu-kernarg-preload-count=8 -DPRELOAD (the first 8 scalar args arrive in SGPRs).
#include <hip/hip_runtime.h>
#include <stdio.h>

struct Lines { int w[32][16]; };  // 32 lines of 64 B

template <int NCH, bool GLOBAL>
__global__ __launch_bounds__(256) void k_chain(int a0, int a1, int a2, int a3, int a4, int a5, int a6, int a7,
                                               const float* p, float* out, const Lines L) {
    int v = a0;  // 0 at run time
#pragma unroll
    for (int i = 0; i < NCH; ++i) v = L.w[2 * i + 1 + (v & 1)][v & 15];  // next line depends on the word just read
    float x = 0.f;
    if (GLOBAL) x = p[((blockIdx.x * 256 + threadIdx.x) * 4 + v) & 0xFFFFF];
    if (v == 12345 || x == 12345.f) out[threadIdx.x] = x + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int NCH, bool GLOBAL>
static float run(hipStream_t s, const float* p, float* out) {
    Lines L{};  // all zeros: every step reads word 0 of the next odd line
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 64; ++i) hipLaunchKernelGGL((k_chain<NCH, GLOBAL>), dim3(256), dim3(256), 0, s, 0, i, 2, 3, 4, 5, 6, 7, p, out, L);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1000.f / (64.f * 20.f);
}

int main() {
    hipStream_t s; hipStreamCreate(&s);
    float *p, *out; hipMalloc(&p, 4 << 20); hipMemset(p, 0, 4 << 20); hipMalloc(&out, 4096);
#ifdef PRELOAD
    printf("kernarg preload build (first 8 dwords in SGPRs at wave start, if the firmware honours it)\n");
#else
    printf("plain build\n");
#endif
    printf("dependent kernarg lines : 0 %.2f | 1 %.2f | 2 %.2f | 4 %.2f | 8 %.2f us per launch\n", run<0, false>(s, p, out),
           run<1, false>(s, p, out), run<2, false>(s, p, out), run<4, false>(s, p, out), run<8, false>(s, p, out));
    printf("... + one global load   : 0 %.2f | 1 %.2f | 2 %.2f | 4 %.2f | 8 %.2f us per launch\n", run<0, true>(s, p, out),
           run<1, true>(s, p, out), run<2, true>(s, p, out), run<4, true>(s, p, out), run<8, true>(s, p, out));
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
