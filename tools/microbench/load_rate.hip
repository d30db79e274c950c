// Calibration: operand-fetch rate of the register-direct GEMM's load patterns with no MFMA at all.
// A 64x64 output tile per workgroup, 4 waves cutting k (as config 6); per 16-k chunk each wave issues
// 4 A loads + 4 B loads of 16 B per lane.  "KC" = k-contiguous operand ([row][k]: a wave load touches 16 rows x 64 B
// per 16-lane group), "KM" = k-major ([k][row]: 4 k-rows x 256 B).  Prints bytes/clk/CU for NN (A KC, B KM), TN
// (both KM) and NT (both KC) at 1 and 2 workgroups per CU.  The fp32 MFMAs of such a tile need 16 B/clk/CU at peak.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool AKM, bool BKM>
__global__ __launch_bounds__(256) void k(const float* __restrict__ A, const float* __restrict__ B, float* out, int M, int N,
                                         int K, int reps, int remap) {
    const int tiles_n = N / 64;
    int tl = blockIdx.x;
    if (remap) {  // XCD x (= blockIdx % 8) gets the contiguous tile range [x*T/8, (x+1)*T/8)
        const int T = gridDim.x;
        tl = (tl & 7) * (T >> 3) + (tl >> 3);
    }
    const int tm = tl / tiles_n, tn = tl % tiles_n;
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const int nc = K / 16 / 4;  // chunks per wave
    f32x4 s[8];
    for (int i = 0; i < 8; ++i) s[i] = f32x4{0, 0, 0, 0};
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll 2
        for (int c = ks * nc; c < (ks + 1) * nc; ++c) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float* pa = AKM ? A + (size_t)(16 * c + 4 * q + b) * M + tm * 64 + 4 * r
                                      : A + (size_t)(tm * 64 + 16 * b + r) * K + 16 * c + 4 * q;
                const float* pb = BKM ? B + (size_t)(16 * c + 4 * q + b) * N + tn * 64 + 4 * r
                                      : B + (size_t)(tn * 64 + 16 * b + r) * K + 16 * c + 4 * q;
                s[b] += *reinterpret_cast<const f32x4*>(pa);
                s[4 + b] += *reinterpret_cast<const f32x4*>(pb);
            }
        }
    }
    f32x4 t = s[0] + s[1] + s[2] + s[3] + s[4] + s[5] + s[6] + s[7];
    out[blockIdx.x * 256 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
}

template <bool AKM, bool BKM>
void run(const char* name, int M, int N, int K, int reps, const float* A, const float* B, float* out, int remap) {
    const int grid = (M / 64) * (N / 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<AKM, BKM>), dim3(grid), dim3(256), 0, 0, A, B, out, M, N, K, reps, remap);
    hipEventRecord(e0, 0);
    const int n = 10;
    for (int w = 0; w < n; ++w) hipLaunchKernelGGL((k<AKM, BKM>), dim3(grid), dim3(256), 0, 0, A, B, out, M, N, K, reps, remap);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * 4 * (K / 64) * 8 * 1024.0 * reps;
    const double sec = ms / n * 1e-3;
    printf("remap=%d %s M=%d N=%d K=%d grid=%d: %.1f us/launch, %.2f TB/s, %.1f B/clk/CU\n", remap, name, M, N, K, grid, sec * 1e6,
           bytes / sec / 1e12, bytes / (sec * 2.4e9) / 256.0);
}

int main() {
    float *A, *B, *out;
    hipMalloc(&A, 4096 * 1024 * 4); hipMalloc(&B, 1024 * 1024 * 4); hipMalloc(&out, 1024 * 256 * 4);
    hipMemset(A, 0, 4096 * 1024 * 4); hipMemset(B, 0, 1024 * 1024 * 4);
    for (int M : {1024, 2048, 4096}) {
        for (int remap = 0; remap < 2; ++remap) {
            run<false, true>("NN (A KC, B KM)", M, 512, 512, 20, A, B, out, remap);
            run<true, true>("TN (both KM)   ", M, 512, 512, 20, A, B, out, remap);
            run<false, false>("NT (both KC)   ", M, 512, 512, 20, A, B, out, remap);
        }
    }
    hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
