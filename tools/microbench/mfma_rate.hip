// Calibration: what fp32 MFMA rate does this MI355X sustain (a) register-only, (b) with LDS fragment
// reads in the loop as the GEMM kernel does, for 1 and 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* in) {
    __shared__ __attribute__((aligned(16))) float sm[64 * 72 * 2];
    for (int i = threadIdx.x; i < 64 * 72 * 2; i += 256) sm[i] = in[i & 1023];
    __syncthreads();
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0, 0, 0, 0};
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    float a0 = in[lane], b0 = in[lane + 64];
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 va = *reinterpret_cast<const f32x4*>(sm + r * 72 + 16 * c + 4 * q);
                f32x4 vb = *reinterpret_cast<const f32x4*>(sm + 64 * 72 + r * 72 + 16 * c + 4 * q);
                f32x4 va2 = *reinterpret_cast<const f32x4*>(sm + (16 + r) * 72 + 16 * c + 4 * q);
                f32x4 vb2 = *reinterpret_cast<const f32x4*>(sm + 64 * 72 + (16 + r) * 72 + 16 * c + 4 * q);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[0 % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[t], vb[t], acc[0 % NACC], 0, 0, 0);
                    acc[1 % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[t], vb2[t], acc[1 % NACC], 0, 0, 0);
                    acc[2 % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(va2[t], vb[t], acc[2 % NACC], 0, 0, 0);
                    acc[3 % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(va2[t], vb2[t], acc[3 % NACC], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc[0 % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0 % NACC], 0, 0, 0);
                acc[1 % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[1 % NACC], 0, 0, 0);
                acc[2 % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[2 % NACC], 0, 0, 0);
                acc[3 % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[3 % NACC], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char* name, int grid, int iters, float* out, float* in) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NACC, LDS>), dim3(grid), dim3(256), 0, 0, out, iters, in);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<NACC, LDS>), dim3(grid), dim3(256), 0, 0, out, iters, in);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma = (double)grid * 4 * iters * 64.0;  // waves * iters * 64 mfma
    double tf = mfma * 2048.0 / (ms / reps * 1e-3) / 1e12;
    double cyc = (ms / reps * 1e-3) * 2.4e9 / (iters * 64.0) / ((grid + 255) / 256);
    printf("%-34s grid %4d iters %5d : %8.2f us/launch  %7.1f TFLOP/s  (%.1f cyc@2.4GHz per MFMA per wave-slot)\n", name, grid, iters,
           ms / reps * 1e3, tf, cyc);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&in, 4096 * 4);
    hipMemset(in, 0, 4096 * 4);
    for (int grid : {256, 512}) {
        for (int iters : {8, 64, 512}) {
            run<4, false>("reg-only 4acc", grid, iters, out, in);
            run<1, false>("reg-only 1acc", grid, iters, out, in);
            run<4, true>("lds b128 frags 4acc", grid, iters, out, in);
            run<2, true>("lds b128 frags 2acc", grid, iters, out, in);
        }
    }
    return 0;
}
