// Prototype: LDS-free "register-direct" fp32 MFMA GEMM.  Each wave loads its MFMA fragments straight
// from global/L2 into VGPRs (16 B per lane), D chunks of 16 k in flight, no barriers in the k loop.
// Measures NT (both operands k-contiguous), NN, TN on the VAE's shapes to decide the production design.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int T> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<2> { typedef f32x2 type; };
template <> struct VecT<4> { typedef f32x4 type; };
template <int T> __device__ __forceinline__ float vget(const typename VecT<T>::type& v, int i);
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }
template <> __device__ __forceinline__ float vget<2>(const f32x2& v, int i) { return v[i]; }
template <> __device__ __forceinline__ float vget<4>(const f32x4& v, int i) { return v[i]; }

// one operand's fragments for one 16-k chunk: f[t][b]
template <int T, bool KM>
__device__ __forceinline__ void load_frag(float (&f)[4][T], const float* __restrict__ base, int ld, int off0, int c,
                                          int r, int q) {
    if (!KM) {  // memory [row][k]: lane (r,q) takes k = 16c+4q..+3 of row off0+16b+r
#pragma unroll
        for (int b = 0; b < T; ++b) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)(off0 + 16 * b + r) * ld + 16 * c + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) f[t][b] = v[t];
        }
    } else {  // memory [k][row]: lane (r,q) takes rows off0 + T*r .. +T-1 of k = 16c+4q+t
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const typename VecT<T>::type v =
                *reinterpret_cast<const typename VecT<T>::type*>(base + (size_t)(16 * c + 4 * q + t) * ld + off0 + T * r);
#pragma unroll
            for (int b = 0; b < T; ++b) f[t][b] = vget<T>(v, b);
        }
    }
}

template <int TM, int TN, int WM, int WN, int KS, int D, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void rd_gemm(const float* __restrict__ A, const float* __restrict__ B,
                                               float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = WM * 16 * TM, BN = WN * 16 * TN;
    const int tiles_n = N / BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = wave / (WM * WN), ws = wave % (WM * WN);
    const int wm = ws / WN, wn = ws % WN;
    const int r = lane & 15, q = lane >> 4;
    const int am = tm * BM + wm * 16 * TM, bn = tn * BN + wn * 16 * TN;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
    float fa[D][4][TM], fb[D][4][TN];
    const int nch = K / 16 / KS;  // chunks per wave (assumed multiple of D here)
    const int cbeg = ks * nch;
#pragma unroll
    for (int s = 0; s < D; ++s) {
        load_frag<TM, AKM>(fa[s], A, lda, am, cbeg + s, r, q);
        load_frag<TN, BKM>(fb[s], B, ldb, bn, cbeg + s, r, q);
    }
    for (int c = 0; c < nch; c += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][t][m], fb[s][t][n], acc[m][n], 0, 0, 0);
            if (c + D < nch) {
                load_frag<TM, AKM>(fa[s], A, lda, am, cbeg + c + D + s, r, q);
                load_frag<TN, BKM>(fb[s], B, ldb, bn, cbeg + c + D + s, r, q);
            }
        }
    }
    // epilogue through LDS: [KS][BM][BN+4], then sum the k-splits and write whole rows
    constexpr int SC = BN + 4;
    float* Cs = smem + ks * BM * SC;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lr = wm * 16 * TM + (AKM ? TM * (4 * q + j) + m : 16 * m + 4 * q + j);
#pragma unroll
            for (int n = 0; n < TN; ++n) Cs[lr * SC + wn * 16 * TN + (BKM ? TN * r + n : 16 * n + r)] = acc[m][n][j];
        }
    __syncthreads();
    constexpr int CQ = BN / 4;
    for (int i = tid; i < BM * CQ; i += 256) {
        const int lr = i / CQ, c4 = i % CQ;
        f32x4 v = *reinterpret_cast<const f32x4*>(smem + lr * SC + 4 * c4);
#pragma unroll
        for (int k2 = 1; k2 < KS; ++k2) v += *reinterpret_cast<const f32x4*>(smem + k2 * BM * SC + lr * SC + 4 * c4);
        *reinterpret_cast<f32x4*>(C + (size_t)(tm * BM + lr) * ldc + tn * BN + 4 * c4) = v;
    }
}

static std::vector<float> hA, hB;
template <int TM, int TN, int WM, int WN, int KS, int D, bool AKM, bool BKM>
void run(const char* name, int M, int N, int K, float* dA, float* dB, float* dC) {
    constexpr int BM = WM * 16 * TM, BN = WN * 16 * TN;
    if (M % BM || N % BN || (K / 16 / KS) % D) { printf("%-40s skipped (shape)\n", name); return; }
    const int lda = AKM ? M : K, ldb = BKM ? N : K;
    const size_t lds = (size_t)KS * BM * (BN + 4) * 4;
    auto kern = rd_gemm<TM, TN, WM, WN, KS, D, AKM, BKM>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = (M / BM) * (N / BN);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, lda, ldb, N);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 50;
    hipEventRecord(e0, 0);
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, lda, ldb, N);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int s = 0; s < 64; ++s) {
        int i = (s * 131) % M, j = (s * 71) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)(AKM ? hA[(size_t)k * M + i] : hA[(size_t)i * K + k]) * (BKM ? hB[(size_t)k * N + j] : hB[(size_t)j * K + k]);
        double e = fabs(ref - hC[(size_t)i * N + j]);
        if (e > maxerr) maxerr = e;
    }
    const double us = ms / reps * 1e3;
    // same 50 launches captured into one hipGraph
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, dA, dB, dC, M, N, K, lda, ldb, N);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int w = 0; w < 4; ++w) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double usg = ms / (4 * reps) * 1e3;
    // cold L2: every launch reads a different copy of A and B (16 sets x 8 MB > 32 MB of L2)
    const int NSETS = 16; const size_t SETSTRIDE = 1024 * 1024;
    hipEventRecord(e0, 0);
    for (int w = 0; w < 64; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA + (w % NSETS) * SETSTRIDE, dB + (w % NSETS) * SETSTRIDE, dC, M, N, K, lda, ldb, N);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double usc = ms / 64 * 1e3;
    printf("%-22s %4dx%4dx%4d grid %4d : hot %6.2f us  graph %6.2f us  cold-L2 %6.2f us (%5.1f TF)  maxerr %.1e\n", name, M, N, K, grid, us, usg,
           usc, 2.0 * M * N * K / (usc * 1e-6) / 1e12, maxerr);
}

extern "C" int conc_main();
int main(int argc, char** argv) {
    if (argc > 1) return conc_main();
    const int MAXE = 1024 * 1024;
    const int NS = 17;
    hA.resize(MAXE); hB.resize(MAXE);
    srand(1);
    for (auto& v : hA) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hB) v = (rand() % 2001 - 1000) / 1000.f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, (size_t)NS * MAXE * 4); hipMalloc(&dB, (size_t)NS * MAXE * 4); hipMalloc(&dC, MAXE * 4);
    hipMemset(dA, 0, (size_t)NS * MAXE * 4); hipMemset(dB, 0, (size_t)NS * MAXE * 4);
    hipMemcpy(dA, hA.data(), MAXE * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), MAXE * 4, hipMemcpyHostToDevice);
    // forward-like NT: [1024,512] x [512,512]^T
    printf("--- NT (KC,KC) 1024x512x512: 0.537 GFLOP, 3.4 us at 157 TF\n");
    run<2, 2, 2, 2, 1, 4, false, false>("NT t22 w2x2 ks1 D4", 1024, 512, 512, dA, dB, dC);
    run<2, 2, 1, 2, 2, 4, false, false>("NT t22 w1x2 ks2 D4", 1024, 512, 512, dA, dB, dC);
    run<2, 2, 1, 1, 4, 4, false, false>("NT t22 w1x1 ks4 D4", 1024, 512, 512, dA, dB, dC);
    run<2, 2, 1, 1, 4, 8, false, false>("NT t22 w1x1 ks4 D8", 1024, 512, 512, dA, dB, dC);
    run<2, 2, 1, 1, 4, 2, false, false>("NT t22 w1x1 ks4 D2", 1024, 512, 512, dA, dB, dC);
    run<1, 2, 2, 2, 1, 4, false, false>("NT t12 w2x2 ks1 D4", 1024, 512, 512, dA, dB, dC);
    run<1, 2, 2, 2, 1, 8, false, false>("NT t12 w2x2 ks1 D8", 1024, 512, 512, dA, dB, dC);
    run<2, 4, 1, 1, 4, 4, false, false>("NT t24 w1x1 ks4 D4", 1024, 512, 512, dA, dB, dC);
    run<4, 4, 1, 1, 4, 2, false, false>("NT t44 w1x1 ks4 D2", 1024, 512, 512, dA, dB, dC);
    run<4, 4, 2, 2, 1, 2, false, false>("NT t44 w2x2 ks1 D2", 1024, 512, 512, dA, dB, dC);
    printf("--- NN (KC,KM) 1024x512x512\n");
    run<2, 2, 2, 2, 1, 4, false, true>("NN t22 w2x2 ks1 D4", 1024, 512, 512, dA, dB, dC);
    run<2, 2, 1, 1, 4, 4, false, true>("NN t22 w1x1 ks4 D4", 1024, 512, 512, dA, dB, dC);
    run<2, 4, 1, 1, 4, 4, false, true>("NN t24 w1x1 ks4 D4", 1024, 512, 512, dA, dB, dC);
    run<1, 4, 2, 2, 1, 4, false, true>("NN t14 w2x2 ks1 D4", 1024, 512, 512, dA, dB, dC);
    printf("--- TN (KM,KM) 512x512x1024 (wgrad)\n");
    run<2, 2, 1, 1, 4, 4, true, true>("TN t22 w1x1 ks4 D4", 512, 512, 1024, dA, dB, dC);
    run<2, 2, 1, 1, 4, 8, true, true>("TN t22 w1x1 ks4 D8", 512, 512, 1024, dA, dB, dC);
    run<4, 4, 1, 1, 4, 4, true, true>("TN t44 w1x1 ks4 D4", 512, 512, 1024, dA, dB, dC);
    run<4, 2, 1, 1, 4, 4, true, true>("TN t42 w1x1 ks4 D4", 512, 512, 1024, dA, dB, dC);
    run<2, 4, 1, 1, 4, 4, true, true>("TN t24 w1x1 ks4 D4", 512, 512, 1024, dA, dB, dC);
    printf("--- all-KM form of the forward GEMM (operands pre-transposed): 1024x512x512\n");
    run<2, 2, 2, 2, 1, 4, true, true>("TT t22 w2x2 ks1 D4", 1024, 512, 512, dA, dB, dC);
    run<2, 2, 1, 1, 4, 4, true, true>("TT t22 w1x1 ks4 D4", 1024, 512, 512, dA, dB, dC);
    run<4, 4, 1, 1, 4, 4, true, true>("TT t44 w1x1 ks4 D4", 1024, 512, 512, dA, dB, dC);
    run<4, 2, 1, 1, 4, 4, true, true>("TT t42 w1x1 ks4 D4", 1024, 512, 512, dA, dB, dC);
    printf("--- backward stage B1 pieces: TN 256x512x1024 (wgrad) and NN 1024x512x256 (dgrad)\n");
    run<2, 2, 1, 1, 4, 4, true, true>("TN t22 w1x1 ks4 D4", 256, 512, 1024, dA, dB, dC);
    run<1, 2, 1, 1, 4, 4, true, true>("TN t12 w1x1 ks4 D4", 256, 512, 1024, dA, dB, dC);
    run<2, 4, 1, 1, 4, 4, true, true>("TN t24 w1x1 ks4 D4", 256, 512, 1024, dA, dB, dC);
    run<2, 2, 1, 1, 4, 4, false, true>("NN t22 w1x1 ks4 D4", 1024, 512, 256, dA, dB, dC);
    run<2, 4, 1, 1, 4, 4, false, true>("NN t24 w1x1 ks4 D4", 1024, 512, 256, dA, dB, dC);
    run<2, 2, 1, 2, 2, 4, false, true>("NN t22 w1x2 ks2 D4", 1024, 512, 256, dA, dB, dC);
    run<2, 2, 2, 2, 1, 4, false, true>("NN t22 w2x2 ks1 D4", 1024, 512, 256, dA, dB, dC);
    printf("--- B3 pieces: TN 512x64x1024, NN 1024x64x512;  B4 pieces: TN 64x512x1024, NN 1024x512x64\n");
    run<2, 2, 1, 1, 4, 4, true, true>("TN t22 w1x1 ks4 D4", 512, 64, 1024, dA, dB, dC);
    run<1, 2, 1, 1, 4, 4, true, true>("TN t12 w1x1 ks4 D4", 512, 64, 1024, dA, dB, dC);
    run<2, 2, 1, 1, 4, 4, false, true>("NN t22 w1x1 ks4 D4", 1024, 64, 512, dA, dB, dC);
    run<1, 2, 1, 1, 4, 4, false, true>("NN t12 w1x1 ks4 D4", 1024, 64, 512, dA, dB, dC);
    run<2, 2, 1, 1, 4, 4, true, true>("TN t22 w1x1 ks4 D4", 64, 512, 1024, dA, dB, dC);
    run<1, 2, 1, 1, 4, 4, true, true>("TN t12 w1x1 ks4 D4", 64, 512, 1024, dA, dB, dC);
    run<2, 2, 2, 2, 1, 4, false, true>("NN t22 w2x2 ks1 D4", 1024, 512, 64, dA, dB, dC);
    printf("--- skinny: mu/logvar heads NT 1024x64x512, dec1 NT 1024x512x64\n");
    run<2, 2, 1, 1, 4, 4, false, false>("NT t22 w1x1 ks4 D4", 1024, 64, 512, dA, dB, dC);
    run<1, 2, 1, 1, 4, 4, false, false>("NT t12 w1x1 ks4 D4", 1024, 64, 512, dA, dB, dC);
    run<1, 4, 1, 1, 4, 4, false, false>("NT t14 w1x1 ks4 D4", 1024, 64, 512, dA, dB, dC);
    run<2, 2, 2, 2, 1, 4, false, false>("NT t22 w2x2 ks1 D4", 1024, 512, 64, dA, dB, dC);
    run<1, 2, 2, 2, 1, 4, false, false>("NT t12 w2x2 ks1 D4", 1024, 512, 64, dA, dB, dC);
    return 0;
}

// ---- concurrency experiment: a dgrad-like chain and a wgrad-like chain, one stream vs two ----------
extern "C" int conc_main() {
    const int MAXE = 1024 * 1024;
    float *dA, *dB, *dC1, *dC2;
    hipMalloc(&dA, MAXE * 4 * 2); hipMalloc(&dB, MAXE * 4 * 2); hipMalloc(&dC1, MAXE * 4); hipMalloc(&dC2, MAXE * 4);
    hipMemset(dA, 0, MAXE * 8); hipMemset(dB, 0, MAXE * 8);
    auto kd = rd_gemm<2, 4, 1, 1, 4, 4, false, true>;   // NN 1024x512x512, 256 WGs
    auto kw = rd_gemm<2, 2, 1, 1, 4, 4, true, true>;    // TN 512x512x1024, 256 WGs
    const size_t ld_ = 4 * 32 * 68 * 4, lw_ = 4 * 32 * 36 * 4;
    hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ld_);
    hipFuncSetAttribute((const void*)kw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lw_);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t e0, e1, ev[16], ej; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&ej);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const int NCH = 6, REP = 20;
    float ms;
    // (a) one stream, alternating d and w
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0, s1);
        for (int r = 0; r < REP; ++r)
            for (int i = 0; i < NCH; ++i) {
                hipLaunchKernelGGL(kd, dim3(256), dim3(256), ld_, s1, dA, dB, dC1, 1024, 512, 512, 512, 512, 512);
                hipLaunchKernelGGL(kw, dim3(256), dim3(256), lw_, s1, dA + MAXE, dB + MAXE, dC2, 512, 512, 1024, 512, 512, 512);
            }
        hipEventRecord(e1, s1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("one stream  : %.2f us per (dgrad+wgrad) pair\n", ms * 1e3 / (REP * NCH));
    // (b) two streams: d chain on s1; w_i on s2 waits for d_i
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0, s1);
        hipStreamWaitEvent(s2, e0, 0);
        for (int r = 0; r < REP; ++r)
            for (int i = 0; i < NCH; ++i) {
                hipLaunchKernelGGL(kd, dim3(256), dim3(256), ld_, s1, dA, dB, dC1, 1024, 512, 512, 512, 512, 512);
                hipEventRecord(ev[i], s1);
                hipStreamWaitEvent(s2, ev[i], 0);
                hipLaunchKernelGGL(kw, dim3(256), dim3(256), lw_, s2, dA + MAXE, dB + MAXE, dC2, 512, 512, 1024, 512, 512, 512);
            }
        hipEventRecord(ej, s2); hipStreamWaitEvent(s1, ej, 0);
        hipEventRecord(e1, s1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("two streams : %.2f us per (dgrad+wgrad) pair\n", ms * 1e3 / (REP * NCH));
    // (c) same two-stream structure captured in one hipGraph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal);
    hipEventRecord(ev[15], s1); hipStreamWaitEvent(s2, ev[15], 0);
    for (int i = 0; i < NCH; ++i) {
        hipLaunchKernelGGL(kd, dim3(256), dim3(256), ld_, s1, dA, dB, dC1, 1024, 512, 512, 512, 512, 512);
        hipEventRecord(ev[i], s1);
        hipStreamWaitEvent(s2, ev[i], 0);
        hipLaunchKernelGGL(kw, dim3(256), dim3(256), lw_, s2, dA + MAXE, dB + MAXE, dC2, 512, 512, 1024, 512, 512, 512);
    }
    hipEventRecord(ej, s2); hipStreamWaitEvent(s1, ej, 0);
    hipStreamEndCapture(s1, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0, s1);
        for (int r = 0; r < REP; ++r) hipGraphLaunch(ge, s1);
        hipEventRecord(e1, s1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("graph 2-br  : %.2f us per (dgrad+wgrad) pair\n", ms * 1e3 / (REP * NCH));
    return 0;
}
