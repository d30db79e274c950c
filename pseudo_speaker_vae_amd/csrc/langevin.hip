// Fused Langevin conditional sampler (reference loop: /root/reference/ps_vae/inference.py:72-103).
//
// The reference runs, per step, a classifier forward, an autograd backward, a randn, an update, a D2H copy
// of z and two .item() host syncs.  Here the whole chain of `num_steps` updates is ONE kernel: one lane
// owns one latent row for its whole life (z and its gradient live in registers), the classifier weights
// sit in LDS and are read as wave-uniform broadcasts, the noise comes from the same Philox4x32-10 stream as
// the VAE's epsilon, nothing touches the host.  fp32 VALU FMAs run at the fp32 MFMA rate on gfx950, so
// the matrix cores would not be faster for this 64 -> 128 -> C chain.
//
//   logits = W2 relu(W1 z + b1) + b2          (latent_classifier.py:30-40; or logits = W1 z + b1 when hidden == 0)
//   grad   = d/dz [ log_softmax(logits)[target] - 0.5 |z|^2 ]                      (inference.py:56-93)
//   z     += 0.5 s^2 grad + s w noise                                              (inference.py:97)
#include "psvae_common.h"

namespace psvae {

__device__ __forceinline__ void philox4(uint64_t blk, uint64_t seed, uint32_t (&out)[4]) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0 = (uint32_t)blk, c1 = (uint32_t)(blk >> 32), c2 = 0, c3 = 0, k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0, hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0; k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ f32x4 normal4(uint64_t blk, uint64_t seed) {
    uint32_t r[4];
    philox4(blk, seed, r);
    constexpr float S = 5.9604644775390625e-08f, TWO_PI = 6.283185307179586f;
    const float u0 = ((float)(r[0] >> 8) + 0.5f) * S, u1 = ((float)(r[1] >> 8) + 0.5f) * S;
    const float u2 = ((float)(r[2] >> 8) + 0.5f) * S, u3 = ((float)(r[3] >> 8) + 0.5f) * S;
    const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
    float sa, ca, sb, cb;
    sincosf(TWO_PI * u1, &sa, &ca);
    sincosf(TWO_PI * u3, &sb, &cb);
    return f32x4{ra * ca, ra * sa, rb * cb, rb * sb};
}

constexpr int LV_MAXC = 8;

// ZD = latent dim (compile time: z and its gradient are register arrays), hidden/C runtime.
template <int ZD>
__global__ __launch_bounds__(256) void langevin_kernel(int64_t N, int hidden, int C, const float* __restrict__ W1,
                                                       const float* __restrict__ b1, const float* __restrict__ W2,
                                                       const float* __restrict__ b2, float* __restrict__ zio, int target,
                                                       float step_size, float noise_weight, int64_t num_steps,
                                                       const float* __restrict__ noises, uint64_t seed, uint64_t offset,
                                                       float* __restrict__ history) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int rows1 = hidden > 0 ? hidden : C;  // rows of the first Linear
    float* sW1 = sm;                            // [rows1][ZD]
    float* sb1 = sW1 + rows1 * ZD;              // [rows1]
    float* sW2 = sb1 + ((rows1 + 3) & ~3);      // [C][hidden]
    float* sb2 = sW2 + C * (hidden > 0 ? hidden : 0);
    for (int i = threadIdx.x; i < rows1 * ZD; i += 256) sW1[i] = W1[i];
    for (int i = threadIdx.x; i < rows1; i += 256) sb1[i] = b1[i];
    if (hidden > 0) {
        for (int i = threadIdx.x; i < C * hidden; i += 256) sW2[i] = W2[i];
        for (int i = threadIdx.x; i < C; i += 256) sb2[i] = b2[i];
    }
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = row < N;
    const int64_t rr = live ? row : N - 1;  // dead lanes shadow the last row (no stores)
    float z[ZD];
#pragma unroll
    for (int k = 0; k < ZD; k += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(zio + rr * ZD + k);
        z[k] = v[0]; z[k + 1] = v[1]; z[k + 2] = v[2]; z[k + 3] = v[3];
    }
    const float drift = 0.5f * step_size * step_size, nsc = step_size * noise_weight;
    const uint64_t blocks_per_step = (uint64_t)((N * ZD + 3) / 4);
    for (int64_t t = 0; t < num_steps; ++t) {
        float logits[LV_MAXC];
        float gz[ZD];
        uint32_t mask[4] = {0u, 0u, 0u, 0u};  // ReLU mask of up to 128 hidden units
        if (hidden > 0) {
#pragma unroll
            for (int c = 0; c < LV_MAXC; ++c) logits[c] = c < C ? sb2[c] : 0.f;
            for (int j = 0; j < hidden; ++j) {  // pass 1: hidden unit j on the fly
                // row j of W1: wave-uniform LDS address = broadcast read; 4 partial sums break the dependent
                // FMA chain (reading the row through the scalar cache was tried: hipcc emits vector loads, slower)
                const f32x4* w = reinterpret_cast<const f32x4*>(sW1 + j * ZD);
                float a0 = sb1[j], a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int k4 = 0; k4 < ZD / 4; ++k4) {
                    const f32x4 wv = w[k4];
                    a0 = fmaf(wv[0], z[4 * k4], a0); a1 = fmaf(wv[1], z[4 * k4 + 1], a1);
                    a2 = fmaf(wv[2], z[4 * k4 + 2], a2); a3 = fmaf(wv[3], z[4 * k4 + 3], a3);
                }
                float a = (a0 + a1) + (a2 + a3);
                const bool on = a > 0.f;
                a = on ? a : 0.f;
                if (on) mask[j >> 5] |= 1u << (j & 31);
#pragma unroll
                for (int c = 0; c < LV_MAXC; ++c)
                    if (c < C) logits[c] = fmaf(sW2[c * hidden + j], a, logits[c]);
            }
        } else {
#pragma unroll
            for (int c = 0; c < LV_MAXC; ++c) {
                float a = 0.f;
                if (c < C) {
                    a = sb1[c];
#pragma unroll
                    for (int k = 0; k < ZD; ++k) a = fmaf(sW1[c * ZD + k], z[k], a);
                }
                logits[c] = a;
            }
        }
        // d log_softmax[target] / d logits = onehot(target) - softmax(logits)     (inference.py:56-70)
        float mx = -3.0e38f;
#pragma unroll
        for (int c = 0; c < LV_MAXC; ++c)
            if (c < C) mx = fmaxf(mx, logits[c]);
        float den = 0.f, gl[LV_MAXC];
#pragma unroll
        for (int c = 0; c < LV_MAXC; ++c) {
            gl[c] = c < C ? expf(logits[c] - mx) : 0.f;
            den += gl[c];
        }
#pragma unroll
        for (int c = 0; c < LV_MAXC; ++c) gl[c] = (c == target ? 1.f : 0.f) - gl[c] / den;
        if (C == 1) gl[0] = 0.f;  // log_softmax over one logit is identically 0 (inference.py:60-66 as written)
#pragma unroll
        for (int k = 0; k < ZD; ++k) gz[k] = -z[k];  // d(-0.5|z|^2)/dz           (inference.py:90)
        if (hidden > 0) {
            for (int j = 0; j < hidden; ++j) {  // pass 2: back through hidden unit j
                float gh = 0.f;
#pragma unroll
                for (int c = 0; c < LV_MAXC; ++c)
                    if (c < C) gh = fmaf(sW2[c * hidden + j], gl[c], gh);
                gh = (mask[j >> 5] >> (j & 31)) & 1u ? gh : 0.f;
                const f32x4* w = reinterpret_cast<const f32x4*>(sW1 + j * ZD);
#pragma unroll
                for (int k4 = 0; k4 < ZD / 4; ++k4) {
                    const f32x4 wv = w[k4];
                    gz[4 * k4] = fmaf(wv[0], gh, gz[4 * k4]); gz[4 * k4 + 1] = fmaf(wv[1], gh, gz[4 * k4 + 1]);
                    gz[4 * k4 + 2] = fmaf(wv[2], gh, gz[4 * k4 + 2]); gz[4 * k4 + 3] = fmaf(wv[3], gh, gz[4 * k4 + 3]);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < LV_MAXC; ++c)
                if (c < C) {
#pragma unroll
                    for (int k = 0; k < ZD; ++k) gz[k] = fmaf(sW1[c * ZD + k], gl[c], gz[k]);
                }
        }
        // update                                                                 (inference.py:95-97)
#pragma unroll
        for (int k = 0; k < ZD; k += 4) {
            f32x4 nz;
            if (noises) {
                nz = *reinterpret_cast<const f32x4*>(noises + ((size_t)t * N + rr) * ZD + k);
            } else {
                nz = normal4(offset + (uint64_t)t * blocks_per_step + (uint64_t)((rr * ZD + k) >> 2), seed);
            }
            z[k] += drift * gz[k] + nsc * nz[0];
            z[k + 1] += drift * gz[k + 1] + nsc * nz[1];
            z[k + 2] += drift * gz[k + 2] + nsc * nz[2];
            z[k + 3] += drift * gz[k + 3] + nsc * nz[3];
            if (history && live)
                *reinterpret_cast<f32x4*>(history + ((size_t)t * N + row) * ZD + k) = f32x4{z[k], z[k + 1], z[k + 2], z[k + 3]};
        }
    }
    if (live) {
#pragma unroll
        for (int k = 0; k < ZD; k += 4)
            *reinterpret_cast<f32x4*>(zio + row * ZD + k) = f32x4{z[k], z[k + 1], z[k + 2], z[k + 3]};
    }
}

hipError_t launch_langevin(int64_t N, int Z, int hidden, int C, const float* W1, const float* b1, const float* W2,
                           const float* b2, float* z, int target, float step_size, float noise_weight,
                           int64_t num_steps, const float* noises, uint64_t seed, uint64_t offset, float* history,
                           hipStream_t s) {
    const int rows1 = hidden > 0 ? hidden : C;
    const size_t lds = (size_t)(rows1 * Z + ((rows1 + 3) & ~3) + C * hidden + 8) * sizeof(float);
    const int grid = (int)((N + 255) / 256);
#define PSVAE_LV(ZD)                                                                                                   \
    case ZD: {                                                                                                         \
        static bool done = false;                                                                                      \
        if (!done && lds > 48 * 1024) {                                                                                \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&langevin_kernel<ZD>),                   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);               \
            if (ea != hipSuccess) return ea;                                                                           \
            done = true;                                                                                               \
        }                                                                                                              \
        hipLaunchKernelGGL(langevin_kernel<ZD>, dim3(grid), dim3(256), lds, s, N, hidden, C, W1, b1, W2, b2, z, target, \
                           step_size, noise_weight, num_steps, noises, seed, offset, history);                         \
        return hipGetLastError();                                                                                      \
    }
    switch (Z) {
        PSVAE_LV(16) PSVAE_LV(32) PSVAE_LV(64) PSVAE_LV(128)
        default: return hipErrorInvalidValue;
    }
#undef PSVAE_LV
}

}  // namespace psvae
