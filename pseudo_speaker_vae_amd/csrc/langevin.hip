// Fused Langevin conditional sampler (reference loop: /root/reference/ps_vae/inference.py:72-103).
//
// The reference runs, per step, a classifier forward, an autograd backward, a randn, an update, a D2H copy
// of z and two .item() host syncs.  Here the whole chain of `num_steps` updates is ONE kernel: one lane
// owns one latent row for its whole life (z and its gradient live in registers), the classifier weights
// sit in LDS and are read as wave-uniform broadcasts, the noise comes from the same Philox4x32-10 stream as
// the VAE's epsilon, nothing touches the host.  That per-lane form (langevin_kernel) serves every supported shape.
//
// For heads with a hidden layer whose sizes are whole 16-blocks (the one the reference trains: 64 -> 128 -> C) the
// chain runs on the matrix cores instead (langevin_mfma_kernel): fp32 MFMA and fp32 VALU have the same peak, but the
// per-lane form leaves the 64 -> 128 layer as 2 x 8,192 SERIAL FMAs per lane and step fed by LDS broadcasts (13.6 % of
// the vector peak measured, 195 ms for 1 M rows x 100 steps), while one wavefront per 16 rows does the same layer as
// 2 x 128 MFMAs whose operands never leave the register file.
//
//   logits = W2 relu(W1 z + b1) + b2          (latent_classifier.py:30-40; or logits = W1 z + b1 when hidden == 0)
//   grad   = d/dz [ log_softmax(logits)[target] - 0.5 |z|^2 ]                      (inference.py:56-93)
//   z     += 0.5 s^2 grad + s w noise                                              (inference.py:97)

#include "psvae_common.h"
#include "psvae_device.h"

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
// The chain's noise: Philox4x32-10 blocks (the VAE's epsilon stream) through Box-Muller on the HARDWARE transcendentals
// (v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32, psvae_device.h philox_normal4_hw) -- what psvae_philox_normal_fast_f32 writes,
// bit for bit.  With libm's logf / sincosf the chain on the matrix cores was bound by the vector ALU's issue port, not by its
// MFMAs: 6.9 vector instructions per MFMA (rocprofv3: SQ_INSTS_VALU 10.96 G against SQ_INSTS_MFMA 1.6 G per 1 M rows x 100
// steps), about 1,000 of a step's 1,750 in the noise.
__device__ __forceinline__ f32x4 normal4(uint64_t blk, uint64_t seed) {
#ifdef PSVAE_LANGEVIN_EXACT_NOISE  // (the libm form, for A/B runs)
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
#else
    return philox_normal4_hw(blk, seed);
#endif
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

// ---------------------------------------------------------------------------------------------------------------
// MFMA form.  One wavefront owns 16 latent rows for the whole chain.  Everything is kept TRANSPOSED so that a row's data
// stays in the lanes r = lane & 15 of its wavefront through both products (no LDS transposes, no shuffles but one):
//   h^T  [hidden x 16 rows] = W1 [hidden x Z] . z^T [Z x 16]            A = W1 fragments,   B = z,   C init = b1
//   gz^T [Z x 16 rows]      = W1^T [Z x hidden] . dh^T [hidden x 16]    A = W1^T fragments, B = dh,  C init = -z (prior)
// v_mfma_f32_16x16x4_f32: lane (r, q) supplies A[i = r][k = q], B[k = q][n = r] and receives D[i = 4q + j][n = r], j < 4.
// So lane (r, q) holds, for its row r: z / gz at features 16 m' + 4q + j (m' < ZB: 4 consecutive features = one Philox
// block per m'), h / dh at hidden units 16 m + 4q + j (m < HB).  The contraction index of each product is enumerated in
// exactly the order the lanes hold the B operand (k slot q of MFMA step (m', j) is feature 16 m' + 4q + j), and the A
// fragments of W1 -- both orientations, 8 ZB HB registers -- are gathered ONCE per wavefront to match: the loop issues
// nothing but MFMAs, the Philox / softmax / update VALU work and C x HB 16-byte LDS reads of W2.
// ---------------------------------------------------------------------------------------------------------------
template <int ZB, int HB>
__global__ __launch_bounds__(256) void langevin_mfma_kernel(int64_t N, int C, const float* __restrict__ W1,
                                                            const float* __restrict__ b1, const float* __restrict__ W2,
                                                            const float* __restrict__ b2, float* __restrict__ zio,
                                                            int target, float step_size, float noise_weight,
                                                            int64_t num_steps, const float* __restrict__ noises,
                                                            uint64_t seed, uint64_t offset, float* __restrict__ history) {
    constexpr int ZD = 16 * ZB, HD = 16 * HB;
    __shared__ __attribute__((aligned(16))) float sW2[LV_MAXC * HD];
    __shared__ float sb2[LV_MAXC];
    for (int i = threadIdx.x; i < C * HD; i += 256) sW2[i] = W2[i];
    if (threadIdx.x < C) sb2[threadIdx.x] = b2[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * 16 + r;
    const bool live = row < N;
    const int64_t rr = live ? row : N - 1;  // dead lanes shadow the last row (no stores)
    // W1 fragments, gathered once: forward A[i = hidden 16m + r][k = feature 16m' + 4q + j'], backward
    // A[i = feature 16m' + r][k = hidden 16m + 4q + j]
    float wf[HB][4 * ZB], wb[ZB][4 * HB], bias1[HB][4];
#pragma unroll
    for (int m = 0; m < HB; ++m) {
#pragma unroll
        for (int mp = 0; mp < ZB; ++mp) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(W1 + (size_t)(16 * m + r) * ZD + 16 * mp + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[m][4 * mp + j] = v[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) bias1[m][j] = b1[16 * m + 4 * q + j];
    }
#pragma unroll
    for (int mp = 0; mp < ZB; ++mp)
#pragma unroll
        for (int m = 0; m < HB; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) wb[mp][4 * m + j] = W1[(size_t)(16 * m + 4 * q + j) * ZD + 16 * mp + r];
    float z[4 * ZB];
#pragma unroll
    for (int mp = 0; mp < ZB; ++mp) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(zio + rr * ZD + 16 * mp + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) z[4 * mp + j] = v[j];
    }
    const float drift = 0.5f * step_size * step_size, nsc = step_size * noise_weight;
    const uint64_t blocks_per_step = (uint64_t)((N * ZD + 3) / 4);
    for (int64_t t = 0; t < num_steps; ++t) {
        // this step's noise first: it depends on nothing, so its VALU work can fill the MFMA shadows below
        f32x4 nz[ZB];
#pragma unroll
        for (int mp = 0; mp < ZB; ++mp) {
            if (noises)
                nz[mp] = *reinterpret_cast<const f32x4*>(noises + ((size_t)t * N + rr) * ZD + 16 * mp + 4 * q);
            else
                nz[mp] = normal4(offset + (uint64_t)t * blocks_per_step + (uint64_t)((rr * ZD + 16 * mp + 4 * q) >> 2), seed);
        }
        // ---- h^T = W1 z^T + b1 ----
        f32x4 h[HB];
#pragma unroll
        for (int m = 0; m < HB; ++m) h[m] = f32x4{bias1[m][0], bias1[m][1], bias1[m][2], bias1[m][3]};
#pragma unroll
        for (int s = 0; s < 4 * ZB; ++s)
#pragma unroll
            for (int m = 0; m < HB; ++m) h[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[m][s], z[s], h[m], 0, 0, 0);
        // ---- logits: this lane's 4 HB hidden units, then the sum over the 4 lanes of the row ----
        float logits[LV_MAXC];
#pragma unroll
        for (int c = 0; c < LV_MAXC; ++c) logits[c] = 0.f;
#pragma unroll
        for (int m = 0; m < HB; ++m) {
#pragma unroll
            for (int j = 0; j < 4; ++j) h[m][j] = fmaxf(h[m][j], 0.f);  // (mask = h > 0)
#pragma unroll
            for (int c = 0; c < LV_MAXC; ++c)
                if (c < C) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(sW2 + c * HD + 16 * m + 4 * q);
                    logits[c] = fmaf(w[0], h[m][0], fmaf(w[1], h[m][1], fmaf(w[2], h[m][2], fmaf(w[3], h[m][3], logits[c]))));
                }
        }
#pragma unroll
        for (int c = 0; c < LV_MAXC; ++c)
            if (c < C) {
                logits[c] += __shfl_xor(logits[c], 16);
                logits[c] += __shfl_xor(logits[c], 32);
                logits[c] += sb2[c];
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
        // ---- dh = (W2^T gl) * [h > 0]  (in place of h) ----
#pragma unroll
        for (int m = 0; m < HB; ++m) {
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < LV_MAXC; ++c)
                if (c < C) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(sW2 + c * HD + 16 * m + 4 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = fmaf(w[j], gl[c], g[j]);
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) h[m][j] = h[m][j] > 0.f ? g[j] : 0.f;
        }
        // ---- gz^T = W1^T dh^T - z ----
        f32x4 gz[ZB];
#pragma unroll
        for (int mp = 0; mp < ZB; ++mp) gz[mp] = f32x4{-z[4 * mp], -z[4 * mp + 1], -z[4 * mp + 2], -z[4 * mp + 3]};
#pragma unroll
        for (int m = 0; m < HB; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mp = 0; mp < ZB; ++mp)
                    gz[mp] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[mp][4 * m + j], h[m][j], gz[mp], 0, 0, 0);
        // ---- update (inference.py:95-97) ----
#pragma unroll
        for (int mp = 0; mp < ZB; ++mp) {
#pragma unroll
            for (int j = 0; j < 4; ++j) z[4 * mp + j] += drift * gz[mp][j] + nsc * nz[mp][j];
            if (history && live)
                *reinterpret_cast<f32x4*>(history + ((size_t)t * N + row) * ZD + 16 * mp + 4 * q) =
                    f32x4{z[4 * mp], z[4 * mp + 1], z[4 * mp + 2], z[4 * mp + 3]};
        }
    }
    if (live) {
#pragma unroll
        for (int mp = 0; mp < ZB; ++mp)
            *reinterpret_cast<f32x4*>(zio + row * ZD + 16 * mp + 4 * q) = f32x4{z[4 * mp], z[4 * mp + 1], z[4 * mp + 2], z[4 * mp + 3]};
    }
}

// Same chain with the W1 fragments in LDS instead of registers: the register-resident form above fills the register file
// (one wavefront per SIMD), so a step's Philox / softmax / update VALU work and its MFMAs take turns.  Here each lane's
// fragments are pre-arranged in LDS as contiguous 16-byte pieces in exactly the order the loop consumes them
//   forward : Wf[s][m/4][q][r][m%4] (k step s = 4 m' + j', hidden block m) -> two ds_read_b128 per k step (HB = 8)
//   backward: Wb[s2][q][r][m'] (k step s2 = 4 m + j, feature block m')  -> one ds_read_b128 per k step (ZB = 4)
// (a wavefront reads 64 consecutive pieces: conflict-free), the kernel needs < 128 registers, and 512-thread persistent
// workgroups put four wavefronts on every SIMD: one wavefront's VALU work runs under the others' MFMAs.
template <int ZB, int HB>
__global__ __launch_bounds__(512, 4) void langevin_mfma_lds_kernel(int64_t N, int C, const float* __restrict__ W1,
                                                                   const float* __restrict__ b1,
                                                                   const float* __restrict__ W2,
                                                                   const float* __restrict__ b2, float* __restrict__ zio,
                                                                   int target, float step_size, float noise_weight,
                                                                   int64_t num_steps, const float* __restrict__ noises,
                                                                   uint64_t seed, uint64_t offset,
                                                                   float* __restrict__ history) {
    constexpr int ZD = 16 * ZB, HD = 16 * HB;
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    float* sWf = lsm;                       // [4 ZB][HB/4][4][16][4]
    float* sWb = sWf + 4 * ZB * 64 * HB;    // [4 HB][4][16][ZB]
    float* sW2 = sWb + 4 * HB * 64 * ZB;    // [C][HD]
    float* sb1 = sW2 + LV_MAXC * HD;        // [HD]
    float* sb2 = sb1 + HD;                  // [C]
    for (int i = threadIdx.x; i < 4 * ZB * 64 * HB; i += 512) {
        const int m = 4 * ((i >> 8) % (HB / 4)) + (i & 3), rr_ = (i >> 2) & 15, qq = (i >> 6) & 3, s_ = i / (HB * 64);
        sWf[i] = W1[(size_t)(16 * m + rr_) * ZD + 16 * (s_ >> 2) + 4 * qq + (s_ & 3)];
    }
    for (int i = threadIdx.x; i < 4 * HB * 64 * ZB; i += 512) {
        const int mp = i % ZB, rr_ = (i / ZB) % 16, qq = (i / (ZB * 16)) % 4, s2 = i / (ZB * 64);
        sWb[i] = W1[(size_t)(16 * (s2 >> 2) + 4 * qq + (s2 & 3)) * ZD + 16 * mp + rr_];
    }
    for (int i = threadIdx.x; i < C * HD; i += 512) sW2[i] = W2[i];
    for (int i = threadIdx.x; i < HD; i += 512) sb1[i] = b1[i];
    if (threadIdx.x < C) sb2[threadIdx.x] = b2[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const float drift = 0.5f * step_size * step_size, nsc = step_size * noise_weight;
    const uint64_t blocks_per_step = (uint64_t)((N * ZD + 3) / 4);
    const int64_t ngroups = (N + 15) / 16;
    const float* __restrict__ myWf = sWf + lane * 4;
    const float* __restrict__ myWb = sWb + (q * 16 + r) * ZB;
    for (int64_t g = (int64_t)blockIdx.x * 8 + wave; g < ngroups; g += (int64_t)gridDim.x * 8) {
        const int64_t row = g * 16 + r;
        const bool live = row < N;
        const int64_t rr = live ? row : N - 1;
        float z[4 * ZB];
#pragma unroll
        for (int mp = 0; mp < ZB; ++mp) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(zio + rr * ZD + 16 * mp + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[4 * mp + j] = v[j];
        }
        for (int64_t t = 0; t < num_steps; ++t) {
            f32x4 h[HB];
#pragma unroll
            for (int m = 0; m < HB; ++m) h[m] = *reinterpret_cast<const f32x4*>(sb1 + 16 * m + 4 * q);
            {   // one k step of fragments ahead of the MFMAs; the sched_barriers keep the compiler from hoisting ALL
                // fragment reads to the top (it did: 768 B of scratch per lane at the 128-register budget)
                f32x4 wn[HB / 4];
#pragma unroll
                for (int m4 = 0; m4 < HB / 4; ++m4) wn[m4] = *reinterpret_cast<const f32x4*>(myWf + 256 * m4);
#pragma unroll
                for (int s = 0; s < 4 * ZB; ++s) {
                    f32x4 w[HB / 4];
#pragma unroll
                    for (int m4 = 0; m4 < HB / 4; ++m4) w[m4] = wn[m4];
                    if (s + 1 < 4 * ZB) {
#pragma unroll
                        for (int m4 = 0; m4 < HB / 4; ++m4) wn[m4] = *reinterpret_cast<const f32x4*>(myWf + (s + 1) * 64 * HB + 256 * m4);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < HB; ++m) h[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[m >> 2][m & 3], z[s], h[m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            float logits[LV_MAXC];
#pragma unroll
            for (int c = 0; c < LV_MAXC; ++c) logits[c] = 0.f;
#pragma unroll
            for (int m = 0; m < HB; ++m) {
#pragma unroll
                for (int j = 0; j < 4; ++j) h[m][j] = fmaxf(h[m][j], 0.f);
#pragma unroll
                for (int c = 0; c < LV_MAXC; ++c)
                    if (c < C) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(sW2 + c * HD + 16 * m + 4 * q);
                        logits[c] = fmaf(w[0], h[m][0], fmaf(w[1], h[m][1], fmaf(w[2], h[m][2], fmaf(w[3], h[m][3], logits[c]))));
                    }
            }
#pragma unroll
            for (int c = 0; c < LV_MAXC; ++c)
                if (c < C) {
                    logits[c] += __shfl_xor(logits[c], 16);
                    logits[c] += __shfl_xor(logits[c], 32);
                    logits[c] += sb2[c];
                }
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
            if (C == 1) gl[0] = 0.f;
#pragma unroll
            for (int m = 0; m < HB; ++m) {
                f32x4 gacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < LV_MAXC; ++c)
                    if (c < C) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(sW2 + c * HD + 16 * m + 4 * q);
#pragma unroll
                        for (int j = 0; j < 4; ++j) gacc[j] = fmaf(w[j], gl[c], gacc[j]);
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j) h[m][j] = h[m][j] > 0.f ? gacc[j] : 0.f;
            }
            f32x4 gz[ZB];
#pragma unroll
            for (int mp = 0; mp < ZB; ++mp) gz[mp] = f32x4{-z[4 * mp], -z[4 * mp + 1], -z[4 * mp + 2], -z[4 * mp + 3]};
            {
                f32x4 wn[ZB / 4];
#pragma unroll
                for (int m4 = 0; m4 < ZB / 4; ++m4) wn[m4] = *reinterpret_cast<const f32x4*>(myWb + 4 * m4);
#pragma unroll
                for (int s2 = 0; s2 < 4 * HB; ++s2) {
                    f32x4 w[ZB / 4];
#pragma unroll
                    for (int m4 = 0; m4 < ZB / 4; ++m4) w[m4] = wn[m4];
                    if (s2 + 1 < 4 * HB) {
#pragma unroll
                        for (int m4 = 0; m4 < ZB / 4; ++m4) wn[m4] = *reinterpret_cast<const f32x4*>(myWb + (s2 + 1) * 64 * ZB + 4 * m4);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mp = 0; mp < ZB; ++mp)
                        gz[mp] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[mp >> 2][mp & 3], h[s2 >> 2][s2 & 3], gz[mp], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int mp = 0; mp < ZB; ++mp) {
                // this step's noise: generated here, at its use (the other wavefronts of the SIMD issue MFMAs meanwhile)
                const f32x4 nzv = noises ? *reinterpret_cast<const f32x4*>(noises + ((size_t)t * N + rr) * ZD + 16 * mp + 4 * q)
                                         : normal4(offset + (uint64_t)t * blocks_per_step + (uint64_t)((rr * ZD + 16 * mp + 4 * q) >> 2), seed);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[4 * mp + j] += drift * gz[mp][j] + nsc * nzv[j];
                if (history && live)
                    *reinterpret_cast<f32x4*>(history + ((size_t)t * N + row) * ZD + 16 * mp + 4 * q) =
                        f32x4{z[4 * mp], z[4 * mp + 1], z[4 * mp + 2], z[4 * mp + 3]};
            }
        }
        if (live) {
#pragma unroll
            for (int mp = 0; mp < ZB; ++mp)
                *reinterpret_cast<f32x4*>(zio + row * ZD + 16 * mp + 4 * q) = f32x4{z[4 * mp], z[4 * mp + 1], z[4 * mp + 2], z[4 * mp + 3]};
        }
    }
}

template <int ZB, int HB>
static hipError_t launch_langevin_mfma_lds(int64_t N, int C, const float* W1, const float* b1, const float* W2,
                                           const float* b2, float* z, int target, float step_size, float noise_weight,
                                           int64_t num_steps, const float* noises, uint64_t seed, uint64_t offset,
                                           float* history, hipStream_t s) {
    constexpr size_t lds = (size_t)(2 * 4 * ZB * 64 * HB + LV_MAXC * 16 * HB + 16 * HB + LV_MAXC) * sizeof(float);
    static std::atomic<unsigned long long> lds_ok{0};  // per device: psvae_common.h lds_opt_in
    if (hipError_t attr_err = lds_opt_in(lds_ok, reinterpret_cast<const void*>(&langevin_mfma_lds_kernel<ZB, HB>), 160 * 1024))
        return attr_err;
    const int64_t groups = (N + 15) / 16, wgs = (groups + 7) / 8;
    const int per_cu = (int)((160 * 1024) / lds) >= 2 ? 2 : 1;
    const int grid = (int)(wgs < 256 * per_cu ? wgs : 256 * per_cu);  // persistent: every workgroup walks its row groups
    hipLaunchKernelGGL((langevin_mfma_lds_kernel<ZB, HB>), dim3(grid), dim3(512), lds, s, N, C, W1, b1, W2, b2, z, target,
                       step_size, noise_weight, num_steps, noises, seed, offset, history);
    return hipGetLastError();
}

template <int ZB, int HB>
static hipError_t launch_langevin_mfma(int64_t N, int C, const float* W1, const float* b1, const float* W2, const float* b2,
                                       float* z, int target, float step_size, float noise_weight, int64_t num_steps,
                                       const float* noises, uint64_t seed, uint64_t offset, float* history, hipStream_t s) {
    const int grid = (int)((N + 63) / 64);
    hipLaunchKernelGGL((langevin_mfma_kernel<ZB, HB>), dim3(grid), dim3(256), 0, s, N, C, W1, b1, W2, b2, z, target,
                       step_size, noise_weight, num_steps, noises, seed, offset, history);
    return hipGetLastError();
}

hipError_t launch_langevin(int64_t N, int Z, int hidden, int C, const float* W1, const float* b1, const float* W2,
                           const float* b2, float* z, int target, float step_size, float noise_weight,
                           int64_t num_steps, const float* noises, uint64_t seed, uint64_t offset, float* history,
                           hipStream_t s) {
    // heads whose two sizes are whole 16-blocks and whose W1 fragments fit the register file run on the matrix cores
    static const bool mfma_on = getenv("PSVAE_LANGEVIN_VALU") == nullptr;
    static const bool lds_form = getenv("PSVAE_LANGEVIN_REGS") == nullptr;  // (0: W1 fragments in registers, 1 wave per SIMD)
    if (mfma_on && lds_form && hidden > 0 && (reinterpret_cast<uintptr_t>(W1) & 15) == 0) {
#define PSVAE_LVL(ZB, HB)                                                                                               \
    if (Z == 16 * ZB && hidden == 16 * HB)                                                                              \
        return launch_langevin_mfma_lds<ZB, HB>(N, C, W1, b1, W2, b2, z, target, step_size, noise_weight, num_steps,    \
                                                noises, seed, offset, history, s);
        PSVAE_LVL(4, 8) PSVAE_LVL(4, 4) PSVAE_LVL(8, 4) PSVAE_LVL(8, 8)
#undef PSVAE_LVL
    }
    if (mfma_on && hidden > 0 && (reinterpret_cast<uintptr_t>(W1) & 15) == 0) {
#define PSVAE_LVM(ZB, HB)                                                                                           \
    if (Z == 16 * ZB && hidden == 16 * HB)                                                                          \
        return launch_langevin_mfma<ZB, HB>(N, C, W1, b1, W2, b2, z, target, step_size, noise_weight, num_steps,    \
                                            noises, seed, offset, history, s);
        PSVAE_LVM(4, 8) PSVAE_LVM(4, 4) PSVAE_LVM(2, 8) PSVAE_LVM(2, 4) PSVAE_LVM(8, 4) PSVAE_LVM(1, 8)
#undef PSVAE_LVM
    }
    const int rows1 = hidden > 0 ? hidden : C;
    const size_t lds = (size_t)(rows1 * Z + ((rows1 + 3) & ~3) + C * hidden + 8) * sizeof(float);
    const int grid = (int)((N + 255) / 256);
#define PSVAE_LV(ZD)                                                                                                   \
    case ZD: {                                                                                                         \
        static std::atomic<unsigned long long> lds_ok{0}; /* per device: psvae_common.h lds_opt_in */                 \
        if (lds > 48 * 1024) {                                                                                         \
            hipError_t ea = lds_opt_in(lds_ok, reinterpret_cast<const void*>(&langevin_kernel<ZD>), 160 * 1024);       \
            if (ea != hipSuccess) return ea;                                                                           \
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
