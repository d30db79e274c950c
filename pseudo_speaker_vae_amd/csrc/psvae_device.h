// Device-side helpers shared by the elementwise / row kernels (gfx950): Philox4x32-10 + Box-Muller, wave and block sums.
#pragma once
#include "psvae_common.h"

namespace psvae {

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011).  One lane = one Philox block = 4 normals for 4 consecutive
// elements, so a wavefront covers 256 contiguous elements and stores 16 B per lane.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        // (one 32 x 32 -> 64 multiply per product -- v_mad_u64_u32 -- instead of a v_mul_hi_u32 / v_mul_lo_u32 pair: both
        //  are quarter-rate instructions, and a chain that draws inside its loop pays for every one of them)
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Box-Muller with libm's logf / sqrtf / sincosf: the stream psvae_philox_normal_f32 writes (the oracle's numpy restatement
// follows it to rounding).  In-kernel draws use philox_normal4 below.
__device__ __forceinline__ f32x4 philox_normal4_exact(uint64_t blk, uint64_t seed) {
    uint32_t r[4];
    philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    constexpr float S = 5.9604644775390625e-08f;  // 2^-24
    const float u0 = ((float)(r[0] >> 8) + 0.5f) * S, u1 = ((float)(r[1] >> 8) + 0.5f) * S;
    const float u2 = ((float)(r[2] >> 8) + 0.5f) * S, u3 = ((float)(r[3] >> 8) + 0.5f) * S;
    constexpr float TWO_PI = 6.283185307179586f;
    const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
    float sa, ca, sb, cb;
    sincosf(TWO_PI * u1, &sa, &ca);
    sincosf(TWO_PI * u3, &sb, &cb);
    return f32x4{ra * ca, ra * sa, rb * cb, rb * sb};
}

// The same draw with the hardware transcendentals only (v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32 -- which take their
// argument in REVOLUTIONS, so sin(2 pi u) is v_sin_f32(u)): straight-line code, no range-reduction branches, which is what
// lets the scheduler weave it between MFMAs (the fused decode kernels draw z this way: psvae_sample_decode_*).  Differs from
// philox_normal4 by the instructions' accuracy, a few 1e-7 of |z| (tests hold it to 2e-6 of the largest |z|).
__device__ __forceinline__ f32x4 philox_normal4_hw(uint64_t blk, uint64_t seed) {
    uint32_t r[4];
    philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    constexpr float S = 5.9604644775390625e-08f;  // 2^-24
    const float u0 = ((float)(r[0] >> 8) + 0.5f) * S, u1 = ((float)(r[1] >> 8) + 0.5f) * S;
    const float u2 = ((float)(r[2] >> 8) + 0.5f) * S, u3 = ((float)(r[3] >> 8) + 0.5f) * S;
    constexpr float M2LN2 = -1.3862943611198906f;  // -2 ln 2: -2 ln u = M2LN2 * log2 u
    const float ra = __builtin_amdgcn_sqrtf(M2LN2 * __builtin_amdgcn_logf(u0));
    const float rb = __builtin_amdgcn_sqrtf(M2LN2 * __builtin_amdgcn_logf(u2));
    return f32x4{ra * __builtin_amdgcn_cosf(u1), ra * __builtin_amdgcn_sinf(u1), rb * __builtin_amdgcn_cosf(u3),
                 rb * __builtin_amdgcn_sinf(u3)};
}
// What every kernel that draws INSIDE its work uses -- the training passes' epsilon (model.py:57 torch.randn_like; the fused
// step's bottleneck kernel, the separate reparameterisation kernels, the bf16 decoder chain), the samplers' z and noise: the
// hardware form.  Measured against the libm form: cfg2 127.6 -> 126.9 us per step, cfg3 94.4 -> 93.2; largest deviation over
// 4 M normals 1.8e-6 (3.3e-7 of the largest |z|); psvae_philox_normal_fast_f32 writes exactly this stream.
__device__ __forceinline__ f32x4 philox_normal4(uint64_t blk, uint64_t seed) { return philox_normal4_hw(blk, seed); }

// ---------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// fixed-order sum over the 256 threads of a block; result valid in thread 0
__device__ __forceinline__ float block_sum_256(float v, float* sh4) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
}

}  // namespace psvae
