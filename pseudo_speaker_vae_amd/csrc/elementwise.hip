// Elementwise / row kernels of the VAE hot path (gfx950): Philox epsilon + reparameterisation
// (model.py:56-57), row L2 normalisation (model.py:60-61), ELBO loss reduction (lightning.py:110-124)
// and the closed-form gradients of those pieces.  All HBM/L2-bound byte work:
// 16 B per lane where alignment allows, one wavefront per row for the row reductions, fixed-order
// two-stage reductions (bitwise reproducible; no float atomics).
#include <hip/hip_bf16.h>

#include "psvae_common.h"
#include "psvae_device.h"

namespace psvae {

// bf16 copies for the bf16 training pass: the elementwise kernels below can emit their result as a row-major bf16 copy
// too, so that no separate cast launch is needed
__device__ __forceinline__ unsigned short f2bf_ew(float f) {
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<unsigned short*>(&h);
}
struct Bf16Out {
    unsigned short* rm[2];   // row-major [B, Z] outputs (nullable)
};
__device__ __forceinline__ void emit_bf16(const Bf16Out& o, int which, int64_t i, float v) {
    if (o.rm[which] == nullptr) return;
    o.rm[which][i] = f2bf_ew(v);
}

__global__ __launch_bounds__(256) void philox_normal_kernel(float* __restrict__ out, int64_t n, uint64_t seed,
                                                            uint64_t offset) {
    const int64_t nblk = (n + 3) / 4;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = philox_normal4_exact(offset + (uint64_t)b, seed);
        const int64_t e = 4 * b;
        if (e + 3 < n && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
            *reinterpret_cast<f32x4*>(out + e) = v;
        } else {
            for (int i = 0; i < 4; ++i)
                if (e + i < n) out[e + i] = v[i];
        }
    }
}

// the same blocks through Box-Muller on the hardware transcendentals (philox_normal4_hw): what the Langevin chains and the
// fused decode kernels draw inside their loops
__global__ __launch_bounds__(256) void philox_normal_fast_kernel(float* __restrict__ out, int64_t n, uint64_t seed,
                                                                 uint64_t offset) {
    const int64_t nblk = (n + 3) / 4;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = philox_normal4_hw(offset + (uint64_t)b, seed);
        const int64_t e = 4 * b;
        if (e + 3 < n && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
            *reinterpret_cast<f32x4*>(out + e) = v;
        } else {
            for (int i = 0; i < 4; ++i)
                if (e + i < n) out[e + i] = v[i];
        }
    }
}

static inline int grid_for(int64_t work_items, int per_block, int cap = 2048) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

hipError_t launch_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(philox_normal_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, s, out, n, seed, offset);
    return hipGetLastError();
}

hipError_t launch_philox_normal_fast(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(philox_normal_fast_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, s, out, n, seed, offset);
    return hipGetLastError();
}

// fused-step reparameterisation: z = mu + exp(0.5 lv) * eps, plus one KL partial per block
//   kl_partials[blk] = sum over the block's elements of (1 + lv - mu^2 - e^lv)   (lightning.py:115-117)
// grid is FIXED to reparam_blocks(n) (each block owns a contiguous slice) so the sum order is fixed.
constexpr int REPARAM_ELEMS_PER_BLOCK = 1024;  // 256 lanes x one Philox block of 4
int reparam_blocks(int64_t n) { return (int)((n + REPARAM_ELEMS_PER_BLOCK - 1) / REPARAM_ELEMS_PER_BLOCK); }

__global__ __launch_bounds__(256) void reparam_kl_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                         const float* __restrict__ eps_in,
                                                         const uint64_t* __restrict__ rng, uint64_t seed,
                                                         uint64_t offset, float* __restrict__ eps_out,
                                                         float* __restrict__ z, float* __restrict__ kl_partials,
                                                         int64_t n, int vec_ok, const Bf16Out bo) {
    __shared__ float sh4[4];
    if (rng) { seed = rng[0]; offset = rng[1]; }
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t e = 4 * b;
    float kl = 0.f;
    if (e < n) {
        if (vec_ok && e + 3 < n) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(mu + e);
            const f32x4 l = *reinterpret_cast<const f32x4*>(lv + e);
            const f32x4 ep = eps_in ? *reinterpret_cast<const f32x4*>(eps_in + e) : philox_normal4(offset + (uint64_t)b, seed);
            f32x4 zz;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                zz[i] = m[i] + expf(0.5f * l[i]) * ep[i];
                kl += 1.0f + l[i] - m[i] * m[i] - expf(l[i]);
            }
            *reinterpret_cast<f32x4*>(eps_out + e) = ep;
            *reinterpret_cast<f32x4*>(z + e) = zz;
#pragma unroll
            for (int i = 0; i < 4; ++i) emit_bf16(bo, 0, e + i, zz[i]);
        } else {
            f32x4 ep = {0.f, 0.f, 0.f, 0.f};
            if (!eps_in) ep = philox_normal4(offset + (uint64_t)b, seed);
            for (int i = 0; i < 4; ++i) {
                if (e + i < n) {
                    const float ee = eps_in ? eps_in[e + i] : ep[i];
                    const float m = mu[e + i], l = lv[e + i];
                    eps_out[e + i] = ee;
                    const float zv = m + expf(0.5f * l) * ee;
                    z[e + i] = zv;
                    emit_bf16(bo, 0, e + i, zv);
                    kl += 1.0f + l - m * m - expf(l);
                }
            }
        }
    }
    const float s = block_sum_256(kl, sh4);
    if (threadIdx.x == 0) kl_partials[blockIdx.x] = s;
}

// z = mu + exp(0.5 lv) * eps     (model.py:56-57)
__global__ __launch_bounds__(256) void reparam_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                      const float* __restrict__ eps_in, uint64_t seed, uint64_t offset,
                                                      const uint64_t* __restrict__ rng, float* __restrict__ eps_out,
                                                      float* __restrict__ z, int64_t n, int vec_ok) {
    const int64_t nblk = (n + 3) / 4;
    if (rng) { seed = rng[0]; offset = rng[1]; }
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = 4 * b;
        if (vec_ok && e + 3 < n) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(mu + e);
            const f32x4 l = *reinterpret_cast<const f32x4*>(lv + e);
            const f32x4 ep = eps_in ? *reinterpret_cast<const f32x4*>(eps_in + e) : philox_normal4(offset + (uint64_t)b, seed);
            f32x4 zz;
#pragma unroll
            for (int i = 0; i < 4; ++i) zz[i] = m[i] + expf(0.5f * l[i]) * ep[i];
            *reinterpret_cast<f32x4*>(eps_out + e) = ep;
            *reinterpret_cast<f32x4*>(z + e) = zz;
        } else {
            f32x4 ep = {0.f, 0.f, 0.f, 0.f};
            if (!eps_in) ep = philox_normal4(offset + (uint64_t)b, seed);
            for (int i = 0; i < 4; ++i) {
                if (e + i < n) {
                    const float ee = eps_in ? eps_in[e + i] : ep[i];
                    eps_out[e + i] = ee;
                    z[e + i] = mu[e + i] + expf(0.5f * lv[e + i]) * ee;
                }
            }
        }
    }
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__global__ void rng_advance_kernel(uint64_t* rng, uint64_t nblk) { rng[1] += nblk; }

hipError_t launch_reparam(const float* mu, const float* lv, const float* eps_in, uint64_t seed, uint64_t offset,
                          uint64_t* rng, float* eps_out, float* z, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int vec_ok = al16(mu) && al16(lv) && al16(eps_out) && al16(z) && (!eps_in || al16(eps_in));
    if (eps_in) rng = nullptr;
    hipLaunchKernelGGL(reparam_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, s, mu, lv, eps_in, seed, offset,
                       rng, eps_out, z, n, vec_ok);
    if (rng) hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, s, rng, (uint64_t)((n + 3) / 4));
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// row L2 normalisation (model.py:60-61): one wavefront per row
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* y, float* xh,  // may alias (in place)
                                                             float* __restrict__ ynorm, int B, int D) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < B; row += gridDim.x * wpb) {
        const float* yr = y + (size_t)row * D;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += yr[d] * yr[d];
        s = wave_sum(s);
        const float n = fmaxf(sqrtf(s), 1e-12f);
        for (int d = lane; d < D; d += 64) xh[(size_t)row * D + d] = yr[d] / n;
        if (lane == 0) ynorm[row] = n;
    }
}

hipError_t launch_normalize_rows(const float* y, float* xh, float* ynorm, int B, int D, hipStream_t s) {
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(grid_for(B, 4)), dim3(256), 0, s, y, xh, ynorm, B, D);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// ELBO loss (lightning.py:110-124): stage 1 = per-block partial sums {recon, kl}, stage 2 = one block
// ---------------------------------------------------------------------------------------------
constexpr int ELBO_MAX_BLOCKS = 1024;
int elbo_partial_blocks(int B) { return grid_for(B, 4, ELBO_MAX_BLOCKS); }

__global__ __launch_bounds__(256) void elbo_partial_kernel(const float* __restrict__ x, const float* __restrict__ xh,
                                                           const float* __restrict__ mu, const float* __restrict__ lv,
                                                           int B, int D, int Z, int cos_loss,
                                                           float* __restrict__ partials) {
    __shared__ float sh[4][2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float recon = 0.f, kl = 0.f;  // lane-0 accumulators of this wave
    for (int row = blockIdx.x * 4 + w; row < B; row += gridDim.x * 4) {
        const float* xr = x + (size_t)row * D;
        const float* hr = xh + (size_t)row * D;
        if (cos_loss) {
            float a = 0.f, b = 0.f, c = 0.f;
            for (int d = lane; d < D; d += 64) {
                const float h = hr[d], xv = xr[d];
                a += h * h; b += xv * xv; c += h * xv;
            }
            a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
            recon += 1.0f - c / sqrtf((a + 1e-12f) * (b + 1e-12f));
        } else {
            float s = 0.f;
            for (int d = lane; d < D; d += 64) {
                const float df = hr[d] - xr[d];
                s += df * df;
            }
            recon += wave_sum(s);
        }
        float k = 0.f;
        for (int j = lane; j < Z; j += 64) {
            const float m = mu[(size_t)row * Z + j], l = lv[(size_t)row * Z + j];
            k += 1.0f + l - m * m - expf(l);
        }
        kl += wave_sum(k);
    }
    if (lane == 0) { sh[w][0] = recon; sh[w][1] = kl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x + 0] = (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]);
        partials[2 * blockIdx.x + 1] = (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]);
    }
}

__global__ __launch_bounds__(256) void elbo_final_kernel(const float* __restrict__ partials, int nblk, int B, int D,
                                                         float beta, int cos_loss, float* __restrict__ loss3) {
    __shared__ float sh[4][2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float r = 0.f, k = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) { r += partials[2 * i]; k += partials[2 * i + 1]; }
    r = wave_sum(r); k = wave_sum(k);
    if (lane == 0) { sh[w][0] = r; sh[w][1] = k; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float rs = (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]);
        const float ks = (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]);
        const float recon = cos_loss ? rs / (float)B : rs / ((float)B * (float)D) / 10.0f;
        const float kl = -0.5f * ks / (float)B;
        loss3[0] = recon + beta * kl;
        loss3[1] = recon;
        loss3[2] = kl;
    }
}

hipError_t launch_elbo_loss(const float* x, const float* xh, const float* mu, const float* lv, int B, int D, int Z,
                            float beta, bool cos_loss, float* partials, float* loss3, hipStream_t s) {
    const int nblk = elbo_partial_blocks(B);
    hipLaunchKernelGGL(elbo_partial_kernel, dim3(nblk), dim3(256), 0, s, x, xh, mu, lv, B, D, Z, (int)cos_loss, partials);
    hipLaunchKernelGGL(elbo_final_kernel, dim3(1), dim3(256), 0, s, partials, nblk, B, D, beta, (int)cos_loss, loss3);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// gradient w.r.t. the decoder's last Linear output y (one wavefront per row):
//   g  = [g_xhat upstream] + go * d recon / d x_hat       (MSE/10: 2(xh-x)/(10 B D);  cos: see oracle)
//   g_y = g                                              without normalize_decoder
//       = (g - xh <g, xh>) / ynorm   (or g / ynorm when the norm was clamped)   with it
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void recon_grad_kernel(const float* __restrict__ x, const float* __restrict__ xh,
                                                         const float* __restrict__ ynorm,
                                                         const float* __restrict__ g_up, int elbo, int cos_loss,
                                                         int normalize, const float* __restrict__ go_p,
                                                         float* __restrict__ g_y, int B, int D) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const float go = (elbo && go_p) ? *go_p : 1.0f;
    const float cmse = go * 2.0f / (10.0f * (float)B * (float)D);
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < B; row += gridDim.x * wpb) {
        const size_t o = (size_t)row * D;
        float a = 1.f, den = 1.f, cosv = 0.f;
        if (elbo && cos_loss) {
            float sa = 0.f, sb = 0.f, sc = 0.f;
            for (int d = lane; d < D; d += 64) {
                const float h = xh[o + d], xv = x[o + d];
                sa += h * h; sb += xv * xv; sc += h * xv;
            }
            sa = wave_sum(sa); sb = wave_sum(sb); sc = wave_sum(sc);
            a = sa + 1e-12f;
            den = sqrtf(a * (sb + 1e-12f));
            cosv = sc / den;
        }
        auto gval = [&](int d) -> float {
            float g = g_up ? g_up[o + d] : 0.f;
            if (elbo) {
                const float h = xh[o + d], xv = x[o + d];
                g += cos_loss ? -(go / (float)B) * (xv / den - cosv * h / a) : cmse * (h - xv);
            }
            return g;
        };
        if (!normalize) {
            for (int d = lane; d < D; d += 64) g_y[o + d] = gval(d);
        } else {
            float dot = 0.f;
            for (int d = lane; d < D; d += 64) dot += gval(d) * xh[o + d];
            dot = wave_sum(dot);
            const float n = ynorm[row];
            const bool clamped = n <= 1e-12f;
            for (int d = lane; d < D; d += 64) {
                const float g = gval(d);
                g_y[o + d] = clamped ? g / n : (g - xh[o + d] * dot) / n;
            }
        }
    }
}

hipError_t launch_recon_grad(const float* x, const float* xh, const float* ynorm, const float* g_xhat, bool elbo,
                             bool cos_loss, bool normalize, const float* go, float* g_y, int B, int D, hipStream_t s) {
    hipLaunchKernelGGL(recon_grad_kernel, dim3(grid_for(B, 4)), dim3(256), 0, s, x, xh, ynorm, g_xhat, (int)elbo,
                       (int)cos_loss, (int)normalize, go, g_y, B, D);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// reparameterisation backward + KL gradient
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reparam_bwd_kernel(const float* __restrict__ g_z, const float* __restrict__ mu,
                                                          const float* __restrict__ lv, const float* __restrict__ eps,
                                                          const float* __restrict__ g_mu_up,
                                                          const float* __restrict__ g_lv_up, int elbo, float beta,
                                                          const float* __restrict__ go_p, float* __restrict__ g_mu,
                                                          float* __restrict__ g_lv, int64_t n, int B) {
    const float go = (elbo && go_p) ? *go_p : 1.0f;
    const float kb = elbo ? go * beta / (float)B : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gz = g_z[i], l = lv[i];
        float gm = gz, gl = gz * eps[i] * 0.5f * expf(0.5f * l);
        if (g_mu_up) gm += g_mu_up[i];
        if (g_lv_up) gl += g_lv_up[i];
        if (elbo) {
            gm += kb * mu[i];
            gl += 0.5f * kb * (expf(l) - 1.0f);
        }
        g_mu[i] = gm;
        g_lv[i] = gl;
    }
}

// d KL / d(mu, logvar) alone, scaled by a device scalar:  g_mu = g mu / B,  g_lv = g (e^lv - 1) / (2B)   (lightning.py:115-117)
__global__ __launch_bounds__(256) void kl_grad_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                      const float* __restrict__ g_p, float* __restrict__ g_mu,
                                                      float* __restrict__ g_lv, int64_t n, int B) {
    const float kb = (g_p ? *g_p : 1.0f) / (float)B;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        g_mu[i] = kb * mu[i];
        g_lv[i] = 0.5f * kb * (expf(lv[i]) - 1.0f);
    }
}
hipError_t launch_kl_grad(const float* mu, const float* lv, const float* g, float* g_mu, float* g_lv, int B, int Z,
                          hipStream_t s) {
    const int64_t n = (int64_t)B * Z;
    hipLaunchKernelGGL(kl_grad_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, mu, lv, g, g_mu, g_lv, n, B);
    return hipGetLastError();
}

// fused-step variant: elementwise part as above (ELBO gradient always on); block 0 then finishes the
// loss from the partial sums and advances the Philox state (every reader of rng[1] ran launches ago)
__global__ __launch_bounds__(256) void reparam_bwd_final_kernel(
    const float* __restrict__ g_z, const float* __restrict__ mu, const float* __restrict__ lv,
    const float* __restrict__ eps, float beta, const float* __restrict__ go_p, float* __restrict__ g_mu,
    float* __restrict__ g_lv, int64_t n, int B, int D, const float* __restrict__ kl_partials, int n_kl,
    const float* __restrict__ recon_partials, int n_recon, float* __restrict__ loss3, uint64_t* rng, uint64_t nblk,
    const Bf16Out bo) {
    __shared__ float sh4[4];
    const float go = go_p ? *go_p : 1.0f;
    const float kb = go * beta / (float)B;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gz = g_z[i], l = lv[i];
        const float gm = gz + kb * mu[i];
        const float gl = gz * eps[i] * 0.5f * expf(0.5f * l) + 0.5f * kb * (expf(l) - 1.0f);
        g_mu[i] = gm;
        g_lv[i] = gl;
        emit_bf16(bo, 0, i, gm);
        emit_bf16(bo, 1, i, gl);
    }
    if (blockIdx.x == 0) {
        float k = 0.f, r = 0.f;
        for (int i = threadIdx.x; i < n_kl; i += 256) k += kl_partials[i];
        for (int i = threadIdx.x; i < n_recon; i += 256) r += recon_partials[i];
        const float ks = block_sum_256(k, sh4);
        __syncthreads();
        const float rs = block_sum_256(r, sh4);
        if (threadIdx.x == 0) {
            const float recon = rs / ((float)B * (float)D) / 10.0f;
            const float kl = -0.5f * ks / (float)B;
            loss3[0] = recon + beta * kl;
            loss3[1] = recon;
            loss3[2] = kl;
            if (rng) rng[1] += nblk;
        }
    }
}

hipError_t launch_reparam_kl(const float* mu, const float* lv, const float* eps_in, const uint64_t* rng, uint64_t seed,
                             uint64_t offset, float* eps_out, float* z, float* kl_partials, int64_t n, hipStream_t s,
                             unsigned short* zb) {
    const int vec_ok = al16(mu) && al16(lv) && al16(eps_out) && al16(z) && (!eps_in || al16(eps_in));
    if (eps_in) rng = nullptr;
    Bf16Out bo{};
    bo.rm[0] = zb;
    hipLaunchKernelGGL(reparam_kl_kernel, dim3(reparam_blocks(n)), dim3(256), 0, s, mu, lv, eps_in, rng, seed, offset,
                       eps_out, z, kl_partials, n, vec_ok, bo);
    return hipGetLastError();
}

hipError_t launch_reparam_bwd_final(const float* g_z, const float* mu, const float* lv, const float* eps, float beta,
                                    const float* go, float* g_mu, float* g_lv, int B, int Z, int D,
                                    const float* kl_partials, int n_kl, const float* recon_partials, int n_recon,
                                    float* loss3, uint64_t* rng, hipStream_t s, unsigned short* const* bf_rm) {
    const int64_t n = (int64_t)B * Z;
    Bf16Out bo{};
    if (bf_rm) { bo.rm[0] = bf_rm[0]; bo.rm[1] = bf_rm[1]; }
    hipLaunchKernelGGL(reparam_bwd_final_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, g_z, mu, lv, eps, beta, go,
                       g_mu, g_lv, n, B, D, kl_partials, n_kl, recon_partials, n_recon, loss3, rng,
                       (uint64_t)((n + 3) / 4), bo);
    return hipGetLastError();
}

hipError_t launch_reparam_bwd(const float* g_z, const float* mu, const float* lv, const float* eps,
                              const float* g_mu_up, const float* g_lv_up, bool elbo, float beta, const float* go,
                              float* g_mu, float* g_lv, int B, int Z, hipStream_t s) {
    const int64_t n = (int64_t)B * Z;
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, g_z, mu, lv, eps, g_mu_up, g_lv_up,
                       (int)elbo, beta, go, g_mu, g_lv, n, B);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// fused multi-tensor Adam on a flat buffer (lightning.py:205; torch.optim.Adam, non-amsgrad)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float step_size, float inv_sqrt_bc2, float b1, float omb1, float b2,
                                                   float omb2, float eps, float wd, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = b1 * m[i] + omb1 * gi;          // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = b2 * v[i] + omb2 * gi * gi;     // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, int64_t step, double lr, double b1,
                       double b2, double eps, double wd, double gscale, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p, g, m, v, n, (float)(lr / bc1),
                       (float)(1.0 / sqrt(bc2)), (float)b1, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps,
                       (float)wd, (float)gscale);
    return hipGetLastError();
}

// the same update over a LIST of tensors in one launch (torch.optim.Adam's foreach form is one launch per operation and list:
// ~190 us of host time per step for the 18 tensors of the model): block b serves chunk b of 4,096 elements; the chunk
// table lives in the kernel arguments
struct AdamMultiArgs {
    float* p[PSVAE_ADAM_MAX_TENSORS];
    const float* g[PSVAE_ADAM_MAX_TENSORS];
    float* m[PSVAE_ADAM_MAX_TENSORS];
    float* v[PSVAE_ADAM_MAX_TENSORS];
    long long n[PSVAE_ADAM_MAX_TENSORS];
    int first_chunk[PSVAE_ADAM_MAX_TENSORS + 1];
    int nt;
};
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamMultiArgs a, float step_size, float inv_sqrt_bc2, float b1,
                                                         float omb1, float b2, float omb2, float eps, float wd, float gscale) {
    int t = 0;
    for (int i = 1; i < a.nt; ++i)
        if ((int)blockIdx.x >= a.first_chunk[i]) t = i;
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const long long n = a.n[t];
    const long long beg = (long long)((int)blockIdx.x - a.first_chunk[t]) * 4096;
    const long long end = beg + 4096 < n ? beg + 4096 : n;
    for (long long i = beg + threadIdx.x; i < end; i += 256) {
        float gi = g[i] * gscale;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = b1 * m[i] + omb1 * gi;
        const float vi = b2 * v[i] + omb2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

hipError_t launch_adam_multi(int nt, float* const* p, const float* const* g, float* const* m, float* const* v,
                             const int64_t* numel, int64_t step, double lr, double b1, double b2, double eps, double wd,
                             double gscale, hipStream_t s) {
    if (nt <= 0) return hipSuccess;
    if (nt > PSVAE_ADAM_MAX_TENSORS) return hipErrorInvalidValue;
    AdamMultiArgs a{};
    int chunks = 0;
    for (int i = 0; i < nt; ++i) {
        a.p[i] = p[i]; a.g[i] = g[i]; a.m[i] = m[i]; a.v[i] = v[i]; a.n[i] = numel[i];
        a.first_chunk[i] = chunks;
        chunks += (int)((numel[i] + 4095) / 4096);
    }
    a.first_chunk[nt] = chunks;
    a.nt = nt;
    if (chunks == 0) return hipSuccess;
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(chunks), dim3(256), 0, s, a, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)),
                       (float)b1, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps, (float)wd, (float)gscale);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// conditional-table gradients (cfg3 extension): gtab[c][n] = sum over rows b with label c of dY[b][n].
// grid.x = column blocks of 64, grid.y = problem; 256 threads = 64 columns x 4 row lanes.
// ---------------------------------------------------------------------------------------------
struct CondSumArgs {
    const float* dY[3];
    float* gtab[3];
    int n;
};

__global__ __launch_bounds__(256) void cond_colsum_kernel(CondSumArgs a, const int32_t* __restrict__ gender,
                                                          const int32_t* __restrict__ age, int B, int N) {
    __shared__ float sh[4][6][64];
    const float* __restrict__ dY = a.dY[blockIdx.y];
    float* __restrict__ gt = a.gtab[blockIdx.y];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        for (int b = rl; b < B; b += 4) {
            const float v = dY[(size_t)b * N + col];
            const int g = gender[b], ag = age[b];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[c] += (g == c) ? v : 0.f;
                acc[3 + c] += (ag == c) ? v : 0.f;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) sh[rl][c][cl] = acc[c];
    __syncthreads();
    if (rl == 0 && col < N) {
#pragma unroll
        for (int c = 0; c < 6; ++c) gt[(size_t)c * N + col] = (sh[0][c][cl] + sh[1][c][cl]) + (sh[2][c][cl] + sh[3][c][cl]);
    }
}

hipError_t launch_cond_colsum(const float* const* dY, float* const* gtab, int nprob, const int32_t* gender,
                              const int32_t* age, int B, int N, hipStream_t s) {
    CondSumArgs a{};
    for (int i = 0; i < nprob; ++i) { a.dY[i] = dY[i]; a.gtab[i] = gtab[i]; }
    a.n = nprob;
    hipLaunchKernelGGL(cond_colsum_kernel, dim3((N + 63) / 64, nprob), dim3(256), 0, s, a, gender, age, B, N);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// cross-workgroup split-K: sum the partial slabs of up to 18 gradient tensors in fixed order
// ---------------------------------------------------------------------------------------------
struct SlabReduceArgs {
    const float* src[24];
    float* dst[24];
    long long numel[24];
    int S[24];
    int n;
};
__global__ __launch_bounds__(256) void slab_reduce_kernel(const SlabReduceArgs a) {
    const int i = blockIdx.y;
    const long long n = a.numel[i];
    const float* __restrict__ src = a.src[i];
    float* __restrict__ dst = a.dst[i];
    const int S = a.S[i];
    const bool v4 = ((n & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    if (v4) {
        for (long long e = 4 * ((long long)blockIdx.x * 256 + threadIdx.x); e < n; e += 4LL * gridDim.x * 256) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(src + e);
            for (int s = 1; s < S; ++s) acc += *reinterpret_cast<const f32x4*>(src + (size_t)s * n + e);
            *reinterpret_cast<f32x4*>(dst + e) = acc;
        }
    } else {
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
            float acc = src[e];
            for (int s = 1; s < S; ++s) acc += src[(size_t)s * n + e];
            dst[e] = acc;
        }
    }
}
hipError_t launch_slab_reduce(const float* const* src, float* const* dst, const int64_t* numel, const int* S, int n,
                              hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n > 24) return hipErrorInvalidValue;
    SlabReduceArgs a{};
    long long mx = 0;
    for (int i = 0; i < n; ++i) {
        a.src[i] = src[i]; a.dst[i] = dst[i]; a.numel[i] = numel[i]; a.S[i] = S[i];
        mx = numel[i] > mx ? numel[i] : mx;
    }
    a.n = n;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid_for((mx + 3) / 4, 256, 256), n), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// grouped fp32 transposes: dst[i] [C,R] = src[i] [R,C]^T (32x32 LDS tiles, whole 128-B rows both ways).
// The forward wants the weights k-MAJOR ([in,out]): a k-contiguous weight operand is read in 64-B pieces
// (half of every 128-B line), which halves the per-CU feed rate of the register-direct GEMM.
// ---------------------------------------------------------------------------------------------
struct TransposeArgs {
    const float* src[9];
    float* dst[9];
    int R[9], Cc[9], tile_begin[9];
    int n;
};
__global__ __launch_bounds__(256) void transpose_f32_kernel(const TransposeArgs a) {
    __shared__ float tile[32][33];
    int pi = 0;
    for (int i = 1; i < a.n; ++i)
        if ((int)blockIdx.x >= a.tile_begin[i]) pi = i;
    const int R = a.R[pi], Cc = a.Cc[pi];
    const int tiles_c = (Cc + 31) / 32;
    const int lt = blockIdx.x - a.tile_begin[pi];
    const int r0 = (lt / tiles_c) * 32, c0 = (lt % tiles_c) * 32;
    const float* __restrict__ src = a.src[pi];
    float* __restrict__ dst = a.dst[pi];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = r0 + ty + 8 * i, cc = c0 + tx;
        tile[ty + 8 * i][tx] = (rr < R && cc < Cc) ? src[(size_t)rr * Cc + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = c0 + ty + 8 * i, rr = r0 + tx;
        if (cc < Cc && rr < R) dst[(size_t)cc * R + rr] = tile[tx][ty + 8 * i];
    }
}
hipError_t launch_transpose_f32(const float* const* src, float* const* dst, const int* R, const int* Cc, int n,
                                hipStream_t s) {
    TransposeArgs a{};
    int t = 0;
    for (int i = 0; i < n; ++i) {
        a.src[i] = src[i]; a.dst[i] = dst[i]; a.R[i] = R[i]; a.Cc[i] = Cc[i];
        a.tile_begin[i] = t;
        t += ((R[i] + 31) / 32) * ((Cc[i] + 31) / 32);
    }
    a.n = n;
    hipLaunchKernelGGL(transpose_f32_kernel, dim3(t), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// packed embedding table -> batch (replaces one torch.load per sample, data/cv.py:58-76):
// out[b, :] = table[idx[b], :]; whole rows, 16 B per lane, one wavefront per row; an out-of-range
// index yields a zero row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, int64_t N, int D,
                                                          const int64_t* __restrict__ idx, int64_t B,
                                                          float* __restrict__ out, int vec_ok) {
    const int lane = threadIdx.x & 63;
    const int64_t wpb = blockDim.x >> 6;
    for (int64_t b = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); b < B; b += (int64_t)gridDim.x * wpb) {
        const int64_t r = idx[b];
        const bool ok = r >= 0 && r < N;
        const float* __restrict__ src = table + (size_t)(ok ? r : 0) * D;
        float* __restrict__ dst = out + (size_t)b * D;
        if (vec_ok) {
            for (int d = 4 * lane; d < D; d += 256) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4*>(src + d);
                *reinterpret_cast<f32x4*>(dst + d) = v;
            }
        } else {
            for (int d = lane; d < D; d += 64) dst[d] = ok ? src[d] : 0.f;
        }
    }
}

hipError_t launch_gather_rows(const float* table, int64_t N, int D, const int64_t* idx, int64_t B, float* out,
                              hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const int vec_ok = ((D & 3) == 0) && al16(table) && al16(out);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(B, 4)), dim3(256), 0, s, table, N, D, idx, B, out, vec_ok);
    return hipGetLastError();
}

}  // namespace psvae
