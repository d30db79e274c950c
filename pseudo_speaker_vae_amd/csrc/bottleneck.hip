// The skinny middle of the VAE training pass as two dedicated row-local kernels (gfx950), plus the row kernel of the
// cosine / normalised loss variants and the dual-layout Adam step.
//
//   forward  (model.py:19,27,56-58; lightning.py:115-117):   mu, logvar  ->  eps, z, KL partials  ->  d1 = relu(z Wd1^T + b)
//   backward (autograd of the same lines):  g_z  ->  g_mu, g_logvar  ->  dh2m, dh2s (ReLU-masked)
//
// Both layers next to the bottleneck have K or N = Z (64): as GEMM launches of their own they sat on the launch floor.
// In the fused step the heads' products (h2 W3^T) and decoder.0's dgrad (dd1 Wd1) arrive as PARTIAL products, one slab
// per 64-column tile of the producing GEMM launch (GemmDesc::B2, gemm_f32.hip): the kernels here sum the S slabs in
// fixed order (bitwise reproducible), so neither product needs a launch of its own.  A workgroup owns 16 rows x 128
// output columns; every workgroup re-derives its 16 x Z slice of the bottleneck values (z forward, g_mu / g_logvar
// backward), keeps it in LDS as the MFMA A operand and multiplies it with weight fragments loaded straight from global.
#include "psvae_common.h"
#include "psvae_device.h"

namespace psvae {

static inline bool al16h(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// explicit address-space-1 loads: pointers that arrive inside a by-value struct otherwise compile to FLAT loads, which
// return out of order and force every wait to zero (gemm_f32.hip has the long version of this note)
#define PSVAE_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ f32x4 gld4(const float* p) {
    return *reinterpret_cast<const PSVAE_GLOBAL f32x4*>(reinterpret_cast<uintptr_t>(p));
}

template <int MAXC>  // Z <= 16 * MAXC
__global__ __launch_bounds__(256) void bottleneck_fwd_kernel(const BnFwdArgs a) {
    constexpr int ZS = 16 * MAXC + 4;  // LDS row stride of the z tile
    constexpr int OS = 128 + 4;        // ... of the output tile
    __shared__ __attribute__((aligned(16))) float zs[16 * ZS];
    __shared__ __attribute__((aligned(16))) float os[16 * OS];
    __shared__ float sh4[4];
    const int B = a.B, Z = a.Z, N = a.N;
    const int tiles_n = (N + 127) / 128;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 16, n0 = tn * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    // Every global read of this kernel is issued up front and consumed later (loads return in order: the kernel then
    // pays ONE memory round trip, not one per slab / per output row -- the launch computes almost nothing, its length
    // is its dependent round trips: 8.9 -> see DESIGN 3.2).  The Philox state is read by a vector load for the same reason.
    uint64_t seed = a.seed, offset = a.offset;
    unsigned long long rng_s = 0, rng_o = 0;
    if (a.rng) {
        rng_s = *reinterpret_cast<const PSVAE_GLOBAL unsigned long long*>(reinterpret_cast<uintptr_t>(a.rng));
        rng_o = *reinterpret_cast<const PSVAE_GLOBAL unsigned long long*>(reinterpret_cast<uintptr_t>(a.rng + 1));
    }
    // ---- 0. this wave's weight fragments: issued first, they land while z is being made ----
    const int nc = Z >> 4;
    f32x4 wf[MAXC][2];
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            int n = n0 + 32 * wave + 16 * b + r;
            n = n < N ? n : N - 1;  // clamped columns are never stored
            // (k chunks past Z: an unconditional load from a clamped address -- a branch around the load would put a
            //  wait behind it -- and the MFMA loop below skips them)
            wf[c][b] = gld4(a.W + (size_t)n * Z + 16 * (c < nc ? c : nc - 1) + 4 * q);
        }
    // (the output pass's bias and table rows: requested now too)
    f32x4 ob[2], og[2], oa[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = tid + 256 * it, row = i >> 5, col = n0 + 4 * (i & 31);
        const bool live = m0 + row < B && col < N;
        ob[it] = live ? gld4(a.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};  // N % 4 == 0
        og[it] = oa[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.ctab && live) {
            const int g = a.gender[m0 + row], ag = a.age[m0 + row];
            if (g >= 0 && g < 3) og[it] = gld4(a.ctab + (size_t)g * N + col);
            if (ag >= 0 && ag < 3) oa[it] = gld4(a.ctab + (size_t)(3 + ag) * N + col);
        }
    }
    if (a.rng) { seed = rng_s; offset = rng_o; }
    // ---- 1. the 16 x Z slice of mu, logvar, z (float4 = one Philox block), KL terms ----
    float kl = 0.f;
    const int zq = Z >> 2;  // float4 per row
    const size_t slab_stride = (size_t)B * Z;
    for (int i = tid; i < 16 * zq; i += 256) {
        const int row = i / zq, col = 4 * (i - row * zq);
        f32x4 zz = {0.f, 0.f, 0.f, 0.f};
        if (m0 + row < B) {
            const int64_t e = (int64_t)(m0 + row) * Z + col;
            f32x4 m, l;
            if (a.S == 8) {  // H / 64 column tiles of the producing launch: all 16 slab reads in flight at once
                f32x4 pm[8], pl[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    pm[s] = gld4(a.slab_m + s * slab_stride + e);
                    pl[s] = gld4(a.slab_s + s * slab_stride + e);
                }
                m = gld4(a.b3m + col);
                l = gld4(a.b3s + col);
#pragma unroll
                for (int s = 0; s < 8; ++s) {  // the same fixed order as the general loop below: bitwise reproducible
                    m += pm[s];
                    l += pl[s];
                }
                if (tn == 0) {
                    *reinterpret_cast<f32x4*>(a.mu + e) = m;
                    *reinterpret_cast<f32x4*>(a.lv + e) = l;
                }
            } else if (a.S > 0) {
                m = *reinterpret_cast<const f32x4*>(a.b3m + col);
                l = *reinterpret_cast<const f32x4*>(a.b3s + col);
                for (int s = 0; s < a.S; ++s) {  // fixed order: bitwise reproducible
                    m += *reinterpret_cast<const f32x4*>(a.slab_m + s * slab_stride + e);
                    l += *reinterpret_cast<const f32x4*>(a.slab_s + s * slab_stride + e);
                }
                if (tn == 0) {
                    *reinterpret_cast<f32x4*>(a.mu + e) = m;
                    *reinterpret_cast<f32x4*>(a.lv + e) = l;
                }
            } else {
                m = *reinterpret_cast<const f32x4*>(a.mu + e);
                l = *reinterpret_cast<const f32x4*>(a.lv + e);
            }
            const f32x4 ep = a.eps_in ? *reinterpret_cast<const f32x4*>(a.eps_in + e) : philox_normal4(offset + (uint64_t)(e >> 2), seed);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                zz[t] = m[t] + expf(0.5f * l[t]) * ep[t];
                kl += 1.0f + l[t] - m[t] * m[t] - expf(l[t]);
            }
            if (tn == 0) {
                *reinterpret_cast<f32x4*>(a.eps_out + e) = ep;
                *reinterpret_cast<f32x4*>(a.z_out + e) = zz;
            }
        }
        *reinterpret_cast<f32x4*>(zs + row * ZS + col) = zz;
    }
    const float kls = block_sum_256(kl, sh4);  // (contains the barrier that publishes zs)
    if (tn == 0 && tid == 0) a.kl_partials[tm] = kls;
    // ---- 2. 16 x 128 tile of z Wd1^T: wave w owns columns n0 + 32 w .. + 31 ----
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nc) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(zs + r * ZS + 16 * c + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], wf[c][b][t], acc[b], 0, 0, 0);
        }
    }
    // ---- 3. whole rows leave through LDS: + bias (+ conditional table rows), ReLU ----
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) os[(4 * q + j) * OS + 32 * wave + 16 * b + r] = acc[b][j];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = tid + 256 * it;
        const int row = i >> 5, c4 = i & 31;
        const int col = n0 + 4 * c4;
        if (m0 + row >= B || col >= N) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(os + row * OS + 4 * c4);
        v += ob[it];
        if (a.ctab) {  // (same order of additions as before: bias, gender row, age row)
            v += og[it];
            v += oa[it];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
        *reinterpret_cast<f32x4*>(a.Y + (size_t)(m0 + row) * N + col) = v;
    }
}

bool reparam_linear_ok(int Z, int N, const void* const* ptrs, int nptr) {
    if ((Z & 15) || Z > 128 || (N & 3)) return false;
    for (int i = 0; i < nptr; ++i)
        if (ptrs[i] && (reinterpret_cast<uintptr_t>(ptrs[i]) & 15)) return false;
    return true;
}
int reparam_linear_tiles(int B) { return (B + 15) / 16; }

hipError_t launch_bottleneck_fwd(const BnFwdArgs& a0, hipStream_t s) {
    BnFwdArgs a = a0;
    if (a.eps_in) a.rng = nullptr;
    const int grid = reparam_linear_tiles(a.B) * ((a.N + 127) / 128);
    if (a.Z <= 64)
        hipLaunchKernelGGL(bottleneck_fwd_kernel<4>, dim3(grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(bottleneck_fwd_kernel<8>, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_reparam_linear(const float* mu, const float* lv, const float* eps_in, const uint64_t* rng,
                                 uint64_t seed, uint64_t offset, const float* W, const float* bias, float* eps_out,
                                 float* z_out, float* Y, float* kl_partials, int B, int Z, int N, hipStream_t s) {
    BnFwdArgs a{};
    a.mu = const_cast<float*>(mu); a.lv = const_cast<float*>(lv); a.S = 0;  // S == 0: mu / lv are only read
    a.eps_in = eps_in; a.rng = rng; a.seed = seed; a.offset = offset; a.W = W; a.bias = bias;
    a.eps_out = eps_out; a.z_out = z_out; a.Y = Y; a.kl_partials = kl_partials; a.B = B; a.Z = Z; a.N = N;
    return launch_bottleneck_fwd(a, s);
}

// ---------------------------------------------------------------------------------------------------------------
// backward mirror: g_z = sum_s slab[s] (partial products dd1 Wd1 of the decoder.2 dgrad launch);
//   g_mu = g_z + kb mu;  g_lv = g_z eps 0.5 exp(0.5 lv) + 0.5 kb (exp(lv) - 1),  kb = go beta / B   (lightning.py:115-121)
//   dh2m = (g_mu W3m) * [h2m > 0];  dh2s = (g_lv W3s) * [h2s > 0]            (encoder_*.4 dgrad, K = Z)
// The workgroup of tile (0, 0) also finishes the loss from the partial sums and advances the Philox state.
// ---------------------------------------------------------------------------------------------------------------
template <int MAXC>
__global__ __launch_bounds__(256) void bottleneck_bwd_kernel(const BnBwdArgs a) {
    constexpr int ZS = 16 * MAXC + 4;
    constexpr int OS = 128 + 4;
    __shared__ __attribute__((aligned(16))) float gs[2][16 * ZS];
    __shared__ __attribute__((aligned(16))) float os[2][16 * OS];
    __shared__ float red[8];
    const int B = a.B, Z = a.Z, N = a.N;
    const int tiles_n = (N + 127) / 128;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 16, n0 = tn * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int nc = Z >> 4;
    // ---- 0. weight fragments of both heads (k contiguous rows of the k-major copies) ----
    f32x4 wf[2][MAXC][2];
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            int n = n0 + 32 * wave + 16 * b + r;
            n = n < N ? n : N - 1;
            const size_t o = (size_t)n * Z + 16 * (c < nc ? c : nc - 1) + 4 * q;  // (clamped, as in the forward kernel)
            wf[0][c][b] = gld4(a.W3mT + o);
            wf[1][c][b] = gld4(a.W3sT + o);
        }
    // (the output pass's ReLU masks do not depend on anything computed here: requested now, consumed at the end -- every
    //  global read of this kernel is in flight before the first is used, see the forward kernel)
    f32x4 hmask[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = tid + 256 * it;
        const int e = i >> 9, row = (i >> 5) & 15, col = n0 + 4 * (i & 31);
        hmask[it] = (m0 + row < B && col < N) ? gld4((e ? a.h2s : a.h2m) + (size_t)(m0 + row) * N + col)
                                              : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- 1. g_z -> g_mu, g_logvar for the 16 x Z slice ----
    const float go_v = a.go ? *reinterpret_cast<const PSVAE_GLOBAL float*>(reinterpret_cast<uintptr_t>(a.go)) : 1.0f;
    const float kb = go_v * a.beta / (float)B;
    const int zq = Z >> 2;
    const size_t slab_stride = (size_t)B * Z;
    for (int i = tid; i < 16 * zq; i += 256) {
        const int row = i / zq, col = 4 * (i - row * zq);
        f32x4 gm = {0.f, 0.f, 0.f, 0.f}, gl = {0.f, 0.f, 0.f, 0.f};
        if (m0 + row < B) {
            const int64_t e = (int64_t)(m0 + row) * Z + col;
            f32x4 gz;
            if (a.S == 8) {  // all eight slab reads in flight at once; summed in the general loop's order
                f32x4 pg[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) pg[s] = gld4(a.slab_gz + s * slab_stride + e);
                gz = pg[0];
#pragma unroll
                for (int s = 1; s < 8; ++s) gz += pg[s];
            } else {
                gz = *reinterpret_cast<const f32x4*>(a.slab_gz + e);
                for (int s = 1; s < a.S; ++s) gz += *reinterpret_cast<const f32x4*>(a.slab_gz + s * slab_stride + e);
            }
            const f32x4 m = gld4(a.mu + e);
            const f32x4 l = gld4(a.lv + e);
            const f32x4 ep = gld4(a.eps + e);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                gm[t] = gz[t] + kb * m[t];
                gl[t] = gz[t] * ep[t] * 0.5f * expf(0.5f * l[t]) + 0.5f * kb * (expf(l[t]) - 1.0f);
            }
            if (tn == 0) {
                *reinterpret_cast<f32x4*>(a.g_mu + e) = gm;
                *reinterpret_cast<f32x4*>(a.g_lv + e) = gl;
            }
        }
        *reinterpret_cast<f32x4*>(gs[0] + row * ZS + col) = gm;
        *reinterpret_cast<f32x4*>(gs[1] + row * ZS + col) = gl;
    }
    __syncthreads();
    // ---- 2. two 16 x 128 tiles: g_mu W3m and g_lv W3s ----
    f32x4 acc[2][2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[e][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nc) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(gs[e] + r * ZS + 16 * c + 4 * q);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[e][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], wf[e][c][b][t], acc[e][b], 0, 0, 0);
            }
        }
    }
    // ---- 3. whole rows leave through LDS with the ReLU masks applied ----
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) os[e][(4 * q + j) * OS + 32 * wave + 16 * b + r] = acc[e][b][j];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = tid + 256 * it;
        const int e = i >> 9, row = (i >> 5) & 15, c4 = i & 31;
        const int col = n0 + 4 * c4;
        if (m0 + row >= B || col >= N) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(os[e] + row * OS + 4 * c4);
        const size_t o = (size_t)(m0 + row) * N + col;
        const f32x4 hm = hmask[it];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = hm[t] > 0.f ? v[t] : 0.f;
        *reinterpret_cast<f32x4*>((e ? a.dh2s : a.dh2m) + o) = v;
    }
    // ---- 4. one workgroup finishes the loss (fixed-order sums) and advances the Philox state ----
    if (blockIdx.x == 0) {
        float kls = 0.f, rs = 0.f;
        for (int i = tid; i < a.n_kl; i += 256) kls += a.kl_partials[i];
        for (int i = tid; i < a.n_recon; i += 256) rs += a.recon_partials[i];
        kls = wave_sum(kls);
        rs = wave_sum(rs);
        if (lane == 0) { red[wave] = kls; red[4 + wave] = rs; }
        __syncthreads();
        if (tid == 0) {
            const float kt = (red[0] + red[1]) + (red[2] + red[3]);
            const float rt = (red[4] + red[5]) + (red[6] + red[7]);
            const float recon = a.recon_mode ? rt / (float)B : rt / ((float)B * (float)a.D) / 10.0f;
            const float klv = -0.5f * kt / (float)B;
            a.loss3[0] = recon + a.beta * klv;
            a.loss3[1] = recon;
            a.loss3[2] = klv;
            if (a.rng) a.rng[1] += a.rng_adv;
        }
    }
}

hipError_t launch_bottleneck_bwd(const BnBwdArgs& a, hipStream_t s) {
    const int grid = reparam_linear_tiles(a.B) * ((a.N + 127) / 128);
    if (a.Z <= 64)
        hipLaunchKernelGGL(bottleneck_bwd_kernel<4>, dim3(grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(bottleneck_bwd_kernel<8>, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Row kernel of the loss variants inside the fused step (one wavefront per row): optional L2 normalisation of the
// decoder output (model.py:60-61), the row's reconstruction-loss term (MSE, lightning.py:113, or cosine, :111), and
// the gradient w.r.t. the decoder's last Linear output -- what normalize_rows + elbo_partial + recon_grad do in the
// separate path, in ONE launch.  partials[block] = sum of the block's 4 rows' loss terms (fixed order).
// ---------------------------------------------------------------------------------------------------------------
int recon_row_blocks(int B) { return (B + 3) / 4; }

__global__ __launch_bounds__(256) void recon_row_kernel(const float* __restrict__ y, const float* __restrict__ x,
                                                        float* __restrict__ xh, float* __restrict__ ynorm,
                                                        float* __restrict__ g_y, float* __restrict__ partials,
                                                        const float* __restrict__ go_p, int normalize, int cos_loss,
                                                        int B, int D, unsigned short* __restrict__ g_yb) {
    __shared__ float sh4[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + w;
    const float go = go_p ? *go_p : 1.0f;
    float term = 0.f;
    if (row < B) {
        const size_t o = (size_t)row * D;
        float n = 1.f;
        if (normalize) {  // xh = y / max(||y||, 1e-12)
            float s = 0.f;
            for (int d = lane; d < D; d += 64) s += y[o + d] * y[o + d];
            s = wave_sum(s);
            n = fmaxf(sqrtf(s), 1e-12f);
            for (int d = lane; d < D; d += 64) xh[o + d] = y[o + d] / n;
            if (lane == 0 && ynorm) ynorm[row] = n;
        }
        // (without normalisation y IS x_hat: the GEMM epilogue wrote it to xh and y == xh)
        const float* __restrict__ hr = normalize ? xh : y;
        float a = 1.f, den = 1.f, cosv = 0.f;
        if (cos_loss) {
            float sa = 0.f, sb = 0.f, sc = 0.f;
            for (int d = lane; d < D; d += 64) {
                const float h = normalize ? y[o + d] / n : hr[o + d], xv = x[o + d];
                sa += h * h; sb += xv * xv; sc += h * xv;
            }
            sa = wave_sum(sa); sb = wave_sum(sb); sc = wave_sum(sc);
            a = sa + 1e-12f;
            den = sqrtf(a * (sb + 1e-12f));
            cosv = sc / den;
            term = 1.0f - cosv;
        } else {
            float s = 0.f;
            for (int d = lane; d < D; d += 64) {
                const float h = normalize ? y[o + d] / n : hr[o + d];
                const float df = h - x[o + d];
                s += df * df;
            }
            term = wave_sum(s);
        }
        const float cmse = go * 2.0f / (10.0f * (float)B * (float)D);
        auto gval = [&](int d) -> float {
            const float h = normalize ? y[o + d] / n : hr[o + d], xv = x[o + d];
            return cos_loss ? -(go / (float)B) * (xv / den - cosv * h / a) : cmse * (h - xv);
        };
        // (bf16 pass: the gradient leaves as bf16, round to nearest even, for the decoder's gradient chain and dW decoder.4)
        auto emit = [&](int d, float g) {
            if (g_y) g_y[o + d] = g;
            if (g_yb) {
                const __bf16 b = (__bf16)g;
                g_yb[o + d] = *reinterpret_cast<const unsigned short*>(&b);
            }
        };
        if (!normalize) {
            for (int d = lane; d < D; d += 64) emit(d, gval(d));
        } else {
            float dot = 0.f;
            for (int d = lane; d < D; d += 64) dot += gval(d) * (y[o + d] / n);
            dot = wave_sum(dot);
            const bool clamped = n <= 1e-12f;
            for (int d = lane; d < D; d += 64) {
                const float g = gval(d);
                emit(d, clamped ? g / n : (g - (y[o + d] / n) * dot) / n);
            }
        }
    }
    if (lane == 0) sh4[w] = term;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
}

hipError_t launch_recon_row(const float* y, const float* x, float* xh, float* ynorm, float* g_y, float* partials,
                            const float* go, bool normalize, bool cos_loss, int B, int D, hipStream_t s,
                            unsigned short* g_yb) {
    hipLaunchKernelGGL(recon_row_kernel, dim3(recon_row_blocks(B)), dim3(256), 0, s, y, x, xh, ynorm, g_y, partials, go,
                       (int)normalize, (int)cos_loss, B, D, g_yb);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Adam with device-resident {lr, step} that ALSO maintains the k-major ([in, out]) copies of the weight matrices the
// forward GEMMs read (lightning.py:205; torch.optim.Adam, non-amsgrad).  One launch over 32 x 32 tiles of the nine
// weight matrices (row-major update, transposed through LDS, whole 128-B rows both ways) plus 1024-element chunks of
// every other trained range.  The step counter lives on the device: a one-thread tick launch in front advances it, the
// main launch only reads it -- a captured HIP graph applies the right bias correction on every replay.  (A ticket inside
// the main kernel -- last workgroup to finish advances the counter -- was measured first: 1,300 workgroups taking an
// atomic ticket on one word behind a device-scope fence each cost 45 us per step; the tick launch costs 2.)
// ---------------------------------------------------------------------------------------------------------------
struct AdamDevArgs {
    float* p; float* pT; const float* g; float* m; float* v;
    long long mat_off[9]; int mat_rows[9], mat_cols[9], mat_tile_begin[9];
    long long vec_off[16], vec_len[16]; int vec_blk_begin[16];
    int n_mat, n_vec, mat_blocks, total_blocks;
    psvae_adam_state* st;
    double b1, b2;
    float eps, wd, gscale;
};

__global__ __launch_bounds__(256) void adam_dual_kernel(const AdamDevArgs a) {
    __shared__ float tile[32][33];
    __shared__ float s_hyper[2];
    const int tid = threadIdx.x;
    if (tid == 0) {
        const double lr = a.st->lr;
        const double t = (double)a.st->step;  // already advanced by adam_tick_kernel
        const double bc1 = 1.0 - pow(a.b1, t), bc2 = 1.0 - pow(a.b2, t);
        s_hyper[0] = (float)(lr / bc1);
        s_hyper[1] = (float)(1.0 / sqrt(bc2));
    }
    __syncthreads();
    const float step_size = s_hyper[0], inv_sqrt_bc2 = s_hyper[1];
    const float b1 = (float)a.b1, omb1 = (float)(1.0 - a.b1), b2 = (float)a.b2, omb2 = (float)(1.0 - a.b2);
    auto upd = [&](long long i) -> float {
        float gi = a.g[i] * a.gscale;
        const float pi = a.p[i];
        if (a.wd != 0.f) gi += a.wd * pi;
        const float mi = b1 * a.m[i] + omb1 * gi;
        const float vi = b2 * a.v[i] + omb2 * gi * gi;
        a.m[i] = mi;
        a.v[i] = vi;
        const float pn = pi - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + a.eps);
        a.p[i] = pn;
        return pn;
    };
    const int blk = blockIdx.x;
    if (blk < a.mat_blocks) {
        int k = 0;
        for (int i = 1; i < a.n_mat; ++i)
            if (blk >= a.mat_tile_begin[i]) k = i;
        const int R = a.mat_rows[k], Cc = a.mat_cols[k];
        const int tiles_c = (Cc + 31) / 32;
        const int lt = blk - a.mat_tile_begin[k];
        const int r0 = (lt / tiles_c) * 32, c0 = (lt % tiles_c) * 32;
        const long long base = a.mat_off[k];
        const int tx = tid & 31, ty = tid >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = r0 + ty + 8 * i, cc = c0 + tx;
            if (rr < R && cc < Cc) tile[ty + 8 * i][tx] = upd(base + (long long)rr * Cc + cc);
        }
        if (a.pT) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cc = c0 + ty + 8 * i, rr = r0 + tx;
                if (cc < Cc && rr < R) a.pT[base + (long long)cc * R + rr] = tile[tx][ty + 8 * i];
            }
        }
    } else {
        const int vb = blk - a.mat_blocks;
        int k = 0;
        for (int i = 1; i < a.n_vec; ++i)
            if (vb >= a.vec_blk_begin[i]) k = i;
        const long long lo = (long long)(vb - a.vec_blk_begin[k]) * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long e = lo + tid + 256 * i;
            if (e < a.vec_len[k]) upd(a.vec_off[k] + e);
        }
    }
}

__global__ void adam_tick_kernel(psvae_adam_state* st) { st->step += 1; }

hipError_t launch_adam_dual(const psvae_adam_dev* L, psvae_adam_state* st, double b1, double b2, double eps, double wd,
                            double gscale, hipStream_t s) {
    AdamDevArgs a{};
    a.p = L->param; a.pT = L->param_kmajor; a.g = L->grad; a.m = L->exp_avg; a.v = L->exp_avg_sq;
    int t = 0;
    a.n_mat = 9;
    for (int k = 0; k < 9; ++k) {
        a.mat_off[k] = L->mat_off[k]; a.mat_rows[k] = (int)L->mat_rows[k]; a.mat_cols[k] = (int)L->mat_cols[k];
        a.mat_tile_begin[k] = t;
        t += ((a.mat_rows[k] + 31) / 32) * ((a.mat_cols[k] + 31) / 32);
    }
    a.mat_blocks = t;
    a.n_vec = L->n_vec;
    int vb = 0;
    for (int k = 0; k < L->n_vec; ++k) {
        a.vec_off[k] = L->vec_off[k]; a.vec_len[k] = L->vec_len[k];
        a.vec_blk_begin[k] = vb;
        vb += (int)((L->vec_len[k] + 1023) / 1024);
    }
    a.total_blocks = t + vb;
    a.st = st; a.b1 = b1; a.b2 = b2; a.eps = (float)eps; a.wd = (float)wd; a.gscale = (float)gscale;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, s, st);
    hipLaunchKernelGGL(adam_dual_kernel, dim3(a.total_blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace psvae
