// General form of the LatentClassifier head kernels: everything the reference's class can be configured
// to with at most one hidden layer (/root/reference/ps_vae/latent_classifier.py:7-70) --
//   * multi-label heads (num_classes a dict: one output Linear per label on a shared trunk, latent_classifier.py:41-53,
//     62-67), which the samplers score as the SUM of the labels' log-probabilities (inference.py:80-84) and the training
//     step as the MEAN of their cross entropies (lightning.py:85-95);
//   * the degenerate one-logit head (inference.py:60-66: log-softmax over one logit is identically 0, so it has no gradient);
//   * activations relu / leaky_relu (slope 0.01) / tanh / sigmoid (latent_classifier.py:17-22);
//   * up to 32 classes over all labels.
// The tuned kernels next door (langevin.hip: per-lane and MFMA forms; classifier.hip) keep the common case -- one label,
// ReLU, C <= 8 -- exactly as it was; these are the same one-lane-per-row organisation with the output Linears of all
// labels STACKED into one [C_total, hidden] matrix and a table of segments (softmax runs per segment), and with the
// activation's derivative recomputed from the pre-activation in the backward sweep (a ReLU bit mask does not carry a
// tanh'), at ~1.5x the arithmetic of the ReLU kernels.
#include "psvae_common.h"
#include "psvae_device.h"

namespace psvae {

constexpr int HG_MAXC = 32;

struct HeadSegs {
    int n;               // labels (1..8)
    int beg[8], end[8];  // rows [beg, end) of the stacked output matrix
    int target[8];       // sampler: absolute row of the label's target class (beg + target); unused by the CE step
    int act;             // 0 relu, 1 leaky_relu(0.01), 2 tanh, 3 sigmoid
    float prior_scale;   // sampler: factor on the prior's gradient (see psvae_heads)
};

__device__ __forceinline__ float hg_act(float pre, int act) {
    if (act == 0) return fmaxf(pre, 0.f);
    if (act == 1) return pre > 0.f ? pre : 0.01f * pre;
    if (act == 2) return tanhf(pre);
    return 1.0f / (1.0f + expf(-pre));
}
__device__ __forceinline__ float hg_dact(float pre, int act) {
    if (act == 0) return pre > 0.f ? 1.f : 0.f;
    if (act == 1) return pre > 0.f ? 1.f : 0.01f;
    if (act == 2) { const float a = tanhf(pre); return 1.f - a * a; }
    const float a = 1.0f / (1.0f + expf(-pre));
    return a * (1.f - a);
}

template <int ZD>
__device__ __forceinline__ float hg_dot(const float* __restrict__ wrow, float bias, const float (&z)[ZD]) {
    const f32x4* w = reinterpret_cast<const f32x4*>(wrow);
    float a0 = bias, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < ZD / 4; ++k4) {
        const f32x4 wv = w[k4];
        a0 = fmaf(wv[0], z[4 * k4], a0); a1 = fmaf(wv[1], z[4 * k4 + 1], a1);
        a2 = fmaf(wv[2], z[4 * k4 + 2], a2); a3 = fmaf(wv[3], z[4 * k4 + 3], a3);
    }
    return (a0 + a1) + (a2 + a3);
}

// logits[c] of every stacked output row for one latent row held in registers
template <int ZD>
__device__ __forceinline__ void hg_forward(const float* sW1, const float* sb1, const float* sW2, const float* sb2, int hidden,
                                           int Ct, int act, const float (&z)[ZD], float (&logits)[HG_MAXC], float* h_row) {
    if (hidden > 0) {
#pragma unroll
        for (int c = 0; c < HG_MAXC; ++c) logits[c] = c < Ct ? sb2[c] : 0.f;
        for (int j = 0; j < hidden; ++j) {
            const float a = hg_act(hg_dot<ZD>(sW1 + j * ZD, sb1[j], z), act);
            if (h_row) h_row[j] = a;
#pragma unroll
            for (int c = 0; c < HG_MAXC; ++c)
                if (c < Ct) logits[c] = fmaf(sW2[c * hidden + j], a, logits[c]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < HG_MAXC; ++c) logits[c] = c < Ct ? hg_dot<ZD>(sW1 + c * ZD, sb1[c], z) : 0.f;
    }
}

// gz += (d logits / d z)^T gl ; dh_row (nullable) receives d/d(pre-activation) of the hidden layer
template <int ZD>
__device__ __forceinline__ void hg_backward(const float* sW1, const float* sb1, const float* sW2, int hidden, int Ct, int act,
                                            const float (&z)[ZD], const float (&gl)[HG_MAXC], float (&gz)[ZD], float* dh_row) {
    if (hidden > 0) {
        for (int j = 0; j < hidden; ++j) {
            float gh = 0.f;
#pragma unroll
            for (int c = 0; c < HG_MAXC; ++c)
                if (c < Ct) gh = fmaf(sW2[c * hidden + j], gl[c], gh);
            gh *= hg_dact(hg_dot<ZD>(sW1 + j * ZD, sb1[j], z), act);
            if (dh_row) dh_row[j] = gh;
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
        for (int c = 0; c < HG_MAXC; ++c)
            if (c < Ct) {
#pragma unroll
                for (int k = 0; k < ZD; ++k) gz[k] = fmaf(sW1[c * ZD + k], gl[c], gz[k]);
            }
    }
}

template <int ZD>
__device__ __forceinline__ void hg_stage_weights(float* sm, int hidden, int Ct, const float* W1, const float* b1,
                                                 const float* W2, const float* b2, float*& sW1, float*& sb1, float*& sW2,
                                                 float*& sb2) {
    const int rows1 = hidden > 0 ? hidden : Ct;
    sW1 = sm;
    sb1 = sW1 + rows1 * ZD;
    sW2 = sb1 + ((rows1 + 3) & ~3);
    sb2 = sW2 + Ct * (hidden > 0 ? hidden : 0);
    for (int i = threadIdx.x; i < rows1 * ZD; i += 256) sW1[i] = W1[i];
    for (int i = threadIdx.x; i < rows1; i += 256) sb1[i] = b1[i];
    if (hidden > 0) {
        for (int i = threadIdx.x; i < Ct * hidden; i += 256) sW2[i] = W2[i];
        for (int i = threadIdx.x; i < Ct; i += 256) sb2[i] = b2[i];
    }
    __syncthreads();
}

// ---- Langevin chain (inference.py:72-103), any head this file covers; one lane = one latent row for the whole chain --------
template <int ZD>
__global__ __launch_bounds__(256) void langevin_general_kernel(int64_t N, int hidden, int Ct, HeadSegs segs,
                                                               const float* __restrict__ W1, const float* __restrict__ b1,
                                                               const float* __restrict__ W2, const float* __restrict__ b2,
                                                               float* __restrict__ zio, float step_size, float noise_weight,
                                                               int64_t num_steps, const float* __restrict__ noises,
                                                               uint64_t seed, uint64_t offset, float* __restrict__ history) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *sW1, *sb1, *sW2, *sb2;
    hg_stage_weights<ZD>(sm, hidden, Ct, W1, b1, W2, b2, sW1, sb1, sW2, sb2);
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = row < N;
    const int64_t rr = live ? row : N - 1;
    float z[ZD];
#pragma unroll
    for (int k = 0; k < ZD; k += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(zio + rr * ZD + k);
        z[k] = v[0]; z[k + 1] = v[1]; z[k + 2] = v[2]; z[k + 3] = v[3];
    }
    const float drift = 0.5f * step_size * step_size, nsc = step_size * noise_weight;
    const uint64_t blocks_per_step = (uint64_t)((N * ZD + 3) / 4);
    for (int64_t t = 0; t < num_steps; ++t) {
        float logits[HG_MAXC], gl[HG_MAXC];
        hg_forward<ZD>(sW1, sb1, sW2, sb2, hidden, Ct, segs.act, z, logits, nullptr);
        // d/dlogits of sum over labels of log_softmax(label's logits)[its target] = onehot(target) - softmax, per label
#pragma unroll
        for (int c = 0; c < HG_MAXC; ++c) gl[c] = 0.f;
        for (int s = 0; s < segs.n; ++s) {  // (wave-uniform trip count; the segment table sits in scalar registers)
            const int b = segs.beg[s], e = segs.end[s];
            if (e - b <= 1) continue;  // one logit: log-softmax is identically 0, no gradient (inference.py:60-66 as written)
            float mx = -3.0e38f;
#pragma unroll
            for (int c = 0; c < HG_MAXC; ++c)
                if (c >= b && c < e) mx = fmaxf(mx, logits[c]);
            float den = 0.f;
#pragma unroll
            for (int c = 0; c < HG_MAXC; ++c)
                if (c >= b && c < e) { gl[c] = expf(logits[c] - mx); den += gl[c]; }
#pragma unroll
            for (int c = 0; c < HG_MAXC; ++c)
                if (c >= b && c < e) gl[c] = (c == segs.target[s] ? 1.f : 0.f) - gl[c] / den;
        }
        float gz[ZD];
#pragma unroll
        for (int k = 0; k < ZD; ++k) gz[k] = -segs.prior_scale * z[k];  // d(-0.5|z|^2)/dz (inference.py:90)
        hg_backward<ZD>(sW1, sb1, sW2, hidden, Ct, segs.act, z, gl, gz, nullptr);
#pragma unroll
        for (int k = 0; k < ZD; k += 4) {
            f32x4 nz;
            if (noises) nz = *reinterpret_cast<const f32x4*>(noises + ((size_t)t * N + rr) * ZD + k);
            else nz = philox_normal4_hw(offset + (uint64_t)t * blocks_per_step + (uint64_t)((rr * ZD + k) >> 2), seed);
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

// ---- cross-entropy step (lightning.py:73-98): loss = mean over labels of mean-over-rows CE; y[s][B] the labels' targets ------
// partials[(1 + 2 n) * block + 0] = sum of the rows' loss terms (already / n labels), [.. + 1 + s] = hits of label s,
// [.. + 1 + n + s] = sum of the rows' cross-entropy terms of label s alone (the reference logs the labels' running sum)
template <int ZD>
__global__ __launch_bounds__(256) void classifier_ce_general_kernel(int64_t B, int hidden, int Ct, HeadSegs segs,
                                                                    const float* __restrict__ zin,
                                                                    const int64_t* __restrict__ y, const float* __restrict__ W1,
                                                                    const float* __restrict__ b1, const float* __restrict__ W2,
                                                                    const float* __restrict__ b2, float scale,
                                                                    float* __restrict__ h_out, float* __restrict__ dh_out,
                                                                    float* __restrict__ dlogits, float* __restrict__ g_z,
                                                                    float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[4 * 17];
    float *sW1, *sb1, *sW2, *sb2;
    hg_stage_weights<ZD>(sm, hidden, Ct, W1, b1, W2, b2, sW1, sb1, sW2, sb2);
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = row < B;
    const int64_t rr = live ? row : B - 1;
    float z[ZD];
#pragma unroll
    for (int k = 0; k < ZD; k += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(zin + rr * ZD + k);
        z[k] = v[0]; z[k + 1] = v[1]; z[k + 2] = v[2]; z[k + 3] = v[3];
    }
    float logits[HG_MAXC], gl[HG_MAXC];
    hg_forward<ZD>(sW1, sb1, sW2, sb2, hidden, Ct, segs.act, z, logits, (live && hidden > 0) ? h_out + row * hidden : nullptr);
#pragma unroll
    for (int c = 0; c < HG_MAXC; ++c) gl[c] = 0.f;
    float loss_row = 0.f, hits[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, lab[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float inv_n = 1.0f / (float)segs.n;
#pragma unroll  // (fully unrolled: hits[] is a register array and needs constant indices)
    for (int s = 0; s < 8; ++s) {
        if (s >= segs.n) break;
        const int b = segs.beg[s], e = segs.end[s];
        const int label = (int)y[(size_t)s * B + rr];
        float mx = -3.0e38f;
        int arg = b;
#pragma unroll
        for (int c = 0; c < HG_MAXC; ++c)
            if (c >= b && c < e && logits[c] > mx) { mx = logits[c]; arg = c; }
        float den = 0.f, ly = 0.f;
#pragma unroll
        for (int c = 0; c < HG_MAXC; ++c)
            if (c >= b && c < e) {
                gl[c] = expf(logits[c] - mx);
                den += gl[c];
                if (c - b == label) ly = logits[c];
            }
        const bool bad = label < 0 || label >= e - b;  // (the loss is poisoned with NaN: see classifier.hip)
        if (live) {
            lab[s] = bad ? __builtin_nanf("") : (mx + logf(den)) - ly;
            loss_row += lab[s] * inv_n;
        }
        if (live && arg - b == label) hits[s] = 1.f;
#pragma unroll
        for (int c = 0; c < HG_MAXC; ++c)
            if (c >= b && c < e) gl[c] = scale * inv_n * (gl[c] / den - (c - b == label ? 1.f : 0.f));
    }
    if (live) {
#pragma unroll
        for (int c = 0; c < HG_MAXC; ++c)
            if (c < Ct) dlogits[row * Ct + c] = gl[c];
    }
    float gz[ZD];
#pragma unroll
    for (int k = 0; k < ZD; ++k) gz[k] = 0.f;
    hg_backward<ZD>(sW1, sb1, sW2, hidden, Ct, segs.act, z, gl, gz, (live && hidden > 0) ? dh_out + row * hidden : nullptr);
    if (live) {
#pragma unroll
        for (int k = 0; k < ZD; k += 4)
            *reinterpret_cast<f32x4*>(g_z + row * ZD + k) = f32x4{gz[k], gz[k + 1], gz[k + 2], gz[k + 3]};
    }
    // block partial sums in fixed order: loss, then the labels' hits
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = 1 + 2 * segs.n;
    {
        float x = wave_sum(loss_row);
        if (lane == 0) red[wave * 17] = x;
    }
#pragma unroll
    for (int sg = 0; sg < 8; ++sg) {
        if (sg >= segs.n) break;
        const float xh = wave_sum(hits[sg]), xl = wave_sum(lab[sg]);
        if (lane == 0) { red[wave * 17 + 1 + sg] = xh; red[wave * 17 + 1 + segs.n + sg] = xl; }
    }
    __syncthreads();
    if (threadIdx.x < nv)
        partials[(size_t)nv * blockIdx.x + threadIdx.x] =
            (red[threadIdx.x] + red[17 + threadIdx.x]) + (red[34 + threadIdx.x] + red[51 + threadIdx.x]);
}

// loss_out[0] = mean loss, loss_out[1 + s] = hits of label s, loss_out[1 + n + s] = mean cross entropy of label s alone
__global__ __launch_bounds__(64) void classifier_ce_general_final_kernel(const float* __restrict__ partials, int nblk, int n,
                                                                         int64_t B, float* __restrict__ loss_out) {
    const int nv = 1 + 2 * n;
    for (int v = 0; v < nv; ++v) {
        float x = 0.f;
        for (int i = threadIdx.x; i < nblk; i += 64) x += partials[(size_t)nv * i + v];
        x = wave_sum(x);
        if (threadIdx.x == 0) loss_out[v] = (v == 0 || v > n) ? x / (float)B : x;
    }
}

static bool make_segs(const psvae_heads* h, int hidden, HeadSegs* out, int* Ct) {
    if (!h || h->n_heads < 1 || h->n_heads > 8 || h->activation < 0 || h->activation > 3) return false;
    int o = 0;
    out->n = h->n_heads;
    out->act = h->activation;
    out->prior_scale = h->prior_scale != 0.f ? h->prior_scale : 1.f;
    for (int s = 0; s < h->n_heads; ++s) {
        if (h->size[s] < 1) return false;
        out->beg[s] = o;
        o += h->size[s];
        out->end[s] = o;
        out->target[s] = out->beg[s] + h->target[s];
    }
    for (int s = h->n_heads; s < 8; ++s) out->beg[s] = out->end[s] = out->target[s] = 0;
    (void)hidden;
    *Ct = o;
    return o <= HG_MAXC;
}

static size_t hg_lds(int Z, int hidden, int Ct) {
    const int rows1 = hidden > 0 ? hidden : Ct;
    return (size_t)(rows1 * Z + ((rows1 + 3) & ~3) + Ct * hidden + 8) * sizeof(float);
}

hipError_t launch_langevin_general(int64_t N, int Z, int hidden, const psvae_heads* heads, const float* W1, const float* b1,
                                   const float* W2, const float* b2, float* z, float step_size, float noise_weight,
                                   int64_t num_steps, const float* noises, uint64_t seed, uint64_t offset, float* history,
                                   hipStream_t s) {
    HeadSegs segs;
    int Ct = 0;
    if (!make_segs(heads, hidden, &segs, &Ct)) return hipErrorInvalidValue;
    for (int k = 0; k < segs.n; ++k)
        if (heads->target[k] < 0 || heads->target[k] >= heads->size[k]) return hipErrorInvalidValue;
    const size_t lds = hg_lds(Z, hidden, Ct);
    const int grid = (int)((N + 255) / 256);
#define PSVAE_LG(ZD)                                                                                                            \
    case ZD: {                                                                                                                  \
        static std::atomic<unsigned long long> lds_ok{0};                                                                       \
        if (lds > 48 * 1024) {                                                                                                  \
            hipError_t ea = lds_opt_in(lds_ok, reinterpret_cast<const void*>(&langevin_general_kernel<ZD>), 160 * 1024);        \
            if (ea != hipSuccess) return ea;                                                                                    \
        }                                                                                                                       \
        hipLaunchKernelGGL(langevin_general_kernel<ZD>, dim3(grid), dim3(256), lds, s, N, hidden, Ct, segs, W1, b1, W2, b2, z,  \
                           step_size, noise_weight, num_steps, noises, seed, offset, history);                                  \
        return hipGetLastError();                                                                                               \
    }
    switch (Z) {
        PSVAE_LG(16) PSVAE_LG(32) PSVAE_LG(64) PSVAE_LG(128)
        default: return hipErrorInvalidValue;
    }
#undef PSVAE_LG
}

hipError_t launch_classifier_ce_general(int64_t B, int Z, int hidden, const psvae_heads* heads, const float* z, const int64_t* y,
                                        const float* W1, const float* b1, const float* W2, const float* b2, float scale,
                                        float* h_out, float* dh_out, float* dlogits, float* g_z, float* partials,
                                        float* loss_out, hipStream_t s) {
    HeadSegs segs;
    int Ct = 0;
    if (!make_segs(heads, hidden, &segs, &Ct)) return hipErrorInvalidValue;
    const size_t lds = hg_lds(Z, hidden, Ct);
    const int grid = (int)((B + 255) / 256);
#define PSVAE_CG(ZD)                                                                                                            \
    case ZD: {                                                                                                                  \
        static std::atomic<unsigned long long> lds_ok{0};                                                                       \
        if (lds > 48 * 1024) {                                                                                                  \
            hipError_t ea = lds_opt_in(lds_ok, reinterpret_cast<const void*>(&classifier_ce_general_kernel<ZD>), 159 * 1024);   \
            if (ea != hipSuccess) return ea;                                                                                    \
        }                                                                                                                       \
        hipLaunchKernelGGL(classifier_ce_general_kernel<ZD>, dim3(grid), dim3(256), lds, s, B, hidden, Ct, segs, z, y, W1, b1,  \
                           W2, b2, scale, h_out, dh_out, dlogits, g_z, partials);                                               \
        break;                                                                                                                  \
    }
    switch (Z) {
        PSVAE_CG(16) PSVAE_CG(32) PSVAE_CG(64) PSVAE_CG(128)
        default: return hipErrorInvalidValue;
    }
#undef PSVAE_CG
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(classifier_ce_general_final_kernel, dim3(1), dim3(64), 0, s, partials, grid, segs.n, B, loss_out);
    return hipGetLastError();
}

}  // namespace psvae
