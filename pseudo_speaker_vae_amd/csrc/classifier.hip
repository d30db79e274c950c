// LatentClassifier training step on latent rows (reference: /root/reference/ps_vae/latent_classifier.py:7-70,
// its use in the training step /root/reference/ps_vae/lightning.py:73-98: y_hat = classifier(mu); cross_entropy(y_hat, y)).
//
// Covers the head the reference configures by default: single label, one Linear (hidden == 0) or
// Linear - ReLU - Linear (hidden <= 128), C <= 8 classes, Z in {16, 32, 64, 128}.  One lane owns one latent row
// (the layout of the Langevin sampler next door): the weights sit in LDS and are read as wave-uniform broadcasts, the
// row's z lives in registers.  Per row it produces the cross-entropy term, the argmax hit, d loss / d logits,
// d loss / d hidden and d loss / d z; the two weight gradients are then batch-contracting GEMMs
// (dW2 = dlogits^T h, dW1 = dh^T z, bias gradients as their column sums) on the grouped fp32 MFMA kernel, and a last
// tiny kernel reduces the loss partials in fixed order.  3 launches instead of ~20 torch ops.
#include "psvae_common.h"

namespace psvae {

constexpr int CL_MAXC = 8;

// scale = (upstream weight of the classifier loss) / B: every gradient leaves multiplied by it
template <int ZD>
__global__ __launch_bounds__(256) void classifier_ce_kernel(int64_t B, int hidden, int C, const float* __restrict__ zin,
                                                            const int64_t* __restrict__ y, const float* __restrict__ W1,
                                                            const float* __restrict__ b1, const float* __restrict__ W2,
                                                            const float* __restrict__ b2, float scale,
                                                            float* __restrict__ h_out, float* __restrict__ dh_out,
                                                            float* __restrict__ dlogits, float* __restrict__ g_z,
                                                            float* __restrict__ partials, int relu_input) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[8];
    const int rows1 = hidden > 0 ? hidden : C;
    float* sW1 = sm;                         // [rows1][ZD]
    float* sb1 = sW1 + rows1 * ZD;           // [rows1]
    float* sW2 = sb1 + ((rows1 + 3) & ~3);   // [C][hidden]
    float* sb2 = sW2 + C * (hidden > 0 ? hidden : 0);
    for (int i = threadIdx.x; i < rows1 * ZD; i += 256) sW1[i] = W1[i];
    for (int i = threadIdx.x; i < rows1; i += 256) sb1[i] = b1[i];
    if (hidden > 0) {
        for (int i = threadIdx.x; i < C * hidden; i += 256) sW2[i] = W2[i];
        for (int i = threadIdx.x; i < C; i += 256) sb2[i] = b2[i];
    }
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = row < B;
    const int64_t rr = live ? row : B - 1;  // dead lanes shadow the last row (no stores, no loss)
    float z[ZD];
#pragma unroll
    for (int k = 0; k < ZD; k += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(zin + rr * ZD + k);
        z[k] = v[0]; z[k + 1] = v[1]; z[k + 2] = v[2]; z[k + 3] = v[3];
    }
    float logits[CL_MAXC];
    uint32_t mask[4] = {0u, 0u, 0u, 0u};
    if (hidden > 0) {
#pragma unroll
        for (int c = 0; c < CL_MAXC; ++c) logits[c] = c < C ? sb2[c] : 0.f;
        for (int j = 0; j < hidden; ++j) {
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
            if (live) h_out[row * hidden + j] = a;  // saved for dW2 = dlogits^T h
#pragma unroll
            for (int c = 0; c < CL_MAXC; ++c)
                if (c < C) logits[c] = fmaf(sW2[c * hidden + j], a, logits[c]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < CL_MAXC; ++c) {
            float a = 0.f;
            if (c < C) {
                a = sb1[c];
#pragma unroll
                for (int k = 0; k < ZD; ++k) a = fmaf(sW1[c * ZD + k], z[k], a);
            }
            logits[c] = a;
        }
    }
    // cross entropy of the row: -log softmax(logits)[y]; d/dlogits = softmax - onehot(y)      (lightning.py:79)
    const int label = (int)y[rr];
    float mx = -3.0e38f;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < CL_MAXC; ++c)
        if (c < C && logits[c] > mx) { mx = logits[c]; arg = c; }
    float den = 0.f, gl[CL_MAXC];
#pragma unroll
    for (int c = 0; c < CL_MAXC; ++c) {
        gl[c] = c < C ? expf(logits[c] - mx) : 0.f;
        den += gl[c];
    }
    float ly = 0.f;
#pragma unroll
    for (int c = 0; c < CL_MAXC; ++c)
        if (c == label) ly = logits[c];
    // a label outside [0, C) (e.g. the -1 "unknown" of the label maps, utils.py:103,120,136) has no class to score: the
    // reference's F.cross_entropy raises on it.  Here the row's loss term becomes NaN, so the step's loss is visibly
    // poisoned without a host round trip; the Python wrappers additionally check the labels and raise (PSVAE_CHECK_LABELS)
    const bool bad_label = live && (label < 0 || label >= C);
    const float loss_row = live ? (bad_label ? __builtin_nanf("") : (mx + logf(den)) - ly) : 0.f;
    const float hit_row = (live && arg == label) ? 1.f : 0.f;
#pragma unroll
    for (int c = 0; c < CL_MAXC; ++c) {
        gl[c] = scale * (gl[c] / den - (c == label ? 1.f : 0.f));
        if (live && c < C) dlogits[row * C + c] = gl[c];
    }
    float gz[ZD];
#pragma unroll
    for (int k = 0; k < ZD; ++k) gz[k] = 0.f;
    if (hidden > 0) {
        for (int j = 0; j < hidden; ++j) {
            float gh = 0.f;
#pragma unroll
            for (int c = 0; c < CL_MAXC; ++c)
                if (c < C) gh = fmaf(sW2[c * hidden + j], gl[c], gh);
            gh = (mask[j >> 5] >> (j & 31)) & 1u ? gh : 0.f;
            if (live) dh_out[row * hidden + j] = gh;  // for dW1 = dh^T z
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
        for (int c = 0; c < CL_MAXC; ++c)
            if (c < C) {
#pragma unroll
                for (int k = 0; k < ZD; ++k) gz[k] = fmaf(sW1[c * ZD + k], gl[c], gz[k]);
            }
    }
    if (relu_input) {  // z is the output of a ReLU (EmbeddingClassifier fc1): hand back the gradient of its pre-activation
#pragma unroll
        for (int k = 0; k < ZD; ++k) gz[k] = z[k] > 0.f ? gz[k] : 0.f;
    }
    if (live) {
#pragma unroll
        for (int k = 0; k < ZD; k += 4)
            *reinterpret_cast<f32x4*>(g_z + row * ZD + k) = f32x4{gz[k], gz[k + 1], gz[k + 2], gz[k + 3]};
    }
    // block partial sums {loss, hits} in fixed order
    float l = loss_row, hsum = hit_row;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        l += __shfl_xor(l, o);
        hsum += __shfl_xor(hsum, o);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = l; red[4 + (threadIdx.x >> 6)] = hsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        partials[2 * blockIdx.x + 1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}

__global__ __launch_bounds__(64) void classifier_ce_final_kernel(const float* __restrict__ partials, int nblk, int64_t B,
                                                                 float* __restrict__ loss2) {
    float l = 0.f, h = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 64) { l += partials[2 * i]; h += partials[2 * i + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        l += __shfl_xor(l, o);
        h += __shfl_xor(h, o);
    }
    if (threadIdx.x == 0) {
        loss2[0] = l / (float)B;  // mean cross entropy (F.cross_entropy default reduction)
        loss2[1] = h;             // number of rows whose argmax equals the label
    }
}

int classifier_blocks(int64_t B) { return (int)((B + 255) / 256); }

hipError_t launch_classifier_ce(int64_t B, int Z, int hidden, int C, const float* z, const int64_t* y, const float* W1,
                                const float* b1, const float* W2, const float* b2, float scale, float* h_out,
                                float* dh_out, float* dlogits, float* g_z, float* partials, float* loss2, hipStream_t s,
                                int relu_input) {
    const int rows1 = hidden > 0 ? hidden : C;
    const size_t lds = (size_t)(rows1 * Z + ((rows1 + 3) & ~3) + C * hidden + 8) * sizeof(float);
    const int grid = classifier_blocks(B);
#define PSVAE_CL(ZD)                                                                                                  \
    case ZD: {                                                                                                        \
        static std::atomic<unsigned long long> lds_ok{0}; /* per device: psvae_common.h lds_opt_in */                \
        if (lds > 48 * 1024) {                                                                                        \
            /* (the kernel also has 32 B of static LDS: the dynamic limit must leave room for it) */                  \
            hipError_t ea = lds_opt_in(lds_ok, reinterpret_cast<const void*>(&classifier_ce_kernel<ZD>), 159 * 1024); \
            if (ea != hipSuccess) return ea;                                                                          \
        }                                                                                                             \
        hipLaunchKernelGGL(classifier_ce_kernel<ZD>, dim3(grid), dim3(256), lds, s, B, hidden, C, z, y, W1, b1, W2, b2, \
                           scale, h_out, dh_out, dlogits, g_z, partials, relu_input);                                           \
        break;                                                                                                        \
    }
    switch (Z) {
        PSVAE_CL(16) PSVAE_CL(32) PSVAE_CL(64) PSVAE_CL(128)
        default: return hipErrorInvalidValue;
    }
#undef PSVAE_CL
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(classifier_ce_final_kernel, dim3(1), dim3(64), 0, s, partials, grid, B, loss2);
    return hipGetLastError();
}

}  // namespace psvae
