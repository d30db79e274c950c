/* psvae.h -- C ABI of libpsvae_hip.so: the MI355X (gfx950) implementation of the pseudo-speaker VAE
 * hot path.  Plain C types only (device pointers as typed pointers, sizes as int64_t, hipStream_t
 * as void*).  The reference (nmehlman/pseudo_speaker_VAE) is pure Python and has no FFI of its own;
 * each entry point below names the reference interface (file:line under /root/reference) it stands
 * in for, and INTEGRATION.md shows the ctypes binding a maintainer adds to ps_vae/model.py.
 *
 * Conventions
 *   - every buffer is caller-owned device memory (fp32, row-major, contiguous unless stated);
 *     the library allocates nothing persistent and keeps no pointer after a call returns; its only process
 *     state is the table of tuned launch plans per (B, D, Z) (psvae_tune_train_step_f32 / psvae_set_plan:
 *     small host structs, speed only) and the memo of launch-configuration choices -- both behind mutexes;
 *     per-kernel attributes are set under std::call_once; the debug hooks are atomics: the entry points may be
 *     called from several host threads (one stream each);
 *   - all work is enqueued on `stream`; no call synchronises the device or the host;
 *   - return 0 on success, <0 for an argument error detected before any launch, >0 = hipError_t;
 *     psvae_strerror() maps both; no C++ exception crosses the boundary;
 *   - weights are nn.Linear layout [out, in]; hidden width is PSVAE_HIDDEN (model.py:15-35).
 */
#ifndef PSVAE_H
#define PSVAE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSVAE_ABI_VERSION 4
#define PSVAE_HIDDEN 512
#define PSVAE_NPARAM 18
#define PSVAE_COND_DIM 6 /* onehot3(gender) || onehot3(age), SURVEY.md section 8a row A9 */

/* flags */
#define PSVAE_NORMALIZE_DECODER 1u /* model.py:60-61, 67-68 */
#define PSVAE_COS_LOSS 2u          /* lightning.py:110-111 */
/* psvae_*_bf16 passes only: the bf16 weight packs in `scratch` are current (psvae_pack_params_bf16 ran after the last
 * parameter update), do not rebuild them in this pass */
#define PSVAE_BF16_PACKS_CURRENT 0x100u

/* backward phases (psvae_backward_f32 `phases`): decoder grads are complete after phase 1, so a
 * data-parallel caller can start their all-reduce while phase 2 (both encoders) still runs */
#define PSVAE_PHASE_DECODER 1u
#define PSVAE_PHASE_ENCODER 2u
#define PSVAE_PHASE_ALL 3u

/* error codes (<0) */
#define PSVAE_OK 0
#define PSVAE_ERR_NULL -1   /* a required pointer is NULL */
#define PSVAE_ERR_SHAPE -2  /* non-positive or unsupported dimension */
#define PSVAE_ERR_ALIGN -3  /* a pointer is not 16-byte aligned */
#define PSVAE_ERR_ARG -4    /* bad enum / flag / phase value */
#define PSVAE_ERR_STATE -5  /* call not valid in this state (e.g. tuning while the stream is capturing) */

typedef struct psvae_dims {
    int64_t B; /* batch rows */
    int64_t D; /* input_dim  (model.py:8) */
    int64_t Z; /* latent_dim (model.py:8) */
} psvae_dims;

/* The 18 parameter tensors in creation order (model.py:14-36):
 *  [0..5]  encoder_mu.{0,2,4}.{weight,bias}    W[512,D] b[512] W[512,512] b[512] W[Z,512] b[Z]
 *  [6..11] encoder_sigma.{0,2,4}.{weight,bias} same shapes
 *  [12..17] decoder.{0,2,4}.{weight,bias}      W[512,Z] b[512] W[512,512] b[512] W[D,512] b[D] */
typedef struct psvae_params {
    const float* p[PSVAE_NPARAM];
} psvae_params;

typedef struct psvae_grads {
    float* g[PSVAE_NPARAM]; /* same order/shapes; OVERWRITTEN, never accumulated */
} psvae_grads;

/* Optional conditional tables (build-defined cfg3 extension; NULL pointers = unconditional).
 * tab[k] is [6, 512]: row j is the first-layer weight column of one-hot slot j. */
typedef struct psvae_cond {
    const float* tab[3];   /* encoder_mu, encoder_sigma, decoder */
    float* gtab[3];        /* their gradients (backward only; may be NULL in forward) */
    const int32_t* gender; /* [B] in {-1,0,1,2}; utils.py:105-120 */
    const int32_t* age;    /* [B] in {-1,0,1,2}; utils.py:82-103 */
} psvae_cond;

/* Activations saved by forward for backward (caller-owned). */
typedef struct psvae_acts {
    float* h1m; /* [B,512] post-ReLU, encoder_mu.1 */
    float* h2m; /* [B,512] post-ReLU, encoder_mu.3 */
    float* h1s; /* [B,512] encoder_sigma.1 */
    float* h2s; /* [B,512] encoder_sigma.3 */
    float* eps; /* [B,Z] the N(0,1) draw of model.py:57 */
    float* z;   /* [B,Z] */
    float* d1;  /* [B,512] decoder.1 */
    float* d2;  /* [B,512] decoder.3 */
    float* ynorm; /* [B] max(||y||,1e-12); only with PSVAE_NORMALIZE_DECODER, else may be NULL */
} psvae_acts;

int psvae_version(void);
/* tuning hook: force the GEMM config (1..9, see gemm_f32.hip) of every later launch; 0 = automatic */
int psvae_debug_force_tile(int tile);
/* tuning hook: weight-gradient GEMMs cut the batch over s workgroups per output tile (1..4); 0 = automatic */
int psvae_debug_force_wgrad_split(int s);
const char* psvae_strerror(int code);

/* bytes of scratch the calls below need for a batch of B rows (any call; take the max over calls) */
size_t psvae_scratch_bytes(const psvae_dims* dims);

/* Test hook for the epsilon generator: n standard normals, Philox4x32-10 + Box-Muller; element e
 * comes from Philox block (offset + e/4), key = seed.  Replaces torch.randn_like (model.py:57). */
int psvae_philox_normal_f32(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
/* The same Philox blocks through Box-Muller on the hardware transcendental instructions (v_log_f32, v_sqrt_f32, v_sin_f32 /
 * v_cos_f32 in revolutions): straight-line code, within 2e-6 absolute of the stream above (measured: 1.8e-6 over 4 M
 * normals).  This is, bit for bit, what every kernel that draws INSIDE its work produces -- the epsilon of the training passes
 * (psvae_forward_* / psvae_train_step_* with eps_in == NULL), the noise of the Langevin samplers, the z of
 * psvae_sample_decode_* -- and what a test injects to reproduce them. */
int psvae_philox_normal_fast_f32(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);

/* VAEModel.forward (model.py:38-63).  eps_in != NULL injects the normal draw (parity tests);
 * eps_in == NULL draws it from Philox(seed, offset) -- either way it is written to acts->eps.
 * rng_state (nullable, device, 2 x uint64 {seed, offset}) overrides the host (seed, offset) and is
 * advanced on the device by the Philox blocks consumed: a captured hipGraph then draws fresh noise
 * on every replay.  Outputs x_hat[B,D], mu[B,Z], logvar[B,Z] (the reference's "log_sigma" is a
 * log-variance). */
int psvae_forward_f32(const psvae_dims* dims, const float* x, const psvae_params* params,
                      const psvae_cond* cond /* nullable */, const float* eps_in /* nullable */,
                      uint64_t seed, uint64_t offset, uint64_t* rng_state /* nullable */,
                      const psvae_acts* acts, float* x_hat, float* mu, float* logvar, void* scratch,
                      uint32_t flags, void* stream);

/* ELBO of training_step (lightning.py:110-124): loss3 = {total, recon, kl} (device, 3 floats).
 * recon = mean((x_hat-x)^2)/10, or mean_b(1-cos) with PSVAE_COS_LOSS; kl = -0.5 mean_b sum_j(..). */
int psvae_elbo_f32(const psvae_dims* dims, const float* x, const float* x_hat, const float* mu,
                   const float* logvar, float beta, float* loss3, void* scratch, uint32_t flags,
                   void* stream);

/* Gradients of the two ELBO terms of psvae_elbo_f32 w.r.t. the forward outputs (for callers that combine the terms
 * with other losses under autograd: the multi-label / frozen-VAE branches of lightning.py:67-131):
 *   g_xhat = *g_recon * d recon / d x_hat;   g_mu = *g_kl * mu / B;   g_logvar = *g_kl * (exp(logvar) - 1) / (2B)
 * g_recon / g_kl: device scalars (NULL = 1).  flags: PSVAE_COS_LOSS selects the cosine reconstruction term.  2 launches. */
int psvae_elbo_grad_f32(const psvae_dims* dims, const float* x, const float* x_hat, const float* mu,
                        const float* logvar, const float* g_recon, const float* g_kl, float* g_xhat, float* g_mu,
                        float* g_logvar, uint32_t flags, void* stream);

/* Backward of forward (+ optionally of the ELBO): overwrites the 18 gradients.
 *   g_xhat / g_mu / g_logvar: upstream gradients of the three forward outputs (each nullable = 0);
 *   elbo != 0 adds grad_loss * d(total)/d(.) of psvae_elbo_f32 (grad_loss: device scalar, NULL = 1);
 *   g_x (nullable): gradient w.r.t. the input x (the reference never needs it; autograd callers may).
 * Stands in for loss.backward() after lightning.py:131. */
int psvae_backward_f32(const psvae_dims* dims, const float* x, const psvae_params* params,
                       const psvae_cond* cond /* nullable */, const psvae_acts* acts, const float* x_hat,
                       const float* mu, const float* logvar, const float* g_xhat, const float* g_mu,
                       const float* g_logvar, int elbo, float beta, const float* grad_loss,
                       const psvae_grads* grads, float* g_x, void* scratch, uint32_t flags,
                       uint32_t phases, void* stream);

/* Optional extras of psvae_train_step_f32 (a NULL struct or NULL / zero members = the defaults). */
typedef struct psvae_step_opts {
    /* k-major ([in, out]) copies of the 9 weight matrices in p[0], p[2], ... p[16] (the bias slots are ignored): the
     * forward GEMMs read these (whole 128-B lines per k).  An optimizer that owns the weights keeps them current
     * (psvae_adam_step_dev_f32 writes both layouts); NULL: the step makes the copies itself, one more launch. */
    const psvae_params* params_kmajor;
    const psvae_cond* cond; /* one-hot conditional tables; NULL = unconditional */
    uint32_t flags;         /* PSVAE_NORMALIZE_DECODER | PSVAE_COS_LOSS (model.py:60-61, lightning.py:110-111) */
} psvae_step_opts;

/* One whole training pass = psvae_forward_f32 + psvae_elbo_f32 + psvae_backward_f32(elbo=1) as ONE launch sequence:
 * what a Lightning step does between lightning.py:67 and the end of loss.backward().  10 launches for the default
 * loss when Z is a multiple of 64 (<= 128) and the k-major weight copies are supplied:
 *   F1, F2 (+ the mu / logvar heads as a second MFMA stage on each finished tile: partial products), the forward
 *   bottleneck kernel (sums the partials; reparameterise, KL, decoder.0), F6, F7 (+ MSE gradient and partial sums),
 *   B1, B2 (+ decoder.0's dgrad as a second stage), the backward bottleneck kernel (g_mu, g_logvar, the heads' dgrad,
 *   loss, Philox advance), B5, B6 (with the three Z-sized weight gradients riding along);
 * +1 (weight transposes) without the copies, +1 (row loss kernel) for the cosine / normalised variants, +1 or 2
 * (table gradients) with conditioning, +1 when a tuned plan splits a weight gradient over workgroups; other Z take the
 * 12-launch sequence with separate heads / decoder.0-dgrad launches.
 * phases: PSVAE_PHASE_DECODER = forward + decoder backward (decoder gradients and loss3 complete),
 * PSVAE_PHASE_ENCODER = encoder backward. */
int psvae_train_step_f32(const psvae_dims* dims, const float* x, const psvae_params* params,
                         const float* eps_in /* nullable */, uint64_t seed, uint64_t offset,
                         uint64_t* rng_state /* nullable */, const psvae_acts* acts, float* x_hat, float* mu,
                         float* logvar, float beta, const float* grad_loss /* nullable = 1 */,
                         const psvae_grads* grads, float* loss3, void* scratch, uint32_t phases,
                         const psvae_step_opts* opts /* nullable */, void* stream);
/* kernel launches the calling thread's most recent psvae_train_step_f32 issued (evidence for the launch count) */
int psvae_last_step_launches(void);
/* make the k-major copies of the 9 weight matrices (params_kmajor->p[2k] <- transpose of params->p[2k]); one launch */
int psvae_transpose_params_f32(const psvae_dims* dims, const psvae_params* params, const psvae_params* params_kmajor,
                               void* stream);

/* Optional, once per (B, D, Z): time the fused step under candidate launch plans (GEMM config per launch, weight-
 * gradient split, workgroups per CU, tile order) and keep the fastest for later psvae_train_step_f32 calls in this
 * process.  Runs the step a few thousand times on `stream` (not while capturing) and overwrites acts / outputs /
 * grads / loss3.  Every plan computes the same result; only speed differs. */
int psvae_tune_train_step_f32(const psvae_dims* dims, const float* x, const psvae_params* params,
                              const psvae_acts* acts, float* x_hat, float* mu, float* logvar,
                              const psvae_grads* grads, float* loss3, void* scratch, int iters,
                              const psvae_step_opts* opts /* nullable */, void* stream);
/* text form of the plan in use for (B, D, Z) ("untuned" if none) */
int psvae_describe_plan(const psvae_dims* dims, char* buf, size_t n);
/* install a plan from that text form ("untuned" removes it); PSVAE_ERR_ARG if it does not parse */
int psvae_set_plan(const psvae_dims* dims, const char* text);

/* bf16 variant of psvae_train_step_f32 (BASELINE config 3: bf16 MFMA operands, fp32 accumulate, fp32 master
 * weights, gradients and loss).  D and Z must be multiples of 32.  x, params, grads, outputs, eps_out[B,Z] and
 * z_out[B,Z] are fp32; all bf16 intermediates live in `scratch` (psvae_bf16_scratch_bytes).  Activations and
 * gradients exist in ONE bf16 layout: the weight gradients read their batch-major operands transposed out of LDS
 * (ds_read_b64_tr_b16).  cond (nullable) adds the one-hot gather tables.
 * From B = 64 up, with D in {128, 256, 512} and Z in {64, 128}, the pass runs as 7 launches (6 with
 * PSVAE_BF16_PACKS_CURRENT): the weight packs, then every 3-layer MLP (the encoders, the decoder) as one "row chain" launch
 * per direction with its activations in LDS (csrc/chain_bf16.hip), all weight gradients as one launch
 * (csrc/wgrad_bf16.hip) and the sums of its partial slabs; other shapes take a launch per layer.
 * flags: the loss variants of the fp32 step (L2-normalised decoder output, model.py:60-61; cosine loss, lightning.py:110-111),
 * available on the row-chain shapes (PSVAE_ERR_SHAPE elsewhere): the decoder chain leaves the raw output and one row kernel
 * normalises, takes the loss term and the gradient w.r.t. the raw output. */
size_t psvae_bf16_scratch_bytes(const psvae_dims* dims);
/* The bf16 working copies of the 9 weight matrices, in the layouts the row chains read (16 packs in `scratch`), from the
 * fp32 master weights: one launch.  A pass rebuilds them itself unless it is told they are current
 * (flags | PSVAE_BF16_PACKS_CURRENT): the mixed-precision arrangement in which the optimizer step, which changes the
 * master weights (lightning.py:205, torch.optim.Adam), also refreshes the working copies -- as the fp32 step's k-major
 * copies are (psvae_step_opts::params_kmajor).  No-op (PSVAE_OK) on shapes that take a launch per layer. */
int psvae_pack_params_bf16(const psvae_dims* dims, const psvae_params* params, void* scratch, void* stream);
int psvae_train_step_bf16(const psvae_dims* dims, const float* x, const psvae_params* params,
                          const psvae_cond* cond /* nullable */, const float* eps_in /* nullable */,
                          uint64_t seed, uint64_t offset, uint64_t* rng_state /* nullable */, float* eps_out,
                          float* z_out, float* x_hat, float* mu, float* logvar, float beta,
                          const float* grad_loss /* nullable = 1 */, const psvae_grads* grads, float* loss3,
                          void* scratch, uint32_t flags /* PSVAE_NORMALIZE_DECODER | PSVAE_COS_LOSS */, void* stream);

/* The weight-gradient launch of the bf16 pass by itself: dW[out,in] = dY^T X and db[out] = column sums of dY (nullable) --
 * what autograd leaves in an nn.Linear's .weight.grad / .bias.grad (model.py:14-36).  dY [B, ldy], X [B, ldx]: bf16,
 * row-major, pitches multiples of 8 elements, bases 16-byte aligned; results fp32.  split (1..8): the batch is cut over that
 * many workgroups per 128 x 128 tile, their partial results summed in slice order (scratch: psvae_wgrad_bf16_scratch_bytes).
 * A building block and a test hook: integer-valued operands make every sum exact (tests/test_gpu_wgrad.py). */
size_t psvae_wgrad_bf16_scratch_bytes(int out, int in, int split);
int psvae_wgrad_bf16(const uint16_t* dY, int64_t ldy, const uint16_t* X, int64_t ldx, int64_t B, int out, int in, float* dW,
                     float* db /* nullable */, int split, void* scratch /* nullable when split == 1 */, void* stream);

/* VAEModel.decode (model.py:65-69): x_hat[N,D] = decoder(z[N,Z]); dec = the 6 decoder tensors
 * (params->p[12..17]).  Used by inference.py:25,105.  scratch: psvae_decode_scratch_bytes(N). */
size_t psvae_decode_scratch_bytes(int64_t N);
/* bf16 variant: bf16 operands and hidden activations, fp32 accumulation / biases / x_hat; Z % 32 == 0,
 * D % 4 == 0; no conditional tables.  About 8-bit-mantissa accuracy on x_hat (see tests/test_gpu_bf16.py). */
size_t psvae_decode_bf16_scratch_bytes(int64_t N, int64_t D, int64_t Z);
int psvae_decode_bf16(int64_t N, int64_t D, int64_t Z, const float* z, const float* const* dec6, float* x_hat,
                      void* scratch, uint32_t flags, void* stream);
int psvae_decode_f32(int64_t N, int64_t D, int64_t Z, const float* z, const float* const* dec6,
                     const float* cond_tab /* nullable [6,512] */, const int32_t* gender,
                     const int32_t* age, float* x_hat, void* scratch, uint32_t flags, void* stream);
/* unconditional_synthesis in ONE call (inference.py:10-27: z = torch.randn((num_samples, dim)); x_hat = decode(z)):
 * z[N,Z] ~ N(0, I) from the Philox stream (seed, offset) -- element e of the flat array is normal (e % 4) of block
 * offset + e / 4, exactly what psvae_philox_normal_f32(z, N * Z, seed, offset) writes -- then x_hat = decoder(z).  On the
 * fused decode's shapes (see psvae_last_decode_launches) z is drawn INSIDE the decode kernel: no draw launch, no round
 * trip of z through memory; z_out is nullable there and receives the fp32 z when given.  Other shapes draw into z_out
 * first (PSVAE_ERR_NULL without it) and call psvae_decode_*.  Same scratch as psvae_decode_f32 / psvae_decode_bf16. */
int psvae_sample_decode_f32(int64_t N, int64_t D, int64_t Z, uint64_t seed, uint64_t offset, const float* const* dec6,
                            float* z_out /* nullable on fused shapes */, float* x_hat, void* scratch, uint32_t flags,
                            void* stream);
int psvae_sample_decode_bf16(int64_t N, int64_t D, int64_t Z, uint64_t seed, uint64_t offset, const float* const* dec6,
                             float* z_out /* nullable on fused shapes */, float* x_hat, void* scratch, uint32_t flags,
                             void* stream);
/* kernel launches the calling thread's most recent psvae_decode_f32 issued: the fused decode (all three
 * layers in one kernel; N >= 16384, no conditional table, Z in {32,64,128}, D in {128,256,512}) issues 4 (three weight
 * packs + the kernel) plus the normalisation launches, the per-layer path 3 per 65,536-row chunk */
int psvae_last_decode_launches(void);

/* Test / bench hook: one fp32 MFMA GEMM of the kind the path is built from.
 *   kind 0 (NT): C[M,N] = A[M,K] B[N,K]^T     (forward Linear, model.py:15)
 *   kind 1 (NN): C[M,N] = A[M,K] B[K,N]       (dgrad)
 *   kind 2 (TN): C[M,N] = A[K,M]^T B[K,N]     (wgrad; colsum != NULL also writes sum_k A[k,:])
 *   epi 0 none, 1 +bias[N], 2 relu(+bias), 3 * (mask[M,N] > 0), 4 C += .  tile: 0 = auto, else a
 *   config id 1..9 (wave tile / k-split table in csrc/gemm_f32.hip). */
int psvae_gemm_f32(int kind, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                   const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias_or_mask,
                   float* colsum, int epi, int tile, void* stream);

/* Fused Langevin conditional sampler: the loop of conditional_synthesis (ps_vae/inference.py:72-103) for a
 * single-label ReLU LatentClassifier with one hidden layer (hidden = its width <= 128, W1[hidden,Z], W2[C,hidden];
 * latent_classifier.py:30-40) or a single Linear (hidden = 0, W1[C,Z]).  z[N,Z] is updated in place over
 * num_steps steps:  z += 0.5 s^2 grad(log p(y=target|z) - 0.5|z|^2) + s w noise.  noises (nullable, [T,N,Z])
 * injects the per-step normals (parity tests), else they come from Philox(seed, offset + t*ceil(N*Z/4) + ..) as
 * psvae_philox_normal_fast_f32 writes them;
 * history (nullable, [T,N,Z]) receives z after every step.  Z in {16,32,64,128}, C <= 8. */
int psvae_langevin_f32(int64_t N, int64_t Z, int64_t hidden, int64_t C, const float* W1, const float* b1,
                       const float* W2, const float* b2, float* z, int32_t target, float step_size,
                       float noise_weight, int64_t num_steps, const float* noises, uint64_t seed, uint64_t offset,
                       float* history, void* stream);

/* The general head: everything LatentClassifier can be configured to with at most one hidden layer
 * (latent_classifier.py:7-70) -- several labels (num_classes a dict: one output Linear per label on the shared trunk,
 * stacked here into ONE [sum size, hidden] matrix W2 / [sum size] bias b2, or [sum size, Z] W1 when hidden == 0), the
 * one-logit head (no gradient: inference.py:60-66 as written), activation 0 relu / 1 leaky_relu(0.01) / 2 tanh / 3 sigmoid,
 * up to 32 classes in all.  The single-label ReLU C <= 8 case keeps its tuned entry points (psvae_langevin_f32,
 * psvae_classifier_ce_f32). */
typedef struct psvae_heads {
    int32_t n_heads;      /* labels, 1..8 */
    int32_t size[8];      /* classes per label */
    int32_t target[8];    /* sampler: the label's target class in [0, size); unused by the cross-entropy step */
    int32_t activation;
    float prior_scale;    /* sampler: factor on the prior's gradient -z; 1, except for the reference's single one-logit head,
                           * whose [N,1] + [N] broadcast (inference.py:62-64,90-92) sums the prior N times: pass N there
                           * (0 is read as 1) */
} psvae_heads;
/* conditional_synthesis's loop (inference.py:72-103) scoring log p(y|z) as the SUM of the labels' log-probabilities
 * (inference.py:80-84); arguments as psvae_langevin_f32 */
int psvae_langevin_heads_f32(int64_t N, int64_t Z, int64_t hidden, const psvae_heads* heads, const float* W1, const float* b1,
                             const float* W2, const float* b2, float* z, float step_size, float noise_weight,
                             int64_t num_steps, const float* noises, uint64_t seed, uint64_t offset, float* history,
                             void* stream);
/* the training step's classifier term (lightning.py:73-98): loss = MEAN over the labels of the mean-over-rows cross
 * entropy (lightning.py:85-95), y = [n_heads][B] int64; loss_out[1 + 2 n_heads]: [0] = that loss, [1 + s] = rows whose
 * argmax hits label s, [1 + n_heads + s] = mean cross entropy of label s alone; g_z, gW1 [rows1, Z], gb1, gW2 [sum size, hidden], gb2 as in psvae_classifier_ce_f32 (stacked like W2).
 * scratch: psvae_classifier_scratch_bytes(B, hidden, sum size). */
int psvae_classifier_heads_ce_f32(int64_t B, int64_t Z, int64_t hidden, const psvae_heads* heads, const float* z,
                                  const int64_t* y, const float* W1, const float* b1, const float* W2, const float* b2,
                                  float grad_scale, float* loss_out, float* g_z, float* gW1, float* gb1, float* gW2,
                                  float* gb2, void* scratch, void* stream);

/* kernel launches the calling thread's most recent bf16 pass (train step, forward or backward call) issued */
int psvae_last_step_launches_bf16(void);

/* The two halves of psvae_train_step_bf16 as separate calls (a validation step runs only the first).  The forward leaves
 * the bf16 activations, the MSE gradient (already times *grad_loss) and the loss partial sums in `scratch`; the backward
 * must be given the same scratch, x, params, cond, eps_out, mu and logvar, untouched.  rng_state (nullable) supplies
 * {seed, offset} to the forward and is advanced by the backward. */
int psvae_forward_bf16(const psvae_dims* dims, const float* x, const psvae_params* params, const psvae_cond* cond,
                       const float* eps_in, uint64_t seed, uint64_t offset, const uint64_t* rng_state, float* eps_out,
                       float* z_out, float* x_hat, float* mu, float* logvar, const float* grad_loss, void* scratch,
                       uint32_t flags, void* stream);
int psvae_backward_bf16(const psvae_dims* dims, const float* x, const psvae_params* params, const psvae_cond* cond,
                        uint64_t* rng_state, float* eps_out, float* z_out, float* x_hat, float* mu, float* logvar,
                        float beta, const float* grad_loss, const psvae_grads* grads, float* loss3, void* scratch,
                        uint32_t flags, void* stream);

/* LatentClassifier training step on latent rows (latent_classifier.py:7-70 as used at lightning.py:73-98):
 * logits = W1 z + b1 (hidden == 0) or W2 relu(W1 z + b1) + b2; loss2 = {mean cross entropy against y, number of rows
 * whose argmax equals the label}; g_z[B,Z] and the parameter gradients are d(mean CE)/d(.) times grad_scale
 * (overwritten).  Single-label heads with C <= 8, hidden <= 128, Z in {16, 32, 64, 128}; labels in [0, C): a row whose
 * label is outside that range (the label maps' -1) makes the loss NaN -- the reference's F.cross_entropy raises on it,
 * and so do the Python wrappers (latent_classifier.check_labels) before the call.
 * 3 launches (per-row kernel, loss reduction, one grouped GEMM for the weight gradients). */
size_t psvae_classifier_scratch_bytes(int64_t B, int64_t hidden, int64_t C);
int psvae_classifier_ce_f32(int64_t B, int64_t Z, int64_t hidden, int64_t C, const float* z, const int64_t* y,
                            const float* W1, const float* b1, const float* W2 /* hidden > 0 */,
                            const float* b2 /* hidden > 0 */, float grad_scale, float* loss2, float* g_z, float* gW1,
                            float* gb1, float* gW2, float* gb2, void* scratch, void* stream);

/* EmbeddingClassifier consistency term on reconstructed embeddings (SURVEY N4;
 * embedding_classifier/embedding_classifier.py:28-31,49-62 as used at lightning.py:44-52,100-108):
 * logits = fc3 relu(fc2 relu(fc1 x_hat)), fc1 [hidden, D], fc2 [hidden, hidden], fc3 [C, hidden];
 * loss2 = {mean cross entropy against y, number of rows whose argmax equals the label};
 * g_xhat[B,D] = d(mean CE)/d(x_hat) * grad_scale (overwritten).  The head is frozen inside the VAE's training step
 * (lightning.py:48-50): pass pg = NULL.  With pg the six parameter gradients {gW1, gb1, gW2, gb2, gW3, gb3} are
 * written too (training the head itself, embedding_classifier.py:64-82).  hidden in {16, 32, 64, 128}, C <= 8,
 * D % 4 == 0, labels in [0, C).  4 launches: fc1 GEMM, per-row kernel, loss reduction, one grouped GEMM. */
typedef struct psvae_mlp3_grads {
    float* g[6];
} psvae_mlp3_grads;
size_t psvae_consistency_scratch_bytes(int64_t B, int64_t hidden, int64_t C);
int psvae_consistency_ce_f32(int64_t B, int64_t D, int64_t hidden, int64_t C, const float* x_hat, const int64_t* y,
                             const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                             const float* b3, float grad_scale, float* loss2, float* g_xhat,
                             const psvae_mlp3_grads* pg /* nullable */, void* scratch, void* stream);

/* Batch assembly from a packed embedding table resident in HBM: out[b,:] = table[idx[b],:] (an
 * out-of-range index gives a zero row).  Stands in for the per-sample torch.load of
 * ps_vae/data/cv.py:58-76 / vctk.py:51-67 (one tiny file per embedding). */
int psvae_gather_rows_f32(const float* table, int64_t N, int64_t D, const int64_t* idx, int64_t B, float* out,
                          void* stream);

/* Fused multi-tensor Adam on flat buffers (lightning.py:205; torch.optim.Adam semantics).
 * grad_scale multiplies the gradient first (1/world_size after an all-reduce(sum)).  Hyper-parameters
 * are doubles so that 1-beta etc. round exactly as torch's (python-float) arithmetic does. */
int psvae_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                        int64_t step, double lr, double beta1, double beta2, double eps, double weight_decay,
                        double grad_scale, void* stream);

/* The same update over a LIST of tensors (any addresses) in ONE launch: torch.optim.Adam(params) as lightning.py:205
 * builds it -- 18 separate parameter tensors, plus the classifier's -- without flattening anything; what
 * pseudo_speaker_vae_amd.optim.FusedAdam.step() calls.  At most PSVAE_ADAM_MAX_TENSORS tensors per call. */
#define PSVAE_ADAM_MAX_TENSORS 48
int psvae_adam_multi_f32(int n_tensors, float* const* param, const float* const* grad, float* const* exp_avg,
                         float* const* exp_avg_sq, const int64_t* numel, int64_t step, double lr, double beta1, double beta2,
                         double eps, double weight_decay, double grad_scale, void* stream);

/* Adam with its step counter and learning rate ON THE DEVICE (a captured hipGraph then applies the right bias
 * correction on every replay, and CosineAnnealingLR, lightning.py:206-213, is one 8-byte write per epoch), which also
 * maintains the k-major weight copies psvae_step_opts::params_kmajor points at.  The flat buffers share one layout:
 * mat_off/rows/cols = the 9 weight matrices ([rows, cols] = [out, in]) inside them, the k-major copy of matrix k lives
 * at the same offset of param_kmajor as [cols, rows]; vec_off/len = every other trained range (biases, conditional
 * tables).  *state must be zero-initialised except lr.  Two launches (a one-thread tick of the step counter, then the
 * update). */
typedef struct psvae_adam_state {
    double lr;
    int64_t step;   /* completed optimizer steps; advanced on the device by every call */
    uint32_t reserved[2];
} psvae_adam_state;
typedef struct psvae_adam_dev {
    float* param;
    float* param_kmajor; /* nullable */
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t mat_off[9], mat_rows[9], mat_cols[9];
    int64_t vec_off[16], vec_len[16];
    int32_t n_vec;
} psvae_adam_dev;
int psvae_adam_step_dev_f32(const psvae_adam_dev* layout, psvae_adam_state* state /* device */, double beta1,
                            double beta2, double eps, double weight_decay, double grad_scale, void* stream);

/* ---- direct gradient exchange between the ranks of one node (training.py:78: DDP's gradient mean) --------------------
 * One-shot reduce-scatter + all-gather through buffers every rank maps from every other rank (hipIpcMemHandle): two hops,
 * each a kernel that pulls 1/W of the payload from each peer over all xGMI links at once -- instead of RCCL's ring.
 * Stream-ordered: the phases are separated by interprocess events (hipIpcEventHandle) the consumers' streams wait on; no
 * kernel spins on a flag.  The hosts hand each other one sequence number per phase through a page of POSIX shared memory
 * (so that a wait binds to THIS step's record), which makes the call not graph-capturable: enqueue it behind the step.
 *   psvae_ipc_comm_create:  allocate this rank's buffers / events; blob_out (psvae_ipc_blob_bytes()) goes to every rank;
 *                           `tag`: the same string on every rank, unique per communicator (names the shared page)
 *   psvae_ipc_comm_connect: all_blobs = the `world` blobs in rank order
 *   psvae_ipc_allreduce_avg_f32: g[0..nfloats) <- mean over ranks, in place, on `stream`; every rank calls it once per step.
 *                           Ranks end with bit-identical values (each chunk is summed once, in rank order).  Returns
 *                           PSVAE_ERR_STATE if a peer's host does not arrive within timeout_s.  nfloats % 4 == 0, g 16-B aligned.
 * Exception to "the library allocates nothing": the exported buffers must come from hipMalloc in the exporting process. */
typedef struct psvae_ipc_comm psvae_ipc_comm;
size_t psvae_ipc_blob_bytes(void);
int psvae_ipc_comm_create(int rank, int world, size_t nfloats, const char* tag, psvae_ipc_comm** out, void* blob_out);
int psvae_ipc_comm_connect(psvae_ipc_comm* comm, const void* all_blobs);
int psvae_ipc_allreduce_avg_f32(psvae_ipc_comm* comm, float* g, double timeout_s, void* stream);
int psvae_ipc_comm_destroy(psvae_ipc_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* PSVAE_H */
