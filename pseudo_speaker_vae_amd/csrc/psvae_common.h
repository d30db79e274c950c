// Internal declarations shared by the HIP translation units of libpsvae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/psvae.h"

namespace psvae {

constexpr int H = PSVAE_HIDDEN;
constexpr int WAVE = 64;

// Raise a kernel's dynamic-LDS limit (> 64 KB needs the opt-in) once per (kernel, DEVICE): the attribute belongs to the
// device that is current at the call, so a process that drives several devices -- psvae.h allows calls from several
// host threads, one stream each -- opts every one of them in.  `done` is the call site's static bit mask over device
// ordinals; a failure is returned and NOT cached (the next call tries again); racing first calls both set the
// attribute, which is idempotent.  No other mutable state per kernel (SURVEY 8b threading contract).
static inline hipError_t lds_opt_in(std::atomic<unsigned long long>& done, const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && bit) done.fetch_or(bit, std::memory_order_release);
    return e;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- grouped fp32 MFMA GEMM -------------------------------------------------------------------
enum GemmKind { GK_NT = 0, GK_NN = 1, GK_TN = 2 };
enum GemmEpi {
    EPI_NONE = 0,
    EPI_BIAS = 1,       // C = acc + bias[col]
    EPI_BIAS_RELU = 2,  // C = relu(acc + bias[col])
    EPI_MASK = 3,       // C = acc * (mask[row][col] > 0)     (ReLU backward, mask = saved activation)
    EPI_ACCUM = 4,      // C += acc                            (second dgrad into the same buffer)
    EPI_BIAS_RECON = 5, // C = acc + bias; C2 = scale*go*(C - aux2); partials[tile] = sum (C - aux2)^2
    EPI_REPARAM_BWD = 6, // acc = g_z: C = g_mu, C2 = g_logvar of reparameterisation + KL (see GemmDesc); loss finalize
                        // (decoder output + MSE gradient + MSE partial sums, lightning.py:113)
};

struct GemmDesc {
    const float* A;
    const float* B;
    float* C;
    const float* aux;      // bias[N] (EPI_BIAS*) or mask[M,ldc] (EPI_MASK)
    float* colsum;         // GK_TN only, nullable: colsum[m] = sum_k A[k][m]  (bias gradient)
    const float* ctab;     // nullable: conditional table [6,N] added per row in EPI_BIAS* (cfg3)
    const int32_t* gender; // [rows] labels for ctab
    const int32_t* age;
    const float* aux2;     // EPI_BIAS_RECON: the target x[M,ldc]
    float* C2;             // EPI_BIAS_RECON: gradient output g_y[M,ldc]
    const float* go;       // EPI_BIAS_RECON: device scalar multiplying the gradient (nullable = 1)
    float* partials;       // EPI_BIAS_RECON: one partial sum per workgroup tile of this problem
    float scale;
    // EPI_REPARAM_BWD (fused step): aux = mu, aux2 = log-variance, aux3 = eps, C = g_mu, C2 = g_logvar, scale = beta;
    // the workgroup of tile (0, 0) also finishes the loss from the partial sums and advances the Philox state
    const float* aux3;
    const float* kl_part;   // n_kl partial sums of (1 + lv - mu^2 - e^lv)
    float* loss3;           // {total, recon, kl}
    unsigned long long* rng;     // nullable: rng[1] += rng_adv
    unsigned long long rng_adv;
    int n_kl, n_recon, loss_D;  // `partials` holds the n_recon loss partial sums; loss_D = input dim (MSE: sum/(M D)/10),
                                // or -D when they are per-row cosine terms (recon = sum / M)
    int M, N, K;
    int lda, ldb, ldc;
    int kind, epi;
    int cfg;  // GEMM config id of this problem (set by finalize_batch)
    int ksplit;  // > 1: the k range is cut over `ksplit` WORKGROUPS; workgroup slice s writes its partial tile to
                 // C + s*M*ldc and its partial column sums to colsum + s*M (summed later by launch_slab_reduce)
    int tiles_m, tiles_n;
    int tile_begin;  // first linear tile of this problem inside the launch
    // Second stage (config 6 only; nullable): after the epilogue the finished 64 x 64 tile T (post bias/ReLU or post mask)
    // is multiplied from LDS with rows [n0, n0+64) of B2[N, Z2] (row-major, Z2 % 64 == 0):
    //   slab2[tn][row][0..Z2) = T[row][:] * B2[n0.., :]        (one partial product per column tile tn of this problem)
    // The consumer sums the tiles_n partials in fixed order.  Folds the skinny layer that FOLLOWS this one (mu / logvar
    // heads after encoder layer 1, decoder.0's dgrad after decoder.2's) into this launch: N % 64 == 0 required.
    const float* B2;
    float* slab2;
    int Z2;
};

constexpr int MAX_GROUP = 6;
struct GemmBatch {
    GemmDesc d[MAX_GROUP];
    int n;
    int total_tiles;
    int dbg;  // reserved (0)
    int xcd_remap;  // 1: XCD-aware tile order
    // host-side launch hints of a tuned plan (0 = cost model / default); the kernel ignores them
    int hint_tn, hint_other;  // config of the TN problems / of the others
    int hint_cap;             // 1: at most one workgroup per CU (through the LDS request), 2: no cap, 0: default
    int hint_remap;           // 1: XCD-aware tile order, 2: plain order, 0: default
};

// A tuned launch plan of the fused train step: one entry per GEMM launch ("site"), chosen by timing the whole
// step (psvae_tune_train_step_f32).  All-zero = heuristics.
struct SitePlan {
    signed char cfg_tn, cfg_other;  // 0 = cost model
    signed char ksplit;             // weight-gradient cross-workgroup split (0 = default rule)
    signed char cap;                // see GemmBatch::hint_cap
    signed char remap;              // see GemmBatch::hint_remap
};
constexpr int NSITE = 12;  // F1 F2 F3 F5 F6 F7 B1 B2 B3 B4 B5 B6
struct StepPlan { SitePlan site[NSITE]; };

// cfg = 1..num_gemm_configs() selects a (wave tile, wave layout, k-split, prefetch depth) config
// (table in gemm_f32.hip); cfg == 0 picks one from the problem sizes with a cost model (pick_tile).
int pick_tile(const GemmBatch& b);
int num_gemm_configs();
void debug_force_tile(int cfg);  // 0 = automatic
void finalize_batch(GemmBatch& b, int cfg);
hipError_t launch_gemm_batch(GemmBatch& b, int cfg, hipStream_t stream);

inline GemmDesc make_desc(int kind, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                          float* C, int ldc, int epi, const float* aux, float* colsum = nullptr) {
    GemmDesc d{};
    d.A = A; d.B = B; d.C = C; d.aux = aux; d.colsum = colsum;
    d.ctab = nullptr; d.gender = nullptr; d.age = nullptr;
    d.aux2 = nullptr; d.C2 = nullptr; d.go = nullptr; d.partials = nullptr; d.scale = 1.f;
    d.ksplit = 1;
    d.aux3 = nullptr; d.kl_part = nullptr; d.loss3 = nullptr; d.rng = nullptr; d.rng_adv = 0;
    d.n_kl = 0; d.n_recon = 0; d.loss_D = 0;
    d.B2 = nullptr; d.slab2 = nullptr; d.Z2 = 0;
    d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc; d.kind = kind; d.epi = epi;
    return d;
}

// ---- bf16 path (gemm_bf16.hip): every GEMM is C[m][n] = sum_k A[m][k] B[n][k], bf16 operands ------------
enum BfEpi { BEPI_NONE = 0, BEPI_RELU = 1, BEPI_MASK = 2, BEPI_RECON = 3 };
struct GemmDescBf {
    const unsigned short* A;     // bf16 [M, lda], k contiguous
    const unsigned short* B;     // bf16 [N, ldb], k contiguous
    float* Cf;                   // nullable: fp32 output [M, ldcf]
    unsigned short* Cb;          // nullable: bf16 output [M, ldcb]
    const float* bias;           // nullable: fp32 [N]
    const unsigned short* mask;  // BEPI_MASK: saved bf16 activation [M, ldmask]
    const float* target;         // BEPI_RECON: fp32 x [M, ldcf]
    const float* go;             // BEPI_RECON: device scalar (nullable = 1)
    float* partials;             // BEPI_RECON: one partial sum of squares per tile
    float* colsum;               // nullable: colsum[m] = sum_k A[m][k]  (bias gradient in the wgrad form)
    const float* ctab;           // nullable: conditional table [6, N]
    const int32_t* gender;
    const int32_t* age;
    float scale;
    int M, N, K;                 // K % 32 == 0
    int ksplit;                  // > 1 (LDS-staged 128x128 kernel only): the k range is cut over `ksplit` workgroups; slice s
                                 // writes its partial fp32 tile to Cf + s*M*ldcf and partial column sums to colsum + s*M
    int lda, ldb, ldcf, ldcb, ldmask;
    int tr;                      // 1 (LDS-staged kernel only): both operands are k-MAJOR in memory -- A is [K, lda] with M valid
                                 // columns, B is [K, ldb] with N (weight gradients: K = batch rows, any K; pitches % 8 == 0)
    int epi;
    int tiles_m, tiles_n, tile_begin;
};
constexpr int MAX_GROUP_BF = 12;  // (all nine weight gradients and the three table gradients of a pass fit one launch)
struct GemmBatchBf {
    GemmDescBf d[MAX_GROUP_BF];
    int n;
    int total_tiles;
    int force_staged;  // host-side: 1 = launch with the LDS-staged 128x128 kernel whatever the tile count
    int xcd_remap;     // 1: XCD-aware tile order inside each problem (set by the launcher)
    int tiles_per_slice;  // wgrad_bf16.hip: tiles of all problems of ONE batch slice (its work list is slice-major)
};
hipError_t launch_gemm_batch_bf16(GemmBatchBf& b, hipStream_t stream);
// wgrad_bf16.hip: a batch of tr = 1 problems (weight gradients) on the LDS-DMA pipeline; same descriptors, fp32 out
hipError_t launch_wgrad_batch_bf16(GemmBatchBf& b, hipStream_t stream);
hipError_t launch_cast_transpose(const float* const* src, unsigned short* const* dst, unsigned short* const* dstT,
                                 const int* R, const int* C, const int* ld, const int* ldt, int n, hipStream_t s,
                                 const int32_t* gender = nullptr, const int32_t* age = nullptr,
                                 unsigned short* oh = nullptr, int B = 0);  // oh: one-hot label matrix [B][8] (rider blocks)
hipError_t launch_cond_colsum_bf16(const unsigned short* const* dY, float* const* gtab, int nprob,
                                   const int32_t* gender, const int32_t* age, int B, int N, hipStream_t s);

// ---- elementwise / row kernels (elementwise.hip) ------------------------------------------------
hipError_t launch_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t s);
hipError_t launch_philox_normal_fast(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t s);
// z = mu + exp(0.5 lv) * eps ; eps from eps_in or Philox; writes eps_out, z
// rng (nullable, device {seed, offset}) overrides seed/offset and is advanced by ceil(n/4) afterwards
hipError_t launch_reparam(const float* mu, const float* lv, const float* eps_in, uint64_t seed,
                          uint64_t offset, uint64_t* rng, float* eps_out, float* z, int64_t n, hipStream_t s);
// fused-step variant: also writes one KL partial sum per block (kl_partials[reparam_blocks(n)]) and
// leaves the rng advance to launch_reparam_bwd_final
int reparam_blocks(int64_t n);
// reparameterisation + KL partials + decoder layer 0 in one launch (Z % 16 == 0, Z <= 128, N % 4 == 0, 16-B aligned
// pointers); kl_partials gets reparam_linear_tiles(B) entries (bottleneck.hip)
bool reparam_linear_ok(int Z, int N, const void* const* ptrs, int nptr);
int reparam_linear_tiles(int B);
hipError_t launch_reparam_linear(const float* mu, const float* lv, const float* eps_in, const uint64_t* rng,
                                 uint64_t seed, uint64_t offset, const float* W, const float* bias, float* eps_out,
                                 float* z_out, float* Y, float* kl_partials, int B, int Z, int N, hipStream_t s);
struct BnFwdArgs {
    // S == 0: mu / lv hold the finished heads' outputs (read).  S >= 1: mu = b3m + sum_s slab_m[s], lv likewise, and
    // the column-0 workgroups write them to mu / lv.
    float* mu;
    float* lv;
    const float* slab_m;  // [S][B][Z]
    const float* slab_s;
    const float* b3m;     // [Z]
    const float* b3s;
    int S;
    const float* eps_in;          // nullable: injected epsilon
    const uint64_t* rng;          // nullable: device {seed, offset}
    uint64_t seed, offset;
    const float* W;               // decoder.0 weight [N, Z] (k contiguous)
    const float* bias;            // [N]
    const float* ctab;            // nullable: conditional table [6, N]
    const int32_t* gender;
    const int32_t* age;
    float* eps_out;               // [B, Z]
    float* z_out;                 // [B, Z]
    float* Y;                     // [B, N] = relu(z W^T + bias (+ table rows))
    float* kl_partials;           // one per 16-row tile
    int B, Z, N;
};

struct BnBwdArgs {
    const float* slab_gz;   // [S][B][Z]
    int S;
    const float* mu;
    const float* lv;
    const float* eps;
    const float* go;        // nullable device scalar (upstream d loss)
    float beta;
    float* g_mu;            // [B, Z] out (the heads' weight gradients read them later)
    float* g_lv;
    const float* W3mT;      // k-major copies [N, Z] of encoder_mu.4 / encoder_sigma.4 weights ([Z, N] row-major originals)
    const float* W3sT;
    const float* h2m;       // [B, N] saved post-ReLU activations = masks
    const float* h2s;
    float* dh2m;            // [B, N] out
    float* dh2s;
    // loss finalisation
    const float* kl_partials; int n_kl;
    const float* recon_partials; int n_recon;
    int recon_mode;         // 0: recon = sum / (B D) / 10 (MSE, lightning.py:113); 1: recon = sum / B (cosine, :111)
    int D;
    float* loss3;
    unsigned long long* rng; unsigned long long rng_adv;
    int B, Z, N;
};

hipError_t launch_bottleneck_fwd(const BnFwdArgs& a, hipStream_t s);
hipError_t launch_bottleneck_bwd(const BnBwdArgs& a, hipStream_t s);
int recon_row_blocks(int B);
hipError_t launch_recon_row(const float* y, const float* x, float* xh, float* ynorm, float* g_y, float* partials,
                            const float* go, bool normalize, bool cos_loss, int B, int D, hipStream_t s,
                            unsigned short* g_yb = nullptr);  // ynorm, g_y, g_yb nullable; g_yb: the gradient as bf16
hipError_t launch_adam_dual(const psvae_adam_dev* L, psvae_adam_state* st, double b1, double b2, double eps, double wd,
                            double gscale, hipStream_t s);
hipError_t launch_reparam_kl(const float* mu, const float* lv, const float* eps_in, const uint64_t* rng,
                             uint64_t seed, uint64_t offset, float* eps_out, float* z, float* kl_partials,
                             int64_t n, hipStream_t s, unsigned short* zb = nullptr);  // zb: optional bf16 copy of z [B,Z]
// row L2 normalise: xh = y / max(||y||, 1e-12), ynorm = that max
hipError_t launch_normalize_rows(const float* y, float* xh, float* ynorm, int B, int D, hipStream_t s);
// loss3 = {total, recon, kl}
hipError_t launch_elbo_loss(const float* x, const float* xh, const float* mu, const float* lv, int B, int D,
                            int Z, float beta, bool cos_loss, float* partials, float* loss3, hipStream_t s);
int elbo_partial_blocks(int B);
// g_y = d/dy [upstream g_xhat + go * recon term], through the optional row normalisation
hipError_t launch_recon_grad(const float* x, const float* xh, const float* ynorm, const float* g_xhat,
                             bool elbo, bool cos_loss, bool normalize, const float* go, float* g_y, int B,
                             int D, hipStream_t s);
// g_mu = g_z + [g_mu_up] + go*beta*mu/B ; g_lv = g_z*eps*0.5*exp(0.5 lv) + [g_lv_up] + go*beta*(e^lv-1)/(2B)
hipError_t launch_reparam_bwd(const float* g_z, const float* mu, const float* lv, const float* eps,
                              const float* g_mu_up, const float* g_lv_up, bool elbo, float beta,
                              const float* go, float* g_mu, float* g_lv, int B, int Z, hipStream_t s);
hipError_t launch_kl_grad(const float* mu, const float* lv, const float* g, float* g_mu, float* g_lv, int B, int Z,
                          hipStream_t s);
// fused-step variant of reparam_bwd: block 0 additionally finishes the loss (fixed-order sums of the KL
// and MSE partials -> loss3 = {total, recon, kl}) and advances the device Philox state
hipError_t launch_reparam_bwd_final(const float* g_z, const float* mu, const float* lv, const float* eps,
                                    float beta, const float* go, float* g_mu, float* g_lv, int B, int Z, int D,
                                    const float* kl_partials, int n_kl, const float* recon_partials, int n_recon,
                                    float* loss3, uint64_t* rng, hipStream_t s,
                                    unsigned short* const* bf_rm = nullptr);  // optional bf16 copies of {g_mu, g_lv} [B,Z]
// gtab[i][c][n] = sum_{b: label_b = c} dY[i][b][n] for up to 3 problems (conditional-table gradients)
hipError_t launch_cond_colsum(const float* const* dY, float* const* gtab, int nprob, const int32_t* gender,
                              const int32_t* age, int B, int N, hipStream_t s);
// dst[i][e] = sum_{s < S[i]} src[i][s*numel[i] + e]   (fixed order: bitwise reproducible), up to 24 tensors
hipError_t launch_slab_reduce(const float* const* src, float* const* dst, const int64_t* numel, const int* S, int n,
                              hipStream_t s);
hipError_t launch_transpose_f32(const float* const* src, float* const* dst, const int* R, const int* C, int n,
                                hipStream_t s);
// LatentClassifier cross-entropy step (classifier.hip): per-row kernel + loss reduction; the weight gradients are GEMMs
int classifier_blocks(int64_t B);
hipError_t launch_classifier_ce(int64_t B, int Z, int hidden, int C, const float* z, const int64_t* y, const float* W1,
                                const float* b1, const float* W2, const float* b2, float scale, float* h_out,
                                float* dh_out, float* dlogits, float* g_z, float* partials, float* loss2, hipStream_t s,
                                int relu_input = 0);
// heads_general.hip: multi-label / non-ReLU / up to 32 classes
hipError_t launch_langevin_general(int64_t N, int Z, int hidden, const psvae_heads* heads, const float* W1, const float* b1,
                                   const float* W2, const float* b2, float* z, float step_size, float noise_weight,
                                   int64_t num_steps, const float* noises, uint64_t seed, uint64_t offset, float* history,
                                   hipStream_t s);
hipError_t launch_classifier_ce_general(int64_t B, int Z, int hidden, const psvae_heads* heads, const float* z, const int64_t* y,
                                        const float* W1, const float* b1, const float* W2, const float* b2, float scale,
                                        float* h_out, float* dh_out, float* dlogits, float* g_z, float* partials,
                                        float* loss_out, hipStream_t s);
hipError_t launch_langevin(int64_t N, int Z, int hidden, int C, const float* W1, const float* b1, const float* W2,
                           const float* b2, float* z, int target, float step_size, float noise_weight,
                           int64_t num_steps, const float* noises, uint64_t seed, uint64_t offset, float* history,
                           hipStream_t s);
hipError_t launch_gather_rows(const float* table, int64_t N, int D, const int64_t* idx, int64_t B, float* out,
                              hipStream_t s);
hipError_t launch_adam_multi(int nt, float* const* p, const float* const* g, float* const* m, float* const* v,
                             const int64_t* numel, int64_t step, double lr, double b1, double b2, double eps, double wd,
                             double gscale, hipStream_t s);
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, int64_t step, double lr,
                       double b1, double b2, double eps, double wd, double gscale, hipStream_t s);

// decode_fused.hip: the three decoder layers in one kernel (hidden activations stay in registers)
bool decode_fused_ok(int64_t N, int Z, int D);
size_t decode_fused_pack_floats(int Z, int D);
// z == NULL: the kernel draws z itself (Philox normals of (seed, offset), the stream psvae_philox_normal_f32 writes) and
// leaves it in z_out when that is given
hipError_t launch_decode_fused(int64_t N, int Z, int D, const float* z, const float* const* dec6, float* x_hat, float* wp,
                               hipStream_t s, unsigned long long seed = 0, unsigned long long offset = 0,
                               float* z_out = nullptr);
bool decode_fused_bf16_ok(int64_t N, int Z, int D);
hipError_t launch_decode_fused_bf16(int64_t N, int Z, int D, const float* z, const float* const* dec6, float* x_hat,
                                    unsigned short* wp, hipStream_t s, unsigned long long seed = 0,
                                    unsigned long long offset = 0, float* z_out = nullptr);

// chain_bf16.hip: a 3-layer MLP of the bf16 training pass (one encoder, or the decoder) over 64-row tiles in one launch
struct ChainArgs {
    const float* in;                 // [B, K0] fp32
    int B, K0, NOUT;
    const unsigned short* wp[3];     // chain-packed bf16 weights (launch_chain_pack)
    const float* bias[3];
    const float* ctab;               // nullable [6, 512]: layer-0 one-hot tables
    const int32_t* gender;
    const int32_t* age;
    unsigned short* save[2];         // row-major bf16 [B, 512]: the two hidden activations
    float* out;                      // [B, NOUT] fp32
    const float* target;             // decoder only (gy != nullptr): x, the upstream scale, 2 / (10 B D)
    const float* go;
    float scale;
    unsigned short* gy;              // [B, NOUT] bf16 MSE gradient
    float* partials;                 // [tiles] sums of squared errors
    // decoder of a training pass (mu != nullptr): the input is z = mu + exp(lv / 2) eps, built in the kernel (`in` unused)
    const float* mu;
    const float* lv;
    const float* eps_in;             // nullable: draw with Philox (rng {seed, offset} on the device if given, else seed / offset)
    const uint64_t* rng;
    uint64_t seed, offset;
    float* eps_out;                  // [B, K0] fp32
    float* z_out;                    // [B, K0] fp32
    unsigned short* zb;              // [B, K0] bf16
    float* kl_partials;              // [tiles]: sum of 1 + lv - mu^2 - exp(lv)
    int raw_out;                     // 1: only write the raw third-layer output to `out` (the loss variants' row kernel follows)
    // encoder of a training pass (nullable): the bf16 row-major copy of its input rows [B, K0] and the one-hot label matrix
    // [B][8] (columns 6, 7 zero) -- the weight-gradient launch's operands, written as the rows pass through
    unsigned short* xb_out;
    unsigned short* oh_out;
};
struct ChainBatch {
    ChainArgs c[2];
    int n;
};
bool chain_fwd_ok(int B, int K0, int NOUT);
int chain_rows_per_tile(int B, int nchains);
hipError_t launch_chain_pack(const float* const* W, unsigned short* const* Wp, const int* OUT, const int* IN, const int* T,
                             int n, const float* x, unsigned short* xb, long nx, const int32_t* gender, const int32_t* age,
                             unsigned short* oh, int B, hipStream_t s);
struct ChainBwdArgs {
    const void* in;                  // [B, K0]: bf16 row-major (in_bf16) or fp32
    int in_bf16;
    int B, K0, NOUT;                 // NOUT = 0: two layers only
    const unsigned short* wp[3];     // transposed chain-packed weights
    const unsigned short* mask[2];   // saved forward activations [B, 512] bf16 (ReLU masks of layers 0 and 1)
    unsigned short* save[2];         // [B, 512] bf16 row-major hidden gradients
    float* out;                      // [B, NOUT] fp32 (plain third-layer output; unused when mu != nullptr)
    // decoder of a training pass (mu != nullptr): the third layer's g_z becomes g_mu / g_logvar on the way out, and tile 0
    // finishes the loss and advances the Philox state
    const float* mu;
    const float* lv;
    const float* eps;
    const float* go;                 // nullable device scalar
    float beta;
    float* g_mu;                     // [B, NOUT] fp32
    float* g_lv;
    unsigned short* gmu_b;           // [B, NOUT] bf16
    unsigned short* glv_b;
    const float* kl_partials;
    const float* recon_partials;
    int n_kl, n_recon;
    int recon_mode;                  // 0: recon = sum / (B D) / 10 (MSE); 1: sum / B (cosine loss)
    float* loss3;
    uint64_t* rng;                   // nullable
    uint64_t nblk;
};
struct ChainBwdBatch {
    ChainBwdArgs c[2];
    int n;
};
hipError_t launch_chain_bwd_bf16(const ChainBwdBatch& b, hipStream_t s);
hipError_t launch_chain_fwd_bf16(const ChainBatch& b, hipStream_t s);

}  // namespace psvae
