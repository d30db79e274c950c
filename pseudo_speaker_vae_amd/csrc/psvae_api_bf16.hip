// extern "C" entry points of the bf16 path (BASELINE config 3): bf16 MFMA operands, fp32 accumulation,
// fp32 master weights / gradients / loss.  One call = one training pass; see gemm_bf16.hip for the layout
// idea (activations in ONE bf16 layout; the weight gradients read their k-major operands transposed out of LDS).
#include "psvae_common.h"

using namespace psvae;
typedef unsigned short u16;

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define RET_IF(e)                            \
    do {                                     \
        hipError_t _e = (e);                 \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)

// Every backward launch uses the LDS-staged 128x128 kernel (half the operand bytes per FLOP of the 64x64 tiles: the bf16
// GEMMs are pure operand traffic; the weight gradients read their batch-major operands transposed out of the staged
// tiles).  The weight gradients have only 16 such tiles each, so they cut the batch over S workgroups per tile (512
// batch rows per slice, at least 256); the fp32 partial slabs are summed in fixed order by one launch at the end.
constexpr int BF_SMAX = 8;  // (S = 2, 4, 6, 8, 10, 12, 16 measured at B = 4096: 380, 262, 229, 211, 222, 220, 213 us per step)
inline int bf_wgrad_split(int B) {
    const char* e = getenv("PSVAE_BF16_WGRAD_S");  // tuning / tests
    if (e) { const int v = atoi(e); return v < 1 ? 1 : (v > BF_SMAX ? BF_SMAX : v); }
    const int s = B >= 2048 ? B / 512 : (B + 255) / 256;
    return s < 1 ? 1 : (s > BF_SMAX ? BF_SMAX : s);
}

struct BfScratch {
    // weights: row-major bf16 copies Wb[i] ([out,in]) and transposed WbT[i] ([in,out]) of the 9 matrices
    u16 *Wb[9], *WbT[9];
    u16 *Wc[9], *WcT[9];                            // chain-packed copies (chain_bf16.hip): W for the forward chains, W^T for the backward
    u16 *xb;                                        // [B,D]
    u16 *h1m, *h2m, *h1s, *h2s, *d1, *d2;           // [B,H] activations, ONE layout each
    u16 *zb;                                        // [B,Z]
    u16 *gy;                                        // [B,D]
    u16 *dd2, *dd1, *dh2m, *dh2s, *dh1m, *dh1s;     // [B,H]
    u16 *gmu, *glv;                                 // [B,Z]
    u16* oh;                                        // [B,8] one-hot labels (columns 6, 7 zero)
    float *gz, *gmu_f, *glv_f, *kl_partials, *recon_partials;
    float* y_raw;                                   // [B,D] un-normalised decoder output (PSVAE_NORMALIZE_DECODER only)
    // large batches: weight-gradient partials of the cross-workgroup split (BF_SMAX slices each), summed at the end
    float *slab_w[9], *slab_b[9], *slab_t[3];
    int n_recon_max;
    size_t bytes;
};

// dims of the 9 weight matrices in params order (weights are params[0,2,4, 6,8,10, 12,14,16])
inline void weight_dims(int D, int Z, int (&out_)[9], int (&in_)[9]) {
    const int o[9] = {H, H, Z, H, H, Z, H, H, D};
    const int i[9] = {D, H, H, D, H, H, Z, H, H};
    for (int k = 0; k < 9; ++k) { out_[k] = o[k]; in_[k] = i[k]; }
}

BfScratch carve_bf(void* base, int64_t B, int64_t D, int64_t Z) {
    BfScratch s{};
    size_t off = 0;
    const int64_t Bp = (B + 31) / 32 * 32;
    auto take = [&](size_t nbytes) {
        char* p = base ? reinterpret_cast<char*>(base) + off : nullptr;
        off += align_up(nbytes, 256);
        return p;
    };
    auto tb = [&](size_t n) { return reinterpret_cast<u16*>(take(n * 2)); };
    auto tf = [&](size_t n) { return reinterpret_cast<float*>(take(n * 4)); };
    int wo[9], wi[9];
    weight_dims((int)D, (int)Z, wo, wi);
    for (int k = 0; k < 9; ++k) {
        s.Wb[k] = tb((size_t)wo[k] * wi[k]);
        s.WbT[k] = tb((size_t)wi[k] * wo[k]);
        s.Wc[k] = tb((size_t)wo[k] * wi[k]);
        s.WcT[k] = tb((size_t)wo[k] * wi[k]);
    }
    (void)Bp;
    s.xb = tb(B * D);
    u16** hs[] = {&s.h1m, &s.h2m, &s.h1s, &s.h2s, &s.d1, &s.d2, &s.dd2, &s.dd1, &s.dh2m, &s.dh2s, &s.dh1m, &s.dh1s};
    for (int k = 0; k < 12; ++k) *hs[k] = tb(B * H);
    s.zb = tb(B * Z);
    s.gy = tb(B * D);
    s.gmu = tb(B * Z); s.glv = tb(B * Z);
    s.oh = tb(8 * B);
    s.gz = tf(B * Z); s.gmu_f = tf(B * Z); s.glv_f = tf(B * Z);
    s.kl_partials = tf((size_t)reparam_blocks(B * Z));
    s.y_raw = tf(B * D);
    s.n_recon_max = (int)(((B + 15) / 16) * ((D + 31) / 32));
    if (s.n_recon_max < recon_row_blocks((int)B)) s.n_recon_max = recon_row_blocks((int)B);
    s.recon_partials = tf((size_t)s.n_recon_max);
    for (int k = 0; k < 9; ++k) {
        s.slab_w[k] = tf((size_t)BF_SMAX * wo[k] * wi[k]);
        s.slab_b[k] = tf((size_t)BF_SMAX * wo[k]);
    }
    for (int k = 0; k < 3; ++k) s.slab_t[k] = tf((size_t)BF_SMAX * PSVAE_COND_DIM * H);
    s.bytes = off;
    return s;
}

thread_local int t_bf_launches = 0;  // launches of this thread's current / last bf16 pass
// which path (row chains or a launch per layer) this thread's last forward took, and on which scratch / shape
struct FwdPath { const void* scratch; int B, D, Z; bool chain; };
thread_local FwdPath t_fwd_path = {nullptr, 0, 0, 0, false};

inline GemmDescBf bdesc(int M, int N, int K, const u16* A, int lda, const u16* B, int ldb) {
    GemmDescBf d{};
    d.A = A; d.B = B; d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb;
    d.epi = BEPI_NONE; d.scale = 1.f;
    return d;
}
inline void addb(GemmBatchBf& b, const GemmDescBf& d) { b.d[b.n++] = d; }

// the 16 weight packs of the row chains (9 plain for the forward chains, 7 transposed for the gradient chains): one launch
inline hipError_t chain_pack_weights(const BfScratch& sc, const float* const* P, const int* wo, const int* wi, hipStream_t s) {
    const float* Wsrc[16]; u16* Wdst[16]; int po[16], pi[16], pt[16]; int np = 0;
    for (int k = 0; k < 9; ++k) { Wsrc[np] = P[2 * k]; Wdst[np] = sc.Wc[k]; po[np] = wo[k]; pi[np] = wi[k]; pt[np++] = 0; }
    for (int k = 0; k < 9; ++k)
        if (k != 0 && k != 3) { Wsrc[np] = P[2 * k]; Wdst[np] = sc.WcT[k]; po[np] = wo[k]; pi[np] = wi[k]; pt[np++] = 1; }
    return launch_chain_pack(Wsrc, Wdst, po, pi, pt, np, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, s);
}

}  // namespace

extern "C" {

int psvae_pack_params_bf16(const psvae_dims* dims, const psvae_params* params, void* scratch, void* stream) {
    if (!dims || !params || !scratch) return PSVAE_ERR_NULL;
    if (dims->B <= 0 || dims->D <= 0 || dims->Z <= 0 || (dims->D & 31) || (dims->Z & 31)) return PSVAE_ERR_SHAPE;
    for (int k = 0; k < 9; ++k)
        if (!params->p[2 * k]) return PSVAE_ERR_NULL;
    const int B = (int)dims->B, D = (int)dims->D, Z = (int)dims->Z;
    if (!(chain_fwd_ok(B, D, Z) && chain_fwd_ok(B, Z, D))) return PSVAE_OK;  // per-layer shapes cast inside every pass
    BfScratch sc = carve_bf(scratch, B, D, Z);
    int wo[9], wi[9];
    weight_dims(D, Z, wo, wi);
    RET_IF(chain_pack_weights(sc, params->p, wo, wi, (hipStream_t)stream));
    return PSVAE_OK;
}

size_t psvae_bf16_scratch_bytes(const psvae_dims* dims) {
    if (!dims || dims->B <= 0 || dims->D <= 0 || dims->Z <= 0) return 0;
    return carve_bf(nullptr, dims->B, dims->D, dims->Z).bytes;
}

// One training pass in bf16 (config 3).  x, params, grads, loss3, x_hat, mu, logvar are fp32; every
// intermediate lives in `scratch` (psvae_bf16_scratch_bytes).  D and Z must be multiples of 32.
// mode: 1 = forward (everything up to x_hat, the MSE gradient and the loss partial sums), 2 = backward, 3 = both
static int bf16_pass(int mode, const psvae_dims* dims, const float* x, const psvae_params* params, const psvae_cond* cond,
                     const float* eps_in, uint64_t seed, uint64_t offset, uint64_t* rng_state, float* eps_out,
                     float* z_out, float* x_hat, float* mu, float* logvar, float beta, const float* grad_loss,
                     const psvae_grads* grads, float* loss3, void* scratch, uint32_t flags, void* stream) {
    const bool fwd = mode & 1, bwd = mode & 2;
    t_bf_launches = 0;
#define LAUNCH(e) do { RET_IF(e); ++t_bf_launches; } while (0)
    const bool normalize = flags & PSVAE_NORMALIZE_DECODER, cos_loss = flags & PSVAE_COS_LOSS;
    const bool packs_current = flags & PSVAE_BF16_PACKS_CURRENT;  // (row-chain path: skip the weight-pack launch)
    if (flags & ~(uint32_t)(PSVAE_NORMALIZE_DECODER | PSVAE_COS_LOSS | PSVAE_BF16_PACKS_CURRENT)) return PSVAE_ERR_ARG;
    if (!dims || !x || !params || !x_hat || !mu || !logvar || !scratch || !eps_out || !z_out) return PSVAE_ERR_NULL;
    if (bwd && (!grads || !loss3)) return PSVAE_ERR_NULL;
    if (dims->B <= 0 || dims->D <= 0 || dims->Z <= 0) return PSVAE_ERR_SHAPE;
    if ((dims->D & 31) || (dims->Z & 31)) return PSVAE_ERR_SHAPE;  // contracted dims must be whole 32-k chunks
    for (int i = 0; i < PSVAE_NPARAM; ++i)
        if (!params->p[i] || (bwd && !grads->g[i])) return PSVAE_ERR_NULL;
    const bool has_cond = cond && cond->tab[0];
    if (has_cond && (!cond->gender || !cond->age || !cond->tab[1] || !cond->tab[2])) return PSVAE_ERR_NULL;
    if (has_cond && bwd && (!cond->gtab[0] || !cond->gtab[1] || !cond->gtab[2])) return PSVAE_ERR_NULL;
    const int B = (int)dims->B, D = (int)dims->D, Z = (int)dims->Z;
    const int Bp = (B + 31) / 32 * 32;
    hipStream_t s = (hipStream_t)stream;
    const float* const* P = params->p;
    float* const* G = bwd ? grads->g : nullptr;
    BfScratch sc = carve_bf(scratch, B, D, Z);
    int wo[9], wi[9];
    weight_dims(D, Z, wo, wi);

    int n_recon = sc.n_recon_max;  // backward-only call: every slot (the forward-only call zeroed the unused ones)
    // B >= 64 on the chain shapes (chain_fwd_ok; PSVAE_BF16_NO_CHAIN / PSVAE_BF16_CHAIN_MIN move it): each 3-layer MLP as
    // ONE launch over 32/64-row tiles (chain_bf16.hip) -- encoders, reparameterisation, decoder.  The switch is read from
    // the environment per call (tests flip it in-process); ONE decision serves both halves of this call, and a
    // backward-only call checks it against what the forward-only call on the same scratch took (the two halves share
    // activation and pack buffers that only the matching path writes): a mismatch is PSVAE_ERR_STATE, not garbage.
    const bool chain = chain_fwd_ok(B, D, Z) && chain_fwd_ok(B, Z, D);
    if (fwd) t_fwd_path = {scratch, B, D, Z, chain};
    else if (t_fwd_path.scratch == scratch && t_fwd_path.B == B && t_fwd_path.D == D && t_fwd_path.Z == Z &&
             t_fwd_path.chain != chain)
        return PSVAE_ERR_STATE;
    if ((normalize || cos_loss) && !chain) return PSVAE_ERR_SHAPE;  // the loss variants exist on the row-chain path only
    if (fwd) {
    if (chain) {  // (the chains read fp32 x and their own weight packs; x -> bf16 and the one-hot matrix, which only the weight
                  //  gradients read, are written by the first encoder chain as the rows pass through it)
    } else {  // fp32 master weights -> bf16 in both layouts (one launch, 9 tensors); x likewise
        const float* src[10]; u16* dst[10]; u16* dstT[10]; int R[10], C[10], ld[10], ldt[10];
        for (int k = 0; k < 9; ++k) {
            src[k] = P[2 * k]; dst[k] = sc.Wb[k]; dstT[k] = sc.WbT[k];
            R[k] = wo[k]; C[k] = wi[k]; ld[k] = wi[k]; ldt[k] = wo[k];
        }
        src[9] = x; dst[9] = sc.xb; dstT[9] = nullptr; R[9] = B; C[9] = D; ld[9] = D; ldt[9] = 0;
        // (+ the one-hot label matrix for the conditional-table gradients, as rider blocks)
        LAUNCH(launch_cast_transpose(src, dst, dstT, R, C, ld, ldt, 10, s, has_cond ? cond->gender : nullptr,
                                     has_cond ? cond->age : nullptr, has_cond ? sc.oh : nullptr, B));
    }
    if (chain) {
        // one launch packs W for the forward chains and W^T (all but the two input layers) for the backward chains -- unless the
        // caller keeps the packs in `scratch` current itself (psvae_pack_params_bf16 after every parameter update)
        if (!packs_current) LAUNCH(chain_pack_weights(sc, P, wo, wi, s));
        auto mlp = [&](const float* in, int K0, int NOUT, int w0, u16* s0, u16* s1, float* out, const float* ctab) {
            ChainArgs c{};
            c.in = in; c.B = B; c.K0 = K0; c.NOUT = NOUT;
            for (int l = 0; l < 3; ++l) { c.wp[l] = sc.Wc[w0 + l]; c.bias[l] = P[2 * (w0 + l) + 1]; }
            if (ctab) { c.ctab = ctab; c.gender = cond->gender; c.age = cond->age; }
            c.save[0] = s0; c.save[1] = s1; c.out = out;
            return c;
        };
        {
            ChainBatch cb{};
            cb.c[0] = mlp(x, D, Z, 0, sc.h1m, sc.h2m, mu, has_cond ? cond->tab[0] : nullptr);
            cb.c[0].xb_out = sc.xb;  // (the weight gradients' operands: x in bf16, the one-hot label rows)
            cb.c[0].oh_out = has_cond ? sc.oh : nullptr;
            cb.c[1] = mlp(x, D, Z, 3, sc.h1s, sc.h2s, logvar, has_cond ? cond->tab[1] : nullptr);
            cb.n = 2;
            LAUNCH(launch_chain_fwd_bf16(cb, s));
        }
        if (!bwd) RET_IF(hipMemsetAsync(sc.recon_partials, 0, (size_t)sc.n_recon_max * sizeof(float), s));
        {   // decoder chain; its input stage is the reparameterisation (eps, z, z in bf16, KL partial per tile)
            ChainBatch cb{};
            cb.c[0] = mlp(nullptr, Z, D, 6, sc.d1, sc.d2, x_hat, has_cond ? cond->tab[2] : nullptr);
            cb.c[0].mu = mu; cb.c[0].lv = logvar; cb.c[0].eps_in = eps_in; cb.c[0].rng = eps_in ? nullptr : rng_state;
            cb.c[0].seed = seed; cb.c[0].offset = offset; cb.c[0].eps_out = eps_out; cb.c[0].z_out = z_out; cb.c[0].zb = sc.zb;
            cb.c[0].kl_partials = sc.kl_partials;
            cb.c[0].target = x; cb.c[0].go = grad_loss; cb.c[0].scale = 2.0f / (10.0f * (float)B * (float)D);
            cb.c[0].gy = sc.gy; cb.c[0].partials = sc.recon_partials;
            cb.n = 1;
            if (normalize || cos_loss) {
                // loss variants (model.py:60-61, lightning.py:110-111): the chain leaves the raw decoder output, the row kernel
                // of the fp32 fused step (bottleneck.hip) normalises, takes the row's loss term and the gradient w.r.t. the raw
                // output -- emitted as bf16 here
                cb.c[0].raw_out = 1;
                if (normalize) cb.c[0].out = sc.y_raw;
                LAUNCH(launch_chain_fwd_bf16(cb, s));
                LAUNCH(launch_recon_row(normalize ? sc.y_raw : x_hat, x, x_hat, nullptr, nullptr, sc.recon_partials, grad_loss,
                                        normalize, cos_loss, B, D, s, sc.gy));
                n_recon = recon_row_blocks(B);
            } else {
                LAUNCH(launch_chain_fwd_bf16(cb, s));
                n_recon = (B + chain_rows_per_tile(B, 1) - 1) / chain_rows_per_tile(B, 1);
            }
            if (n_recon > sc.n_recon_max) return PSVAE_ERR_STATE;
        }
    } else {
    auto hidden = [&](const u16* A, int K, int wk, u16* out, const float* ctab) {
        GemmDescBf d = bdesc(B, H, K, A, K, sc.Wb[wk], K);
        d.bias = P[2 * wk + 1]; d.epi = BEPI_RELU; d.Cb = out; d.ldcb = H;
        if (ctab) { d.ctab = ctab; d.gender = cond->gender; d.age = cond->age; }
        return d;
    };
    {  // encoder layer 1 (model.py:15,23)
        GemmBatchBf b{};
        addb(b, hidden(sc.xb, D, 0, sc.h1m, has_cond ? cond->tab[0] : nullptr));
        addb(b, hidden(sc.xb, D, 3, sc.h1s, has_cond ? cond->tab[1] : nullptr));
        LAUNCH(launch_gemm_batch_bf16(b, s));
    }
    {
        GemmBatchBf b{};
        addb(b, hidden(sc.h1m, H, 1, sc.h2m, nullptr));
        addb(b, hidden(sc.h1s, H, 4, sc.h2s, nullptr));
        LAUNCH(launch_gemm_batch_bf16(b, s));
    }
    {  // heads: fp32 outputs
        GemmBatchBf b{};
        GemmDescBf dm = bdesc(B, Z, H, sc.h2m, H, sc.Wb[2], H);
        dm.bias = P[5]; dm.Cf = mu; dm.ldcf = Z;
        GemmDescBf ds = bdesc(B, Z, H, sc.h2s, H, sc.Wb[5], H);
        ds.bias = P[11]; ds.Cf = logvar; ds.ldcf = Z;
        addb(b, dm); addb(b, ds);
        LAUNCH(launch_gemm_batch_bf16(b, s));
    }
    // reparameterisation + KL partials; the kernel also writes z as bf16 (no cast launch)
    LAUNCH(launch_reparam_kl(mu, logvar, eps_in, rng_state, seed, offset, eps_out, z_out, sc.kl_partials,
                             (int64_t)B * Z, s, sc.zb));
    {
        GemmBatchBf b{};
        addb(b, hidden(sc.zb, Z, 6, sc.d1, has_cond ? cond->tab[2] : nullptr));
        LAUNCH(launch_gemm_batch_bf16(b, s));
    }
    {
        GemmBatchBf b{};
        addb(b, hidden(sc.d1, H, 7, sc.d2, nullptr));
        LAUNCH(launch_gemm_batch_bf16(b, s));
    }
    if (!bwd)  // forward-only call: the backward call sums ALL partial slots, so the unused ones must be zero
        RET_IF(hipMemsetAsync(sc.recon_partials, 0, (size_t)sc.n_recon_max * sizeof(float), s));
    {  // decoder output (fp32) + MSE gradient (bf16) + MSE partial sums
        GemmBatchBf b{};
        GemmDescBf d = bdesc(B, D, H, sc.d2, H, sc.Wb[8], H);
        d.bias = P[17]; d.Cf = x_hat; d.ldcf = D; d.epi = BEPI_RECON; d.target = x; d.go = grad_loss;
        d.scale = 2.0f / (10.0f * (float)B * (float)D);
        d.Cb = sc.gy; d.ldcb = D; d.partials = sc.recon_partials;
        addb(b, d);
        LAUNCH(launch_gemm_batch_bf16(b, s));
        n_recon = b.d[0].tiles_m * b.d[0].tiles_n;
        if (n_recon > sc.n_recon_max) return PSVAE_ERR_STATE;
    }
    }  // per-layer forward
    }
    if (!bwd) return 0;
    // backward: wgrad  dW[o][i] = sum_b dY[b][o] X[b][i]   (A = dY [B,out], B = X [B,in]: both batch-major, read transposed
    //                                                       out of the staged LDS tiles; fp32 out, colsum = db)
    //           dgrad  dX[b][i] = sum_o dY[b][o] W^T[i][o]  (A = dY [B,out], B = W^T [in,out], masked, bf16 out)
    const bool chain_bwd = chain;  // the forward's decision (see above)
    int S = bf_wgrad_split(B);
    if (chain_bwd && !getenv("PSVAE_BF16_WGRAD_S")) {
        // all weight gradients share ONE launch there: cut the batch so that its 128x128 tiles make two rounds of workgroups
        // (B = 4096, conditional: 96 tiles; S = 4 / 5 / 6 / 8 measured 133.5 / 130.7 / 139.3 / 142.2 us per step)
        int T = 0;
        for (int k = 0; k < 9; ++k) T += ((wo[k] + 127) / 128) * ((wi[k] + 127) / 128);
        if (has_cond) T += 3 * ((H + 127) / 128);
        S = 512 / T;
        S = S < 1 ? 1 : (S > BF_SMAX ? BF_SMAX : S);
        while (S > 1 && B / S < 256) --S;  // (at least 256 batch rows per slice)
    }
    const float* rsrc[24]; float* rdst[24]; int64_t rnum[24]; int rS[24]; int nred = 0;
    auto wgrad = [&](const u16* dY, int out_, int ldy, const u16* X, int in_, float* dW, float* db, float* slabW, float* slabB) {
        GemmDescBf d = bdesc(out_, in_, B, dY, ldy, X, in_);
        d.tr = 1;
        d.ldcf = in_;
        if (S > 1) {
            d.ksplit = S;
            d.Cf = slabW; d.colsum = db ? slabB : nullptr;
            rsrc[nred] = slabW; rdst[nred] = dW; rnum[nred] = (int64_t)out_ * in_; rS[nred++] = S;
            if (db) { rsrc[nred] = slabB; rdst[nred] = db; rnum[nred] = out_; rS[nred++] = S; }
        } else {
            d.Cf = dW; d.colsum = db;
        }
        return d;
    };
    auto dgrad = [&](const u16* dY, int out_, int wk, int in_, u16* dX, const u16* act) {
        GemmDescBf d = bdesc(B, in_, out_, dY, out_, sc.WbT[wk], out_);
        d.Cb = dX; d.ldcb = in_;
        if (act) { d.epi = BEPI_MASK; d.mask = act; d.ldmask = in_; }
        return d;
    };
    if (chain_bwd) {
        // every input gradient by two chain launches (decoder; then, behind the reparameterisation's backward, both encoders),
        // then ALL weight gradients -- nine matrices and the three tables -- as one launch, and the slab sums
        {
            ChainBwdBatch cb{};
            ChainBwdArgs& c = cb.c[0];
            c.in = sc.gy; c.in_bf16 = 1; c.B = B; c.K0 = D; c.NOUT = Z;
            c.wp[0] = sc.WcT[8]; c.wp[1] = sc.WcT[7]; c.wp[2] = sc.WcT[6];
            c.mask[0] = sc.d2; c.mask[1] = sc.d1; c.save[0] = sc.dd2; c.save[1] = sc.dd1; c.out = nullptr;
            // g_z -> g_mu, g_logvar (fp32 for the encoder chains, bf16 for the heads' weight gradients), loss, Philox advance
            c.mu = mu; c.lv = logvar; c.eps = eps_out; c.go = grad_loss; c.beta = beta;
            c.g_mu = sc.gmu_f; c.g_lv = sc.glv_f; c.gmu_b = sc.gmu; c.glv_b = sc.glv;
            c.kl_partials = sc.kl_partials; c.recon_partials = sc.recon_partials;
            c.n_kl = (B + chain_rows_per_tile(B, 1) - 1) / chain_rows_per_tile(B, 1); c.n_recon = n_recon;
            c.recon_mode = cos_loss ? 1 : 0;
            c.loss3 = loss3; c.rng = eps_in ? nullptr : rng_state; c.nblk = (uint64_t)(((int64_t)B * Z + 3) / 4);
            cb.n = 1;
            LAUNCH(launch_chain_bwd_bf16(cb, s));
        }
        {
            ChainBwdBatch cb{};
            for (int e = 0; e < 2; ++e) {
                ChainBwdArgs& c = cb.c[e];
                c.in = e ? sc.glv_f : sc.gmu_f; c.in_bf16 = 0; c.B = B; c.K0 = Z; c.NOUT = 0;
                c.wp[0] = sc.WcT[e ? 5 : 2]; c.wp[1] = sc.WcT[e ? 4 : 1]; c.wp[2] = nullptr;
                c.mask[0] = e ? sc.h2s : sc.h2m; c.mask[1] = e ? sc.h1s : sc.h1m;
                c.save[0] = e ? sc.dh2s : sc.dh2m; c.save[1] = e ? sc.dh1s : sc.dh1m; c.out = nullptr;
            }
            cb.n = 2;
            LAUNCH(launch_chain_bwd_bf16(cb, s));
        }
        {
            GemmBatchBf b{};
            // (order = the launch's work list inside one batch slice: problems that read the same array are neighbours, so
            // that they run under the same L2 -- a table gradient next to the layer whose dY it shares, the two first
            // encoder layers, which share x: wgrad_bf16.hip)
            addb(b, wgrad(sc.gy, D, D, sc.d2, H, G[16], G[17], sc.slab_w[8], sc.slab_b[8]));
            addb(b, wgrad(sc.dd2, H, H, sc.d1, H, G[14], G[15], sc.slab_w[7], sc.slab_b[7]));
            addb(b, wgrad(sc.dd1, H, H, sc.zb, Z, G[12], G[13], sc.slab_w[6], sc.slab_b[6]));
            if (has_cond) addb(b, wgrad(sc.oh, PSVAE_COND_DIM, 8, sc.dd1, H, cond->gtab[2], nullptr, sc.slab_t[2], nullptr));
            addb(b, wgrad(sc.gmu, Z, Z, sc.h2m, H, G[4], G[5], sc.slab_w[2], sc.slab_b[2]));
            addb(b, wgrad(sc.dh2m, H, H, sc.h1m, H, G[2], G[3], sc.slab_w[1], sc.slab_b[1]));
            addb(b, wgrad(sc.glv, Z, Z, sc.h2s, H, G[10], G[11], sc.slab_w[5], sc.slab_b[5]));
            addb(b, wgrad(sc.dh2s, H, H, sc.h1s, H, G[8], G[9], sc.slab_w[4], sc.slab_b[4]));
            addb(b, wgrad(sc.dh1m, H, H, sc.xb, D, G[0], G[1], sc.slab_w[0], sc.slab_b[0]));
            if (has_cond) addb(b, wgrad(sc.oh, PSVAE_COND_DIM, 8, sc.dh1m, H, cond->gtab[0], nullptr, sc.slab_t[0], nullptr));
            addb(b, wgrad(sc.dh1s, H, H, sc.xb, D, G[6], G[7], sc.slab_w[3], sc.slab_b[3]));
            if (has_cond) addb(b, wgrad(sc.oh, PSVAE_COND_DIM, 8, sc.dh1s, H, cond->gtab[1], nullptr, sc.slab_t[1], nullptr));
            b.force_staged = 1;
            static const bool old_wgrad = getenv("PSVAE_WGRAD_V1") != nullptr && atoi(getenv("PSVAE_WGRAD_V1")) != 0;
            if (old_wgrad) LAUNCH(launch_gemm_batch_bf16(b, s));  // round 2's register-staged kernel (A/B runs)
            else LAUNCH(launch_wgrad_batch_bf16(b, s));
        }
        if (nred) LAUNCH(launch_slab_reduce(rsrc, rdst, rnum, rS, nred, s));
        return 0;
    }
    {
        GemmBatchBf b{};
        addb(b, wgrad(sc.gy, D, D, sc.d2, H, G[16], G[17], sc.slab_w[8], sc.slab_b[8]));
        addb(b, dgrad(sc.gy, D, 8, H, sc.dd2, sc.d2));
        b.force_staged = 1;
        LAUNCH(launch_gemm_batch_bf16(b, s));
    }
    {
        GemmBatchBf b{};
        addb(b, wgrad(sc.dd2, H, H, sc.d1, H, G[14], G[15], sc.slab_w[7], sc.slab_b[7]));
        addb(b, dgrad(sc.dd2, H, 7, H, sc.dd1, sc.d1));
        b.force_staged = 1;
        LAUNCH(launch_gemm_batch_bf16(b, s));
    }
    {
        GemmBatchBf b{};
        addb(b, wgrad(sc.dd1, H, H, sc.zb, Z, G[12], G[13], sc.slab_w[6], sc.slab_b[6]));
        if (has_cond) addb(b, wgrad(sc.oh, PSVAE_COND_DIM, 8, sc.dd1, H, cond->gtab[2], nullptr, sc.slab_t[2], nullptr));  // table grad
        GemmDescBf gz = bdesc(B, Z, H, sc.dd1, H, sc.WbT[6], H);  // g_z in fp32
        gz.Cf = sc.gz; gz.ldcf = Z;
        addb(b, gz);
        b.force_staged = 1;
        LAUNCH(launch_gemm_batch_bf16(b, s));
    }
    {   // g_mu / g_logvar (+ loss finalize, rng advance); the kernel also writes their bf16 copies
        u16* rm[2] = {sc.gmu, sc.glv};
        LAUNCH(launch_reparam_bwd_final(sc.gz, mu, logvar, eps_out, beta, grad_loss, sc.gmu_f, sc.glv_f, B, Z, D,
                                        sc.kl_partials, reparam_blocks((int64_t)B * Z), sc.recon_partials, n_recon, loss3,
                                        eps_in ? nullptr : rng_state, s, rm));
    }
    {
        GemmBatchBf b{};
        addb(b, wgrad(sc.gmu, Z, Z, sc.h2m, H, G[4], G[5], sc.slab_w[2], sc.slab_b[2]));
        addb(b, wgrad(sc.glv, Z, Z, sc.h2s, H, G[10], G[11], sc.slab_w[5], sc.slab_b[5]));
        addb(b, dgrad(sc.gmu, Z, 2, H, sc.dh2m, sc.h2m));
        addb(b, dgrad(sc.glv, Z, 5, H, sc.dh2s, sc.h2s));
        b.force_staged = 1;
        LAUNCH(launch_gemm_batch_bf16(b, s));
    }
    {
        GemmBatchBf b{};
        addb(b, wgrad(sc.dh2m, H, H, sc.h1m, H, G[2], G[3], sc.slab_w[1], sc.slab_b[1]));
        addb(b, wgrad(sc.dh2s, H, H, sc.h1s, H, G[8], G[9], sc.slab_w[4], sc.slab_b[4]));
        addb(b, dgrad(sc.dh2m, H, 1, H, sc.dh1m, sc.h1m));
        addb(b, dgrad(sc.dh2s, H, 4, H, sc.dh1s, sc.h1s));
        b.force_staged = 1;
        LAUNCH(launch_gemm_batch_bf16(b, s));
    }
    {
        GemmBatchBf b{};
        addb(b, wgrad(sc.dh1m, H, H, sc.xb, D, G[0], G[1], sc.slab_w[0], sc.slab_b[0]));
        addb(b, wgrad(sc.dh1s, H, H, sc.xb, D, G[6], G[7], sc.slab_w[3], sc.slab_b[3]));
        if (has_cond) {  // gtab[c][n] = sum_b onehot[b][c] dh1[b][n]: two more wgrad problems in this launch
            addb(b, wgrad(sc.oh, PSVAE_COND_DIM, 8, sc.dh1m, H, cond->gtab[0], nullptr, sc.slab_t[0], nullptr));
            addb(b, wgrad(sc.oh, PSVAE_COND_DIM, 8, sc.dh1s, H, cond->gtab[1], nullptr, sc.slab_t[1], nullptr));
        }
        b.force_staged = 1;
        LAUNCH(launch_gemm_batch_bf16(b, s));
    }
    if (nred) LAUNCH(launch_slab_reduce(rsrc, rdst, rnum, rS, nred, s));
    return 0;
}

#undef LAUNCH

int psvae_last_step_launches_bf16(void) { return t_bf_launches; }

// ---- the weight-gradient launch of the bf16 pass as an entry of its own (csrc/wgrad_bf16.hip): dW = dY^T X, db = column
// sums of dY.  What autograd computes for an nn.Linear's .weight.grad / .bias.grad (model.py:14-36) from bf16 operands;
// exposed so that the kernel can be pinned by itself at full batch sizes (tests/test_gpu_wgrad.py: integer-valued operands
// make every fp32 sum exact, so the result must equal an integer matrix product bit for bit whatever the batch split).
size_t psvae_wgrad_bf16_scratch_bytes(int out, int in, int split) {
    if (out <= 0 || in <= 0 || split < 1 || split > BF_SMAX) return 0;
    return split > 1 ? (size_t)split * ((size_t)out * in + out) * sizeof(float) : 0;
}

int psvae_wgrad_bf16(const uint16_t* dY, int64_t ldy, const uint16_t* X, int64_t ldx, int64_t B, int out, int in, float* dW,
                     float* db, int split, void* scratch, void* stream) {
    if (!dY || !X || !dW) return PSVAE_ERR_NULL;
    if (B <= 0 || B > INT32_MAX || out <= 0 || in <= 0 || ldy < out || ldx < in) return PSVAE_ERR_SHAPE;
    if (split < 1 || split > BF_SMAX) return PSVAE_ERR_ARG;
    if ((ldy & 7) || (ldx & 7) || (reinterpret_cast<uintptr_t>(dY) & 15) || (reinterpret_cast<uintptr_t>(X) & 15)) return PSVAE_ERR_SHAPE;
    if (split > 1 && !scratch) return PSVAE_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    GemmBatchBf b{};
    GemmDescBf d = bdesc(out, in, (int)B, dY, (int)ldy, X, (int)ldx);
    d.tr = 1;
    d.ldcf = in;
    float* slabW = reinterpret_cast<float*>(scratch);
    float* slabB = slabW ? slabW + (size_t)split * out * in : nullptr;
    if (split > 1) {
        d.ksplit = split;
        d.Cf = slabW;
        d.colsum = db ? slabB : nullptr;
    } else {
        d.Cf = dW;
        d.colsum = db;
    }
    addb(b, d);
    RET_IF(launch_wgrad_batch_bf16(b, s));
    if (split > 1) {
        const float* rsrc[2] = {slabW, slabB}; float* rdst[2] = {dW, db}; int64_t rnum[2] = {(int64_t)out * in, out}; int rS[2] = {split, split};
        RET_IF(launch_slab_reduce(rsrc, rdst, rnum, rS, db ? 2 : 1, s));
    }
    return PSVAE_OK;
}

int psvae_train_step_bf16(const psvae_dims* dims, const float* x, const psvae_params* params, const psvae_cond* cond,
                          const float* eps_in, uint64_t seed, uint64_t offset, uint64_t* rng_state, float* eps_out,
                          float* z_out, float* x_hat, float* mu, float* logvar, float beta, const float* grad_loss,
                          const psvae_grads* grads, float* loss3, void* scratch, uint32_t flags, void* stream) {
    return bf16_pass(3, dims, x, params, cond, eps_in, seed, offset, rng_state, eps_out, z_out, x_hat, mu, logvar, beta,
                     grad_loss, grads, loss3, scratch, flags, stream);
}

// The two halves as separate calls (validation runs only the first).  The forward leaves the bf16 activations, the MSE
// gradient (already times *grad_loss) and the loss partial sums in `scratch`; the backward call must get the same
// scratch, x, params, cond, eps_out, mu, logvar untouched.  The Philox state advances in the backward call.
int psvae_forward_bf16(const psvae_dims* dims, const float* x, const psvae_params* params, const psvae_cond* cond,
                       const float* eps_in, uint64_t seed, uint64_t offset, const uint64_t* rng_state, float* eps_out,
                       float* z_out, float* x_hat, float* mu, float* logvar, const float* grad_loss, void* scratch,
                       uint32_t flags, void* stream) {
    return bf16_pass(1, dims, x, params, cond, eps_in, seed, offset, const_cast<uint64_t*>(rng_state), eps_out, z_out, x_hat,
                     mu, logvar, 1.0f, grad_loss, nullptr, nullptr, scratch, flags, stream);
}

int psvae_backward_bf16(const psvae_dims* dims, const float* x, const psvae_params* params, const psvae_cond* cond,
                        uint64_t* rng_state, float* eps_out, float* z_out, float* x_hat, float* mu, float* logvar,
                        float beta, const float* grad_loss, const psvae_grads* grads, float* loss3, void* scratch,
                        uint32_t flags, void* stream) {
    // eps_in = eps_out marks "epsilon was given": the backward never draws; rng_state (nullable) is advanced
    return bf16_pass(2, dims, x, params, cond, nullptr, 0, 0, rng_state, eps_out, z_out, x_hat, mu, logvar, beta, grad_loss,
                     grads, loss3, scratch, flags, stream);
}

// ---- decode-only sampling in bf16 (VAEModel.decode, model.py:65-69; SURVEY 8b psvae_decode_bf16) ----------------
// Every GEMM reads its activation row-major, so this path needs no second layout: z and the three decoder weights are
// cast to bf16 (one launch per call / per chunk), the hidden activations stay bf16, biases, accumulation and x_hat fp32.
// rows per pass (PSVAE_DECODE_CHUNK_BF: tuning; a multiple of 128)
static int64_t decode_chunk_bf() {
    static const int64_t v = [] {
        const char* e = getenv("PSVAE_DECODE_CHUNK_BF");
        const long long n = e ? atoll(e) : 0;
        return (int64_t)((n >= 128 && n % 128 == 0) ? n : 131072);
    }();
    return v;
}
#define DECODE_CHUNK_BF decode_chunk_bf()

size_t psvae_decode_bf16_scratch_bytes(int64_t N, int64_t D, int64_t Z) {
    if (N <= 0 || D <= 0 || Z <= 0) return 0;
    const int64_t c = N < DECODE_CHUNK_BF ? N : DECODE_CHUNK_BF;
    return align_up((size_t)(H * Z + H * H + D * H) * 2, 256) + align_up((size_t)c * Z * 2, 256) +
           2 * align_up((size_t)c * H * 2, 256) + align_up((size_t)c * sizeof(float), 256);
}

int psvae_decode_bf16(int64_t N, int64_t D, int64_t Z, const float* z, const float* const* dec6, float* x_hat,
                      void* scratch, uint32_t flags, void* stream) {
    if (N <= 0 || D <= 0 || Z <= 0) return PSVAE_ERR_SHAPE;
    if ((Z & 31) || (D & 3)) return PSVAE_ERR_SHAPE;  // contracted dims are whole 32-k chunks; 16-B fp32 output rows
    if (!z || !dec6 || !x_hat || !scratch) return PSVAE_ERR_NULL;
    for (int i = 0; i < 6; ++i)
        if (!dec6[i]) return PSVAE_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t c = N < DECODE_CHUNK_BF ? N : DECODE_CHUNK_BF;
    char* base = reinterpret_cast<char*>(scratch);
    u16* Wb0 = reinterpret_cast<u16*>(base);            // [H, Z]
    u16* Wb1 = Wb0 + (size_t)H * Z;                     // [H, H]
    u16* Wb2 = Wb1 + (size_t)H * H;                     // [D, H]
    size_t off = align_up((size_t)(H * Z + H * H + D * H) * 2, 256);
    u16* zb = reinterpret_cast<u16*>(base + off); off += align_up((size_t)c * Z * 2, 256);
    u16* d1 = reinterpret_cast<u16*>(base + off); off += align_up((size_t)c * H * 2, 256);
    u16* d2 = reinterpret_cast<u16*>(base + off); off += align_up((size_t)c * H * 2, 256);
    float* yn = reinterpret_cast<float*>(base + off);
    const bool normalize = flags & PSVAE_NORMALIZE_DECODER;
    if (decode_fused_bf16_ok(N, (int)Z, (int)D)) {
        // all three layers in one kernel (decode_fused.hip); the packed bf16 weights take the place of Wb0..Wb2
        RET_IF(launch_decode_fused_bf16(N, (int)Z, (int)D, z, dec6, x_hat, Wb0, s));
        if (normalize)
            for (int64_t r0 = 0; r0 < N; r0 += c) {
                const int n = (int)((N - r0) < c ? (N - r0) : c);
                RET_IF(launch_normalize_rows(x_hat + r0 * D, x_hat + r0 * D, yn, n, (int)D, s));
            }
        return 0;
    }
    for (int64_t r0 = 0; r0 < N; r0 += c) {
        const int n = (int)((N - r0) < c ? (N - r0) : c);
        {   // casts: the chunk of z, and (first chunk only) the three weights
            const float* src[4] = {z + r0 * Z, dec6[0], dec6[2], dec6[4]};
            u16* dst[4] = {zb, Wb0, Wb1, Wb2};
            u16* dstT[4] = {nullptr, nullptr, nullptr, nullptr};
            int R[4] = {n, H, H, (int)D}, C[4] = {(int)Z, (int)Z, H, H}, ld[4] = {(int)Z, (int)Z, H, H}, ldt[4] = {0, 0, 0, 0};
            RET_IF(launch_cast_transpose(src, dst, dstT, R, C, ld, ldt, r0 == 0 ? 4 : 1, s));
        }
        {
            GemmBatchBf b{};
            GemmDescBf d = bdesc(n, H, (int)Z, zb, (int)Z, Wb0, (int)Z);
            d.bias = dec6[1]; d.epi = BEPI_RELU; d.Cb = d1; d.ldcb = H;
            addb(b, d);
            RET_IF(launch_gemm_batch_bf16(b, s));
        }
        {
            GemmBatchBf b{};
            GemmDescBf d = bdesc(n, H, H, d1, H, Wb1, H);
            d.bias = dec6[3]; d.epi = BEPI_RELU; d.Cb = d2; d.ldcb = H;
            addb(b, d);
            RET_IF(launch_gemm_batch_bf16(b, s));
        }
        {
            GemmBatchBf b{};
            GemmDescBf d = bdesc(n, (int)D, H, d2, H, Wb2, H);
            d.bias = dec6[5]; d.Cf = x_hat + r0 * D; d.ldcf = (int)D;
            addb(b, d);
            RET_IF(launch_gemm_batch_bf16(b, s));
        }
        if (normalize) RET_IF(launch_normalize_rows(x_hat + r0 * D, x_hat + r0 * D, yn, n, (int)D, s));
    }
    return 0;
}

// psvae_sample_decode_f32 with bf16 operands (see there): on the fused decode's shapes z is drawn inside the decode kernel
int psvae_sample_decode_bf16(int64_t N, int64_t D, int64_t Z, uint64_t seed, uint64_t offset, const float* const* dec6,
                             float* z_out, float* x_hat, void* scratch, uint32_t flags, void* stream) {
    if (N <= 0 || D <= 0 || Z <= 0) return PSVAE_ERR_SHAPE;
    if ((Z & 31) || (D & 3)) return PSVAE_ERR_SHAPE;
    if (!dec6 || !x_hat || !scratch) return PSVAE_ERR_NULL;
    for (int i = 0; i < 6; ++i)
        if (!dec6[i]) return PSVAE_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    if (decode_fused_bf16_ok(N, (int)Z, (int)D)) {
        const int64_t c = N < DECODE_CHUNK_BF ? N : DECODE_CHUNK_BF;
        char* base = reinterpret_cast<char*>(scratch);
        u16* Wb0 = reinterpret_cast<u16*>(base);
        size_t off = align_up((size_t)(H * Z + H * H + D * H) * 2, 256);
        off += align_up((size_t)c * Z * 2, 256) + 2 * align_up((size_t)c * H * 2, 256);
        float* yn = reinterpret_cast<float*>(base + off);
        RET_IF(launch_decode_fused_bf16(N, (int)Z, (int)D, nullptr, dec6, x_hat, Wb0, s, seed, offset, z_out));
        if (flags & PSVAE_NORMALIZE_DECODER)
            for (int64_t r0 = 0; r0 < N; r0 += c) {
                const int n = (int)((N - r0) < c ? (N - r0) : c);
                RET_IF(launch_normalize_rows(x_hat + r0 * D, x_hat + r0 * D, yn, n, (int)D, s));
            }
        return 0;
    }
    if (!z_out) return PSVAE_ERR_NULL;
    RET_IF(launch_philox_normal(z_out, N * Z, seed, offset, s));
    return psvae_decode_bf16(N, D, Z, z_out, dec6, x_hat, scratch, flags, stream);
}

}  // extern "C"
