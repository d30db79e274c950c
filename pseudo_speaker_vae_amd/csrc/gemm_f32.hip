// Grouped fp32 MFMA GEMM for gfx950 (MI355X): the building block of every Linear layer of the VAE
// (forward NT, dgrad NN, wgrad TN; reference: nn.Linear at /root/reference/ps_vae/model.py:14-36 and
// its autograd).  Exact fp32: v_mfma_f32_16x16x4_f32, which is a k-ordered fmaf chain.
//
// Design -- "register-direct", chosen by measurement (tools/microbench/rd_gemm.hip):
// the VAE's GEMMs are small (<= 1 GFLOP, K <= 1024) and fp32 MFMA is slow enough (64 FLOP/clk/SIMD)
// that L1/L2 can feed the matrix pipe directly, while LDS staging (ds_write + barrier per k tile)
// cost as much as the MFMAs themselves at the small tiles needed to fill 256 CUs.  So:
//   * no LDS and no barrier in the k loop: every wave loads its own MFMA fragments global -> VGPR,
//     16 B per lane, D chunks of 16 k in flight (the 512-entry register file is the prefetch buffer);
//       "KC" operand (k contiguous: activations, W[out,in] in NT, dY in NN): lane (r,q) loads k =
//            16c+4q..+3 of row 16b+r; the 4 values feed 4 consecutive MFMAs (the same k permutation
//            is applied to both operands, so the sum is unchanged);
//       "KM" operand (k-major: W in NN, both operands of the batch-contracting wgrad): lane (r,q)
//            loads T adjacent rows of k = 16c+4q+t -- whole 64..256-B row segments per k;
//   * one workgroup = 4 waves = WM x WN spatial waves x KS k-slices (split-K INSIDE the workgroup):
//     a 1024-row batch still gives >= 1 wave per SIMD, and the long-K wgrad (K = batch) is cut in 4;
//     partial tiles meet in LDS once, after the loop, where the k-slices are summed in fixed order
//     (bitwise reproducible) and the tile leaves as whole rows, 16 B per lane, with the fused
//     epilogue (+bias, ReLU, ReLU-mask, accumulate) applied on the way out;
//   * one launch = a GROUP of independent problems (both encoders; a layer's wgrad + dgrad);
//   * bias gradients (column sums of dY) come from the A fragments already in registers.
#include <string.h>

#include <atomic>
#include <mutex>

#include "psvae_common.h"

namespace psvae {

constexpr int NTHR = 256;

template <int T> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<2> { typedef f32x2 type; };
template <> struct VecT<4> { typedef f32x4 type; };
template <int T> __device__ __forceinline__ float vget(const typename VecT<T>::type& v, int i);
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }
template <> __device__ __forceinline__ float vget<2>(const f32x2& v, int i) { return v[i]; }
template <> __device__ __forceinline__ float vget<4>(const f32x4& v, int i) { return v[i]; }

// Pointers arrive inside a by-value struct, for which hipcc falls back to FLAT accesses (flat loads
// count on vmcnt AND lgkmcnt and return out of order, so every use drains both counters and the
// prefetch pipeline collapses).  All global traffic therefore goes through explicit address-space-1
// accessors.
#define PSVAE_GLOBAL __attribute__((address_space(1)))
template <typename V>
__device__ __forceinline__ V ldg(const float* p) {
    return *reinterpret_cast<const PSVAE_GLOBAL V*>(reinterpret_cast<uintptr_t>(p));
}
template <typename V>
__device__ __forceinline__ void stg(float* p, const V& v) {
    *reinterpret_cast<PSVAE_GLOBAL V*>(reinterpret_cast<uintptr_t>(p)) = v;
}

// One operand's fragments for the 16-k chunk c: f[t][b] (t = MFMA step, b = 16-row block).
//   R = extent of the non-contracted dimension (rows are CLAMPED to R-1: they only feed outputs that
//   the epilogue never stores), K = contracted extent (out-of-range k is ZERO-filled).
//   FAST: the chunk is interior in k, the operand is 16-B aligned and -- for KM -- the wave's span of
//   16*T rows lies inside R (all wave-uniform, decided by the caller: no per-lane branch on this path).
template <int T, bool KM, bool FAST>
__device__ __forceinline__ void load_frag(float (&f)[4][T], const float* __restrict__ base, int ld, int off0, int R,
                                          int K, int c, int r, int q) {
    if (!KM) {  // memory [row][k]
#pragma unroll
        for (int b = 0; b < T; ++b) {
            int row = off0 + 16 * b + r;
            row = row < R ? row : R - 1;
            const float* p = base + (size_t)row * ld + 16 * c + 4 * q;
            if (FAST) {
                const f32x4 v = ldg<f32x4>(p);
#pragma unroll
                for (int t = 0; t < 4; ++t) f[t][b] = v[t];
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) f[t][b] = (16 * c + 4 * q + t < K) ? ldg<float>(p + t) : 0.f;
            }
        }
    } else {  // memory [k][row]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = 16 * c + 4 * q + t;
            const int col = off0 + T * r;
            if (FAST) {  // caller guarantees the wave's whole span [off0, off0 + 16T) lies inside R
                const typename VecT<T>::type v = ldg<typename VecT<T>::type>(base + (size_t)k * ld + col);
#pragma unroll
                for (int b = 0; b < T; ++b) f[t][b] = vget<T>(v, b);
            } else {
#pragma unroll
                for (int b = 0; b < T; ++b) {
                    int cc = col + b;
                    cc = cc < R ? cc : R - 1;
                    f[t][b] = (k < K) ? ldg<float>(base + (size_t)k * ld + cc) : 0.f;
                }
            }
        }
    }
}

// actual index (inside the wave's 16*T span) of MFMA slot `slot` (0..15) of block `blk`
template <int T, bool KM>
__device__ __forceinline__ int slot_index(int blk, int slot) { return KM ? T * slot + blk : 16 * blk + slot; }

template <int TM, int TN, int WM, int WN, int KS, int D, bool AKM, bool BKM, bool S2 = false>
__device__ __forceinline__ void gemm_tile(const GemmDesc& d, int tm, int tn, int ksg, float* __restrict__ smem) {
    constexpr int NW = WM * WN * KS;  // waves per workgroup: 4 or 8
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    constexpr int NT = 64 * NW;
    constexpr int BM = WM * 16 * TM, BN = WN * 16 * TN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wave / (WM * WN), ws = wave % (WM * WN);
    const int wm = ws / WN, wn = ws % WN;
    const int r = lane & 15, q = lane >> 4;
    const int M = d.M, N = d.N, K = d.K;
    const int m0 = tm * BM, n0 = tn * BN;
    const int am = m0 + wm * 16 * TM, bn = n0 + wn * 16 * TN;
    const float* __restrict__ A = d.A;
    const float* __restrict__ B = d.B;
    const int lda = d.lda, ldb = d.ldb;
    // vector loads need 16-B aligned bases and row pitches
    const bool vecA = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool vecB = ((ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    // KM vector loads additionally need the wave's whole span inside the matrix (else: guarded path)
    const bool fastA = vecA && (!AKM || am + 16 * TM <= M);
    const bool fastB = vecB && (!BKM || bn + 16 * TN <= N);
    const bool vec = fastA && fastB;

#ifdef PSVAE_STAMP  // diagnostic build only (tools/build_variant.sh stamp -DPSVAE_STAMP; tools/stamp_f32.py): wave 0 of every
                    // workgroup prints its phase lengths in shader cycles and its entry / exit on the 100 MHz clock
    unsigned long long st_[8];
    int nst_ = 0;
    const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime();
#define STAMP() do { if (nst_ < 8) st_[nst_++] = __builtin_amdgcn_s_memtime(); } while (0)
    STAMP();
#else
#define STAMP() do {} while (0)
#endif
    f32x4 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float csum[TM];
#pragma unroll
    for (int m = 0; m < TM; ++m) csum[m] = 0.f;
    const bool do_colsum = AKM && (d.colsum != nullptr) && (tn == 0) && (wn == 0);

    // chunk range: the ceil(K/16) chunks are cut over the `ksplit` workgroup slices (cross-workgroup split-K:
    // slice ksg writes a partial tile), and this workgroup's share over its KS waves
    const int nc_all = (K + 15) >> 4;
    const int per_wg = (nc_all + d.ksplit - 1) / d.ksplit;
    const int wbeg = ksg * per_wg;
    const int nc = (wbeg + per_wg < nc_all) ? wbeg + per_wg : nc_all;  // end of this workgroup's range
    const int per = (nc - wbeg + KS - 1) / KS;
    const int cbeg = wbeg + ks * (per > 0 ? per : 0);
    const int cend = (cbeg + per < nc) ? cbeg + per : nc;

    float fa[D][4][TM], fb[D][4][TN];
    auto mma = [&](int s) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][t][m], fb[s][t][n], acc[m][n], 0, 0, 0);
        }
        if (do_colsum) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < TM; ++m) csum[m] += fa[s][t][m];
        }
    };

    STAMP();  // 1: descriptor decoded, addresses ready
    int c = cbeg;
    if (vec) {
        // fast region: whole groups of D chunks that are interior in k.  Software pipeline over a ring of D
        // fragment slots: at the top of chunk i the loads of chunk i+D-1 are issued, then chunk i (whose
        // loads left D-1 chunks ago) is multiplied.  The sched_barriers pin that order: left alone, the
        // scheduler sinks every load of a group below the group's last MFMA and the next iteration stalls
        // on them at once (measured: ~35 % of wave time in s_waitcnt).
        const int cf_end = cend < (K >> 4) ? cend : (K >> 4);
        const int nfull = cf_end > cbeg ? (cf_end - cbeg) / D : 0;
        if (nfull > 0) {
#pragma unroll
            for (int s = 0; s < D - 1; ++s) {
                load_frag<TM, AKM, true>(fa[s], A, lda, am, M, K, cbeg + s, r, q);
                load_frag<TN, BKM, true>(fb[s], B, ldb, bn, N, K, cbeg + s, r, q);
            }
            for (int g = 0; g < nfull - 1; ++g) {
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    const int ls = (s + D - 1) % D;
                    load_frag<TM, AKM, true>(fa[ls], A, lda, am, M, K, cbeg + g * D + s + D - 1, r, q);
                    load_frag<TN, BKM, true>(fb[ls], B, ldb, bn, N, K, cbeg + g * D + s + D - 1, r, q);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(s);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            load_frag<TM, AKM, true>(fa[D - 1], A, lda, am, M, K, cbeg + nfull * D - 1, r, q);
            load_frag<TN, BKM, true>(fb[D - 1], B, ldb, bn, N, K, cbeg + nfull * D - 1, r, q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < D; ++s) mma(s);
            c = cbeg + nfull * D;
        }
    }
    // remainder: a partial last group (vector loads while the chunk is interior in k), a k-ragged last
    // chunk or unaligned operands (guarded scalar loads) -- one group at a time, loads then MFMAs
    const int c_int = vec ? (K >> 4) : 0;  // chunks below this are interior and aligned
    for (; c < cend; c += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            if (c + s < c_int && c + s < cend) {
                load_frag<TM, AKM, true>(fa[s], A, lda, am, M, K, c + s, r, q);
                load_frag<TN, BKM, true>(fb[s], B, ldb, bn, N, K, c + s, r, q);
            } else if (c + s < cend) {
                load_frag<TM, AKM, false>(fa[s], A, lda, am, M, K, c + s, r, q);
                load_frag<TN, BKM, false>(fb[s], B, ldb, bn, N, K, c + s, r, q);
            }
        }
#pragma unroll
        for (int s = 0; s < D; ++s)
            if (c + s < cend) mma(s);
    }

    STAMP();  // 2: k loop done (all MFMAs issued)
    // second stage (see below): its weight fragments are requested NOW, so that they land while the k-slices are being
    // reduced through LDS (the main loop's fragment ring is dead from here on: the registers are free)
    f32x4 b4[S2 ? 4 : 1][4];
    if (S2 && d.B2 != nullptr) {
        const float* __restrict__ W2 = d.B2 + (size_t)n0 * d.Z2;
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2)
#pragma unroll
            for (int t = 0; t < 4; ++t) b4[c2][t] = ldg<f32x4>(W2 + (size_t)(16 * c2 + 4 * q + t) * d.Z2 + 4 * r);
    }
    // ---- epilogue: k-slices meet in LDS ([KS][BM][BN+4]), summed in fixed order; whole rows leave ----
    constexpr int SC = BN + 4;
    float* __restrict__ Cs = smem + ks * (BM * SC);
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lr = wm * 16 * TM + slot_index<TM, AKM>(m, 4 * q + j);
#pragma unroll
            for (int n = 0; n < TN; ++n) Cs[lr * SC + wn * 16 * TN + slot_index<TN, BKM>(n, r)] = acc[m][n][j];
        }
    float* __restrict__ csum_s = smem + KS * (BM * SC);  // [KS][BM] bias-gradient partials
    if (do_colsum) {
        // lanes with equal r hold partial sums over k = 4q+t: reduce over q, then park per k-slice
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            float s = csum[m];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (q == 0) csum_s[ks * BM + wm * 16 * TM + slot_index<TM, true>(m, r)] = s;
        }
    }
    __syncthreads();
    STAMP();  // 3: every wave's partial tile is in LDS
    const int epi = d.epi;
    float* __restrict__ C = d.C + (size_t)ksg * M * d.ldc;  // ksplit > 1: this slice's slab
    const float* __restrict__ aux = d.aux;
    const int ldc = d.ldc;
    const bool vecC = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(aux) & 15) == 0);
    constexpr int CQ = BN / 4;        // float4 per tile row
    constexpr int RSTEP = NT / CQ;    // rows per pass (CQ <= 32 for every config)
    const int c4 = tid % CQ, r0 = tid / CQ;
    const int col = n0 + 4 * c4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    float sq = 0.f;  // EPI_BIAS_RECON: this thread's share of sum (x_hat - x)^2
    const float gsc = (epi == EPI_BIAS_RECON) ? d.scale * (d.go ? ldg<float>(d.go) : 1.0f) : 0.f;
    const float kb = (epi == EPI_REPARAM_BWD) ? (d.go ? ldg<float>(d.go) : 1.0f) * d.scale / (float)M : 0.f;
    if (epi == EPI_BIAS || epi == EPI_BIAS_RELU || epi == EPI_BIAS_RECON) {
        // one 16-byte load where the four columns are whole and aligned, else four loads from clamped columns: never a
        // branch per element (`if (col + e < N) load` compiles to branch + load + s_waitcnt, four serial round trips);
        // values of columns >= N are never stored
        if ((col + 3 < N) && ((reinterpret_cast<uintptr_t>(aux + col) & 15) == 0)) {
            bias4 = ldg<f32x4>(aux + col);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) bias4[e] = ldg<float>(aux + (col + e < N ? col + e : N - 1));
        }
    }
    // NOT unrolled: this code runs once per workgroup, straight from a cold instruction cache (consecutive launches of
    // a step are different kernels) -- four copies of the row body cost more in instruction fetch than the loop
    // overhead saves (config 6: 72 -> 46 KB of code; step 153.9 -> 150.8 us untuned, 146.8 -> 144.9 us tuned, same box)
#ifndef PSVAE_ROW_UNROLL
#define PSVAE_ROW_UNROLL 1
#endif
#pragma unroll PSVAE_ROW_UNROLL
    for (int rr = 0; rr < BM; rr += RSTEP) {
        const int lr = rr + r0;
        const int row = m0 + lr;
        if (lr >= BM || row >= M || col >= N) continue;
#ifdef PSVAE_DIAG_NOLDSRD  // (diagnostic builds only: wrong results, for timing what a phase costs)
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#else
        f32x4 v = *reinterpret_cast<const f32x4*>(smem + lr * SC + 4 * c4);
#pragma unroll
        for (int k2 = 1; k2 < KS; ++k2) v += *reinterpret_cast<const f32x4*>(smem + k2 * (BM * SC) + lr * SC + 4 * c4);
#endif
        const size_t o = (size_t)row * ldc + col;
        const bool full = vecC && (col + 3 < N);
        if (epi == EPI_BIAS || epi == EPI_BIAS_RELU) {
            v += bias4;
            if (d.ctab) {
                const int g = d.gender[row], a = d.age[row];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (col + e < N) {
                        if (g >= 0 && g < 3) v[e] += ldg<float>(d.ctab + (size_t)g * N + col + e);
                        if (a >= 0 && a < 3) v[e] += ldg<float>(d.ctab + (size_t)(3 + a) * N + col + e);
                    }
                }
            }
            if (epi == EPI_BIAS_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
        } else if (epi == EPI_BIAS_RECON) {
            v += bias4;
            f32x4 xt = {0.f, 0.f, 0.f, 0.f};
            if (full) {
                xt = ldg<f32x4>(d.aux2 + o);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < N) xt[e] = ldg<float>(d.aux2 + o + e);
            }
            f32x4 gy;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float df = (col + e < N) ? v[e] - xt[e] : 0.f;
                sq += df * df;
                gy[e] = gsc * df;
            }
            if (full) {
                stg<f32x4>(d.C2 + o, gy);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < N) stg<float>(d.C2 + o + e, gy[e]);
            }
        } else if (epi == EPI_REPARAM_BWD) {  // lightning.py:117-121 + model.py:57-59 differentiated; v = g_z
            f32x4 gl, lv4, mu4, ep4;
            if (full && ((reinterpret_cast<uintptr_t>(d.aux2) & 15) == 0) && ((reinterpret_cast<uintptr_t>(d.aux3) & 15) == 0)) {
                lv4 = ldg<f32x4>(d.aux2 + o); mu4 = ldg<f32x4>(aux + o); ep4 = ldg<f32x4>(d.aux3 + o);
            } else {  // clamped columns, no branch per element (values past N are never stored)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const size_t oe = (size_t)row * ldc + (col + e < N ? col + e : N - 1);
                    lv4[e] = ldg<float>(d.aux2 + oe); mu4[e] = ldg<float>(aux + oe); ep4[e] = ldg<float>(d.aux3 + oe);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gz = v[e], l = lv4[e];
                v[e] = gz + kb * mu4[e];
                gl[e] = gz * ep4[e] * 0.5f * expf(0.5f * l) + 0.5f * kb * (expf(l) - 1.0f);
            }
            if (full) {
                stg<f32x4>(d.C2 + o, gl);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < N) stg<float>(d.C2 + o + e, gl[e]);
            }
        } else if (epi == EPI_MASK || epi == EPI_ACCUM) {
            const float* __restrict__ src = (epi == EPI_MASK) ? aux : C;
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            if (full) {
                w = ldg<f32x4>(src + o);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < N) w[e] = ldg<float>(src + o + e);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (epi == EPI_MASK) ? (w[e] > 0.f ? v[e] : 0.f) : v[e] + w[e];
        }
#ifdef PSVAE_DIAG_NOSTORE
        if (v[0] == 123.456f) stg<f32x4>(C + o, v);
#else
        if (full) {
            stg<f32x4>(C + o, v);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < N) stg<float>(C + o + e, v[e]);
        }
#endif
        if (S2) *reinterpret_cast<f32x4*>(smem + lr * SC + 4 * c4) = v;  // the finished tile stays in LDS for stage 2
    }
    STAMP();  // 4: row loop done (stores issued)
    if (S2) {
        // ---- second stage: slab2[tn][row][:] = T[row][0..BN) * B2[n0.., :] (see GemmDesc::B2).  Wave w owns tile rows
        // 16w..16w+15; A fragments come from LDS (4 consecutive k per lane, as the "KC" operand of the main loop), B
        // fragments straight from global (k-major rows of Z2 floats, 16 B per lane).  Only compiled into the dedicated
        // <.., 6> instantiations, so the generic kernels stay as small as they were.
        static_assert(!S2 || (WM == 1 && WN == 1 && KS == 4 && TM == 4 && TN == 4), "stage 2 is written for config 6");
        __syncthreads();
        if (d.B2 != nullptr) {
            const int Z2 = d.Z2;
            const float* __restrict__ W2 = d.B2 + (size_t)n0 * Z2;
            float* __restrict__ slab = d.slab2 + (size_t)tn * M * Z2;
            for (int zc = 0; zc < Z2; zc += 64) {
                f32x4 a4[4];
                if (zc > 0) {  // (Z = 128: the second column group's fragments are fetched here)
#pragma unroll
                    for (int c2 = 0; c2 < 4; ++c2)
#pragma unroll
                        for (int t = 0; t < 4; ++t) b4[c2][t] = ldg<f32x4>(W2 + (size_t)(16 * c2 + 4 * q + t) * Z2 + zc + 4 * r);
                }
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) a4[c2] = *reinterpret_cast<const f32x4*>(smem + (16 * wave + r) * SC + 16 * c2 + 4 * q);
                f32x4 acc2[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) acc2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc2[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c2][t], b4[c2][t][n], acc2[n], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = m0 + 16 * wave + 4 * q + j;
                    if (row < M)
                        stg<f32x4>(slab + (size_t)row * Z2 + zc + 4 * r, f32x4{acc2[0][j], acc2[1][j], acc2[2][j], acc2[3][j]});
                }
            }
        }
        __syncthreads();
    }
    if (epi == EPI_BIAS_RECON) {  // fixed-order block sum of the squared errors -> one partial per tile
        float* __restrict__ red = smem + KS * (BM * SC) + KS * BM;  // 4 floats past the csum area
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) sq += __shfl_xor(sq, o2);
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += red[w];
            stg<float>(d.partials + tm * d.tiles_n + tn, tot);
        }
    }
    if (epi == EPI_REPARAM_BWD && tm == 0 && tn == 0 && ksg == 0) {
        // one workgroup finishes the loss: KL and squared-error partial sums in fixed order (lightning.py:113-124)
        float* __restrict__ red = smem + KS * (BM * SC) + KS * BM;
        float kls = 0.f, rs = 0.f;
        for (int i = tid; i < d.n_kl; i += NT) kls += ldg<float>(d.kl_part + i);
        for (int i = tid; i < d.n_recon; i += NT) rs += ldg<float>(d.partials + i);
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) {
            kls += __shfl_xor(kls, o2);
            rs += __shfl_xor(rs, o2);
        }
        __syncthreads();
        if (lane == 0) { red[wave] = kls; red[NW + wave] = rs; }
        __syncthreads();
        if (tid == 0) {
            float kt = 0.f, rt = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { kt += red[w]; rt += red[NW + w]; }
            const float recon = d.loss_D < 0 ? rt / (float)M : rt / ((float)M * (float)d.loss_D) / 10.0f;
            const float kl = -0.5f * kt / (float)M;
            stg<float>(d.loss3, recon + d.scale * kl);
            stg<float>(d.loss3 + 1, recon);
            stg<float>(d.loss3 + 2, kl);
            if (d.rng) d.rng[1] += d.rng_adv;
        }
    }
    if (AKM && d.colsum != nullptr && tn == 0 && tid < BM) {
        float s = csum_s[tid];
#pragma unroll
        for (int k2 = 1; k2 < KS; ++k2) s += csum_s[k2 * BM + tid];
        if (m0 + tid < M) stg<float>(d.colsum + (size_t)ksg * M + m0 + tid, s);
    }
#ifdef PSVAE_STAMP
    __builtin_amdgcn_s_waitcnt(0);  // stores of this wave done (vmcnt 0)
    STAMP();  // 5: stage 2 / loss tail / own stores drained
    if (tid == 0) {
        const unsigned long long rt1_ = __builtin_amdgcn_s_memrealtime();
        printf("S kind=%d M=%d N=%d K=%d cfg=%d epi=%d s2=%d wg=%d rt0=%llu rt1=%llu d=%llu,%llu,%llu,%llu,%llu\n", d.kind, M, N, K,
               d.cfg, d.epi, (int)(S2 && d.B2 != nullptr), (int)blockIdx.x, rt0_, rt1_, st_[1] - st_[0], st_[2] - st_[1],
               st_[3] - st_[2], st_[4] - st_[3], st_[5] - st_[4]);
    }
#endif
}

// ---- configurations -------------------------------------------------------------------------------
// id -> (wave tile TMxTN in 16x16 blocks, wave grid WMxWN, k slices KS, prefetch depth D)
template <int ID> struct CfgT;
#define PSVAE_CFG(ID, TM_, TN_, WM_, WN_, KS_, D_) \
    template <> struct CfgT<ID> { static constexpr int TM = TM_, TN = TN_, WM = WM_, WN = WN_, KS = KS_, D = D_; }
#ifndef PSVAE_D_BIG
#define PSVAE_D_BIG 2  // ring slots of the 4x4-block wave tiles (32 VGPRs per slot)
#endif
#ifndef PSVAE_D_SMALL
#define PSVAE_D_SMALL 4  // ring slots of the smaller wave tiles
#endif
PSVAE_CFG(1, 2, 2, 1, 1, 4, PSVAE_D_SMALL);  // 32x32   tile, k split 4
PSVAE_CFG(2, 2, 4, 1, 1, 4, PSVAE_D_SMALL);  // 32x64   tile, k split 4
PSVAE_CFG(3, 4, 2, 1, 1, 4, PSVAE_D_SMALL);  // 64x32   tile, k split 4
PSVAE_CFG(4, 2, 2, 1, 2, 2, PSVAE_D_SMALL);  // 32x64   tile, k split 2
PSVAE_CFG(5, 2, 2, 2, 2, 1, PSVAE_D_SMALL);  // 64x64   tile, no split
PSVAE_CFG(6, 4, 4, 1, 1, 4, PSVAE_D_BIG);    // 64x64   tile, k split 4
PSVAE_CFG(7, 4, 4, 2, 2, 1, PSVAE_D_BIG);    // 128x128 tile, no split (large batches)
PSVAE_CFG(8, 1, 2, 1, 1, 4, PSVAE_D_SMALL);  // 16x32   tile, k split 4 (small outputs, long k)
PSVAE_CFG(9, 4, 4, 1, 2, 2, PSVAE_D_BIG);    // 64x128  tile, k split 2
// (8-wave workgroups with k split 8 were measured too -- tools: profiles/r01 sweep -- and were never
// faster than these; the tile code supports them: NW = WM*WN*KS may be 4 or 8)
constexpr int NCFG = 9;

struct Cfg {
    int TM, TN, WM, WN, KS, D;
    float eff;   // MFMA-issue efficiency of the wave tile, measured (fragment reuse)
    bool small;  // <= 168 VGPRs: member of the high-occupancy kernel
};
static const Cfg kCfgs[NCFG] = {
    {2, 2, 1, 1, 4, 4, 0.80f, true},  {2, 4, 1, 1, 4, 4, 0.90f, false}, {4, 2, 1, 1, 4, 4, 0.88f, false},
    {2, 2, 1, 2, 2, 4, 0.80f, true},  {2, 2, 2, 2, 1, 4, 0.80f, true},  {4, 4, 1, 1, 4, 2, 0.92f, false},
    {4, 4, 2, 2, 1, 2, 0.92f, false}, {1, 2, 1, 1, 4, 4, 0.62f, true},  {4, 4, 1, 2, 2, 2, 0.92f, false},
};

template <int ID, bool S2 = false>
__device__ __forceinline__ void run_cfg(const GemmDesc& d, int tm, int tn, int ksg, float* __restrict__ smem) {
    typedef CfgT<ID> C;
    if (S2) {  // the stage-2 launches of the fused step are forward (k-major weights) or dgrad problems: NN only
        if (d.kind == GK_NN) gemm_tile<C::TM, C::TN, C::WM, C::WN, C::KS, C::D, false, true, S2>(d, tm, tn, ksg, smem);
        return;
    }
    if (d.kind == GK_NT)
        gemm_tile<C::TM, C::TN, C::WM, C::WN, C::KS, C::D, false, false>(d, tm, tn, ksg, smem);
    else if (d.kind == GK_NN)
        gemm_tile<C::TM, C::TN, C::WM, C::WN, C::KS, C::D, false, true>(d, tm, tn, ksg, smem);
    else
        gemm_tile<C::TM, C::TN, C::WM, C::WN, C::KS, C::D, true, true>(d, tm, tn, ksg, smem);
}

// One launch = a group of problems with up to TWO configs: the long-k wgrad problems (TN) run C0 (small
// tiles, deep k split), everything else runs C1 (wide tiles): workgroups of both shapes share the
// machine and fill each other's bubbles.  (One kernel holding all nine configs made hipcc park the
// fragment arrays in scratch; pairs compile clean.)
template <int C0, int C1, bool S2 = false>
__global__ __launch_bounds__(64 * CfgT<C0>::WM * CfgT<C0>::WN * CfgT<C0>::KS) void gemm_grouped_kernel(const GemmBatch batch) {
    static_assert(CfgT<C0>::WM * CfgT<C0>::WN * CfgT<C0>::KS == CfgT<C1>::WM * CfgT<C1>::WN * CfgT<C1>::KS,
                  "both configs of a launch need the same workgroup size");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware order (speed only): workgroups are dealt round-robin over the 8 XCDs, so give XCD x the
    // contiguous tile range [x*T/8, (x+1)*T/8): tiles that share an A row-panel then share one L2
    // instead of every L2 fetching every panel.  Bijective for any T (cdna guide, T1).
    const int tile = blockIdx.x;
    int di = 0;
    for (int i = 1; i < batch.n; ++i)
        if (tile >= batch.d[i].tile_begin) di = i;
    const GemmDesc& d = batch.d[di];
    int lt = tile - d.tile_begin;
    if (batch.xcd_remap) {  // within one problem (every XCD keeps its share of heavy and light problems)
        const int T = d.tiles_m * d.tiles_n * d.ksplit, q8 = T >> 3, r8 = T & 7;
        const int xcd = lt & 7, idx = lt >> 3;
        lt = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int tiles_mn = d.tiles_m * d.tiles_n;
    const int ksg = lt / tiles_mn;  // cross-workgroup k slice (0 when ksplit == 1)
    lt -= ksg * tiles_mn;
    const int tm = lt / d.tiles_n, tn = lt - tm * d.tiles_n;
    if (S2) {  // weight gradients (TN) run C0, every other problem is a config-6 tile with the second stage
        if (d.kind == GK_TN)
            run_cfg<C0>(d, tm, tn, ksg, smem);
        else
            run_cfg<C1, S2>(d, tm, tn, ksg, smem);
    } else if (C0 == C1 || d.cfg == C0) {
        run_cfg<C0>(d, tm, tn, ksg, smem);
    } else {
        run_cfg<C1>(d, tm, tn, ksg, smem);
    }
}

// Persistent variant for decode-sized launches of ONE config without weight-gradient problems: one workgroup per CU
// walks the tile list with a stride of the grid (no dispatch gap between a CU's consecutive tiles: +1.4 % on the 1 M-row
// decode).  A separate kernel on purpose: wrapping the common kernel's body in this loop cost the 13-launch step 3 %.
template <int C0>
__global__ __launch_bounds__(64 * CfgT<C0>::WM * CfgT<C0>::WN * CfgT<C0>::KS) void gemm_persistent_kernel(const GemmBatch batch) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int tile = blockIdx.x; tile < batch.total_tiles; tile += gridDim.x) {
        int di = 0;
        for (int i = 1; i < batch.n; ++i)
            if (tile >= batch.d[i].tile_begin) di = i;
        const GemmDesc& d = batch.d[di];
        const int lt = tile - d.tile_begin;
        const int tm = lt / d.tiles_n, tn = lt - tm * d.tiles_n;
        run_cfg<C0>(d, tm, tn, 0, smem);
        __syncthreads();  // the next tile reuses this tile's LDS
    }
}

template <int C0>
static hipError_t launch_persistent(const GemmBatch& b, size_t lds, hipStream_t stream) {
    static std::atomic<unsigned long long> lds_ok{0};  // per device: psvae_common.h lds_opt_in
    if (hipError_t attr_err = lds_opt_in(lds_ok, reinterpret_cast<const void*>(&gemm_persistent_kernel<C0>), 160 * 1024))
        return attr_err;
    constexpr int NT = 64 * CfgT<C0>::WM * CfgT<C0>::WN * CfgT<C0>::KS;
    hipLaunchKernelGGL((gemm_persistent_kernel<C0>), dim3(256), dim3(NT), lds, stream, b);
    return hipGetLastError();
}

// ---- host side ----------------------------------------------------------------------------------
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static std::atomic<int> g_forced{-1};  // -1: PSVAE_TILE (read once); the tuning hook psvae_debug_force_tile overrides
void debug_force_tile(int cfg) { g_forced.store(cfg); }
int num_gemm_configs() { return NCFG; }

// Cost model (cycles), fitted to rocprofv3 sweeps and tools/gemm_bench.py.  One workgroup of config c on problem d:
// per-wave MFMA cycles = chunks * 4 * blocks * 32 / eff, chunks outside whole groups of D without load/MFMA
// overlap (x1.6), plus a fixed prologue/epilogue.  A launch is then list-scheduled on the 256 CUs, one
// workgroup per CU at a time, in tile order (launch_cost).
static double wg_cycles(const GemmDesc& d, const Cfg& c) {
    const int S = d.ksplit > 1 ? d.ksplit : 1;
    const int chunks = ((d.K + 15) / 16 + S - 1) / S;
    const int per = (chunks + c.KS - 1) / c.KS;
    const int full = per / c.D * c.D, rem = per - full;
    double per_chunk = 4.0 * c.TM * c.TN * 32.0 / c.eff;
    if (d.kind == GK_NT) per_chunk *= 1.25;  // both operands k-contiguous: half-used 128-B lines per load
    if (d.kind == GK_TN) per_chunk *= 0.9;   // both operands k-major: whole lines, fewest address-unit cycles
    double cyc = full * per_chunk + rem * per_chunk * 1.6;
    // operand delivery: a workgroup pulls (BM + BN) * K * 4 bytes through its CU's L1 at ~11 B/clk when every CU is
    // busy (tools/microbench/load_rate.hip); wide tiles are MFMA-bound, 64x64 and smaller are bound by this
    static const double bpc = env_int("PSVAE_COST_BPC", 11);
    if (bpc > 0) {
        const int BM = c.WM * 16 * c.TM, BN = c.WN * 16 * c.TN;
        const double bytes = (double)(BM + BN) * (16.0 * chunks) * 4.0;
        const double feed = bytes / bpc;
        cyc = cyc > feed ? cyc : feed;
    }
    return cyc + 2200.0 + 150.0 * c.KS;
}
static long wg_count(const GemmDesc& d, const Cfg& c) {
    const int BM = c.WM * 16 * c.TM, BN = c.WN * 16 * c.TN;
    const int S = d.ksplit > 1 ? d.ksplit : 1;
    return (long)((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN) * S;
}
// makespan of the launch with config c_tn on the TN problems and c_other on the rest: workgroups are dispatched in
// tile order, each to the CU that frees first (one workgroup per CU at a time)
static double launch_cost(const GemmBatch& b, int c_tn, int c_other) {
    double free_at[256], tmp[256];  // kept ascending
    for (int i = 0; i < 256; ++i) free_at[i] = 0.0;
    for (int i = 0; i < b.n; ++i) {
        const GemmDesc& d = b.d[i];
        const Cfg& c = kCfgs[(d.kind == GK_TN ? c_tn : c_other) - 1];
        const double cyc = wg_cycles(d, c);
        long n = wg_count(d, c);
        if (n > 256 * 8) {  // many rounds: every CU gets the same share
            const double add = (double)(n / 256) * cyc;
            for (int x = 0; x < 256; ++x) free_at[x] += add;
            n %= 256;
        }
        while (n > 0) {  // the next min(n, 256) workgroups go to the earliest-free CUs; then merge the two runs
            const int take = n < 256 ? (int)n : 256;
            int x = 0, y = take, o = 0;
            while (x < take && y < 256) tmp[o++] = (free_at[x] + cyc <= free_at[y]) ? free_at[x++] + cyc : free_at[y++];
            while (x < take) tmp[o++] = free_at[x++] + cyc;
            while (y < 256) tmp[o++] = free_at[y++];
            memcpy(free_at, tmp, sizeof(free_at));
            n -= take;
        }
    }
    return free_at[255];
}

int pick_tile(const GemmBatch& b) {  // config of the first problem (test hook semantics)
    GemmBatch c = b;
    finalize_batch(c, 0);
    return c.d[0].cfg;
}

static const int kTnCands[] = {1, 2, 8};           // wgrad: small tiles, k split 4 (a tuned plan may also pick 6)
constexpr int NTN = 3, NOTHER = 5, NOTHER_NO_TN = 7;
static const int kOtherCands[] = {1, 2, 4, 6, 8, 7, 9};  // forward / dgrad; 7 and 9 (128-wide tiles) only in launches
                                                        // without weight-gradient problems (no mixed pair compiled)

// cfg == 0: choose one config for the TN problems and one for the others; else force `cfg` everywhere
void finalize_batch(GemmBatch& b, int cfg) {
    static const int env_tile = env_int("PSVAE_TILE", 0);
    const int forced = g_forced.load() < 0 ? env_tile : g_forced.load();
    if (cfg == 0 && forced > 0 && forced <= NCFG) cfg = forced;
    int pick_tn = cfg, pick_other = cfg;
    static const int force_tn = env_int("PSVAE_CFG_TN", 0), force_other = env_int("PSVAE_CFG_OTHER", 0);  // tuning
    if (cfg == 0) {
        // joint choice: the pair (TN config, config of the others) with the smallest modelled makespan
        bool has_tn = false, has_other = false;
        for (int i = 0; i < b.n; ++i) (b.d[i].kind == GK_TN ? has_tn : has_other) = true;
        // (memoised per problem signature: the model costs ~10 us of host time, which an eager launch sequence --
        // the autograd path -- would pay 13 times per step)
        struct Memo { int key[MAX_GROUP][5]; int n, tn, other; };
        static Memo memo[64];
        static int memo_n = 0;
        static std::mutex memo_mu;
        int key[MAX_GROUP][5] = {};
        for (int i = 0; i < b.n; ++i) {
            key[i][0] = b.d[i].M; key[i][1] = b.d[i].N; key[i][2] = b.d[i].K; key[i][3] = b.d[i].kind;
            key[i][4] = b.d[i].ksplit > 1 ? b.d[i].ksplit : 1;
        }
        bool hit = false;
        {
            std::lock_guard<std::mutex> lk(memo_mu);
            for (int m = 0; m < memo_n && !hit; ++m)
                if (memo[m].n == b.n && memcmp(memo[m].key, key, sizeof(key)) == 0) {
                    pick_tn = memo[m].tn; pick_other = memo[m].other; hit = true;
                }
        }
        if (!hit) {
            double best = 1e30;
            for (int a = 0; a < (has_tn ? NTN : 1); ++a)
                for (int o = 0; o < (has_other ? (has_tn ? NOTHER : NOTHER_NO_TN) : 1); ++o) {
                    const int ct = has_tn ? kTnCands[a] : kOtherCands[o], co = has_other ? kOtherCands[o] : ct;
                    const double e = launch_cost(b, ct, co);
                    if (e < best) { best = e; pick_tn = ct; pick_other = co; }
                }
            if (has_tn && has_other && pick_other != pick_tn) {
                // Two corrections the makespan model cannot see (one workgroup per CU at a time, a fixed cost per
                // workgroup), both measured on cfg2's backward, together 150.1 -> 144.5 us/step untuned:
                //  * forward/dgrad problems with k <= 64 have no k loop to amortise a 64x64 tile's LDS round trip: the
                //    weight gradients' small config serves them too, as many co-resident workgroups
                //    (heads' dgrad beside their weight gradients: <8,8> uncapped 10.3 us, <8,6> 14.1 us);
                //  * near a tie one config for both classes wins: <c,c> is half the code of a mixed pair, and a
                //    launch starts from a cold instruction cache (<8,8> 8.0 us, <8,1> 9.1 us on decoder.0's backward)
                bool short_k = true;
                for (int i = 0; i < b.n; ++i)
                    if (b.d[i].kind != GK_TN && b.d[i].K > 64) short_k = false;
                if (short_k || launch_cost(b, pick_tn, pick_tn) <= 1.02 * best) pick_other = pick_tn;
            }
            std::lock_guard<std::mutex> lk(memo_mu);
            Memo& m = memo[memo_n < 64 ? memo_n++ : (memo_n = 1, 0)];  // full: start over
            memcpy(m.key, key, sizeof(key));
            m.n = b.n; m.tn = pick_tn; m.other = pick_other;
        }
        if (force_tn > 0 && force_tn <= NCFG) pick_tn = force_tn;
        if (force_other > 0 && force_other <= NCFG) pick_other = force_other;
        if (b.hint_tn > 0 && b.hint_tn <= NCFG) pick_tn = b.hint_tn;  // a tuned plan wins
        if (b.hint_other > 0 && b.hint_other <= NCFG) pick_other = b.hint_other;
    }
    for (int i = 0; i < b.n; ++i)
        if (b.d[i].B2 != nullptr) {  // the second stage exists for config 6 only, beside weight gradients in 1, 2, 6 or 8
            pick_other = 6;
            if (pick_tn != 1 && pick_tn != 2 && pick_tn != 6 && pick_tn != 8) pick_tn = 2;
        }
    int t = 0;
    for (int i = 0; i < b.n; ++i) {
        GemmDesc& d = b.d[i];
        d.cfg = d.kind == GK_TN ? pick_tn : pick_other;
        const Cfg& c = kCfgs[d.cfg - 1];
        const int BM = c.WM * 16 * c.TM, BN = c.WN * 16 * c.TN;
        d.tiles_m = (d.M + BM - 1) / BM;
        d.tiles_n = (d.N + BN - 1) / BN;
        d.tile_begin = t;
        if (d.ksplit < 1) d.ksplit = 1;
        t += d.tiles_m * d.tiles_n * d.ksplit;
    }
    b.total_tiles = t;
    b.dbg = 0;
    static const int remap = env_int("PSVAE_XCD_REMAP", 0);  // measured: no gain at these sizes (L2 fills are not the limiter)
    b.xcd_remap = b.hint_remap ? (b.hint_remap == 1) : remap;
}

static size_t lds_bytes(const Cfg& c) {
    const int BM = c.WM * 16 * c.TM, BN = c.WN * 16 * c.TN;
    return (size_t)(c.KS * BM * (BN + 4) + c.KS * BM + 16) * sizeof(float);
}

// the dynamic-LDS limit of a kernel is raised once per (kernel, device): lds_opt_in (psvae_common.h), safe from any thread
// (the library keeps no other mutable state per kernel: SURVEY 8b threading contract)
template <int C0, int C1, bool S2 = false>
static hipError_t launch_pair(const GemmBatch& b, size_t lds, hipStream_t stream) {
    static std::atomic<unsigned long long> lds_ok{0};  // per device: psvae_common.h lds_opt_in
    if (hipError_t attr_err = lds_opt_in(lds_ok, reinterpret_cast<const void*>(&gemm_grouped_kernel<C0, C1, S2>), 160 * 1024))
        return attr_err;
    constexpr int NT = 64 * CfgT<C0>::WM * CfgT<C0>::WN * CfgT<C0>::KS;
    hipLaunchKernelGGL((gemm_grouped_kernel<C0, C1, S2>), dim3(b.total_tiles), dim3(NT), lds, stream, b);
    return hipGetLastError();
}

hipError_t launch_gemm_batch(GemmBatch& b, int cfg, hipStream_t stream) {
    if (b.n <= 0) return hipSuccess;
    if (cfg < 0 || cfg > NCFG) return hipErrorInvalidValue;
    finalize_batch(b, cfg);
    if (b.total_tiles == 0) return hipSuccess;
    int c0 = 0, c1 = 0;  // c0: config of the TN problems, c1: of the others (equal if a class is absent)
    size_t lds = 0;
    for (int i = 0; i < b.n; ++i) {
        (b.d[i].kind == GK_TN ? c0 : c1) = b.d[i].cfg;
        const size_t l = lds_bytes(kCfgs[b.d[i].cfg - 1]);
        lds = l > lds ? l : lds;
    }
    if (c0 == 0) c0 = c1;
    if (c1 == 0) c1 = c0;
    // Two co-resident workgroups of these kernels share one L1 and run slower than one after the other
    // (measured: 512 workgroups of config 6 take 37 us co-resident, 26 us capped), so a launch may ask for more
    // than half the CU's LDS to keep one workgroup per CU.
    static const int lds_min = env_int("PSVAE_LDS_MIN", 0);  // tuning
    if ((size_t)lds_min > lds) lds = lds_min;
    static const int cap_default = env_int("PSVAE_CAP_DEFAULT", 1);  // untuned launches with more workgroups than CUs
    bool many_small = c0 == 8 && c1 == 8;  // 16x32 tiles: light workgroups, meant to be co-resident (see finalize_batch)
    const bool cap = b.hint_cap == 1 || (b.hint_cap == 0 && cap_default && b.total_tiles > 256 && !many_small);
    if (cap && lds < 82 * 1024) lds = 82 * 1024;
    bool stage2 = false;  // a problem with a second stage (GemmDesc::B2): only the dedicated <.., 6, true> kernels run it
    for (int i = 0; i < b.n; ++i) stage2 = stage2 || b.d[i].B2 != nullptr;
    {   // decode-sized: >= 4 tiles per CU, one config, no weight gradients, no k split across workgroups
        static const int persist_min = env_int("PSVAE_PERSIST_MIN", 1024);
        bool plain = c0 == c1 && b.xcd_remap == 0 && !stage2;
        for (int i = 0; i < b.n; ++i) plain = plain && b.d[i].kind != GK_TN && b.d[i].ksplit == 1;
        if (plain && persist_min > 0 && b.total_tiles >= persist_min) {
            if (c0 == 7) return launch_persistent<7>(b, lds, stream);
            if (c0 == 6) return launch_persistent<6>(b, lds, stream);
        }
    }
    if (stage2) {
        for (int i = 0; i < b.n; ++i) {
            const GemmDesc& d = b.d[i];
            if (d.kind == GK_TN) continue;
            if (d.cfg != 6 || d.kind != GK_NN || d.B2 == nullptr || d.slab2 == nullptr || (d.N & 63) || (d.Z2 & 63) ||
                d.Z2 <= 0 || d.ksplit != 1 || (reinterpret_cast<uintptr_t>(d.B2) & 15) ||
                (reinterpret_cast<uintptr_t>(d.slab2) & 15))
                return hipErrorInvalidValue;
        }
        if (c0 == 6) return launch_pair<6, 6, true>(b, lds, stream);
        if (c0 == 1) return launch_pair<1, 6, true>(b, lds, stream);
        if (c0 == 2) return launch_pair<2, 6, true>(b, lds, stream);
        if (c0 == 8) return launch_pair<8, 6, true>(b, lds, stream);
        return hipErrorInvalidValue;
    }
#define PSVAE_PAIR(A, B) if (c0 == A && c1 == B) return launch_pair<A, B>(b, lds, stream)
    PSVAE_PAIR(1, 1); PSVAE_PAIR(2, 2); PSVAE_PAIR(3, 3); PSVAE_PAIR(4, 4); PSVAE_PAIR(5, 5);
    PSVAE_PAIR(6, 6); PSVAE_PAIR(7, 7); PSVAE_PAIR(8, 8); PSVAE_PAIR(9, 9);
    PSVAE_PAIR(1, 2); PSVAE_PAIR(1, 4); PSVAE_PAIR(1, 6); PSVAE_PAIR(1, 8);
    PSVAE_PAIR(8, 1); PSVAE_PAIR(8, 2); PSVAE_PAIR(8, 4); PSVAE_PAIR(8, 6);
    PSVAE_PAIR(6, 1); PSVAE_PAIR(6, 2); PSVAE_PAIR(6, 4); PSVAE_PAIR(6, 8);
    PSVAE_PAIR(2, 1); PSVAE_PAIR(2, 4); PSVAE_PAIR(2, 6); PSVAE_PAIR(2, 8);
    PSVAE_PAIR(6, 9); PSVAE_PAIR(2, 9);  // 64 x 128 dgrad tiles beside the weight gradients (heavy two-round launches)
#undef PSVAE_PAIR
    return hipErrorInvalidValue;
}

}  // namespace psvae
