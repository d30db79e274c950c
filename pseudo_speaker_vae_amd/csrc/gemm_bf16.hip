// bf16 MFMA path (BASELINE config 3: bf16 operands, fp32 accumulate, fp32 master weights).
//
// v_mfma_f32_16x16x32_bf16 wants 8 consecutive k per lane for BOTH operands, i.e. both operands
// k-contiguous in memory.  So the bf16 path keeps every activation / gradient / weight tensor in TWO bf16
// layouts, [rows][features] and [features][rows(padded to 32)], written by the producing kernel's epilogue,
// and every GEMM is the same "NT" product C[m][n] = sum_k A[m][k] * B[n][k]:
//   forward  h' = h W^T        A = h  [B,K]        B = W   [N,K]
//   dgrad    dX = dY W         A = dY [B,N]        B = W^T [K,N]
//   wgrad    dW = dY^T X       A = dY^T [N,Bpad]   B = X^T [K,Bpad]      (contraction over the padded batch)
// Structure as the fp32 kernel (csrc/gemm_f32.hip): register-direct fragments (one 16-B load = 8 bf16 per
// lane per 16-row block per 32-k chunk, no permutation needed), D chunks in flight, split-K across the 4
// waves, one LDS meeting point for the k slices, fused epilogue writing fp32 and/or both bf16 layouts.
#include <hip/hip_bf16.h>

#include "psvae_common.h"

namespace psvae {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define PSVAE_GLOBAL __attribute__((address_space(1)))
template <typename V, typename P>
__device__ __forceinline__ V ldgb(const P* p) {
    return *reinterpret_cast<const PSVAE_GLOBAL V*>(reinterpret_cast<uintptr_t>(p));
}
template <typename V, typename P>
__device__ __forceinline__ void stgb(P* p, const V& v) {
    *reinterpret_cast<PSVAE_GLOBAL V*>(reinterpret_cast<uintptr_t>(p)) = v;
}

__device__ __forceinline__ u16 f2bf(float f) {  // round-to-nearest-even, NaN preserved (plain cast)
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<u16*>(&h);
}
__device__ __forceinline__ float bf2f(u16 u) { return __uint_as_float(((unsigned)u) << 16); }

// STAGED: the 128x128 tile's operand panels go through LDS (each element fetched once per workgroup, whole 128-B
// lines per 8 lanes) instead of every wave fetching its own fragments -- at these sizes the bf16 GEMMs are pure
// operand traffic (the MFMAs of a 2-GFLOP launch take 1 us), so bytes per workgroup decide the time.
// CT: launches that carry a conditional table get their own instantiation (the table tile goes through LDS, below);
// the CT = false code is what every other launch has always run.
template <int TM, int TN, int WM, int WN, int KS, int D, bool STAGED, bool CT>
__device__ __forceinline__ void gemm_tile_bf16(const GemmDescBf& d, int tm, int tn, int ksg, float* __restrict__ smem) {
    constexpr int NW = WM * WN * KS;
    static_assert(NW == 4, "4 waves per workgroup");
    constexpr int NT = 256;
    constexpr int BM = WM * 16 * TM, BN = WN * 16 * TN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wave / (WM * WN), ws = wave % (WM * WN);
    const int wm = ws / WN, wn = ws % WN;
    const int r = lane & 15, q = lane >> 4;
    const int M = d.M, N = d.N, K = d.K;  // K % 32 == 0 (host pads the contracted dimension)
    const int m0 = tm * BM, n0 = tn * BN;
    const int am = m0 + wm * 16 * TM, bn = n0 + wn * 16 * TN;
    const u16* __restrict__ A = d.A;
    const u16* __restrict__ B = d.B;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float csum[TM];
#pragma unroll
    for (int m = 0; m < TM; ++m) csum[m] = 0.f;
    const bool do_colsum = (d.colsum != nullptr) && (tn == 0) && (wn == 0);

    if constexpr (STAGED) {
        static_assert(KS == 1 && WM == 2 && WN == 2 && TM == 4 && TN == 4, "staged loop: 128x128 tile, 2x2 waves");
        constexpr int BK = 64, LS = BK + 8;  // LDS row stride in bf16: 144 B keeps ds_read_b128 of 16 rows conflict-free
        u16* sA = reinterpret_cast<u16*>(smem);            // [2][BM][LS]
        u16* sB = sA + 2 * BM * LS;                        // [2][BN][LS]
        // global -> register -> LDS: 16-B pieces, 8 consecutive lanes = one 128-B line of one row
        const int prow = tid >> 3, pk = (tid & 7) * 8;     // piece (row, k) of pass i: row = prow + 32 i
        const u16* ga[4];
        const u16* gb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ra = m0 + prow + 32 * i, rb = n0 + prow + 32 * i;
            ra = ra < M ? ra : M - 1;  // clamped rows only feed outputs the epilogue never stores
            rb = rb < N ? rb : N - 1;
            ga[i] = A + (size_t)ra * d.lda + pk;
            gb[i] = B + (size_t)rb * d.ldb + pk;
        }
        bf16x8 ra_[4], rb_[4];
        const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        auto gload = [&](int t) {
            const int k0 = t * BK;
            const bool in = k0 + pk < K;  // K % 32 == 0: the last tile may hold only its first half
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra_[i] = in ? ldgb<bf16x8>(ga[i] + k0) : zero8;
                rb_[i] = in ? ldgb<bf16x8>(gb[i] + k0) : zero8;
            }
        };
        auto lstore = [&](int buf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<bf16x8*>(sA + (buf * BM + prow + 32 * i) * LS + pk) = ra_[i];
                *reinterpret_cast<bf16x8*>(sB + (buf * BN + prow + 32 * i) * LS + pk) = rb_[i];
            }
        };
        // k tiles of this workgroup: all of them, or slice ksg of a cross-workgroup split (weight gradients)
        const int nt_all = (K + BK - 1) / BK;
        const int S = d.ksplit > 1 ? d.ksplit : 1;
        const int per = (nt_all + S - 1) / S;
        const int t_beg = ksg * per;
        const int nt = (t_beg + per < nt_all ? t_beg + per : nt_all);  // end (exclusive)
        if (t_beg < nt) {
            gload(t_beg);
            lstore(t_beg & 1);
        }
        __syncthreads();
        for (int t = t_beg; t < nt; ++t) {
            const int buf = t & 1;
            if (t + 1 < nt) gload(t + 1);  // in flight under this tile's MFMAs
#pragma unroll
            for (int s2 = 0; s2 < BK / 32; ++s2) {
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int m = 0; m < TM; ++m)
                    fa[m] = *reinterpret_cast<const bf16x8*>(sA + (buf * BM + wm * 64 + 16 * m + r) * LS + 32 * s2 + 8 * q);
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    fb[n] = *reinterpret_cast<const bf16x8*>(sB + (buf * BN + wn * 64 + 16 * n + r) * LS + 32 * s2 + 8 * q);
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
                if (do_colsum) {
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int e = 0; e < 8; ++e) csum[m] += bf2f((u16)fa[m][e]);
                }
            }
            if (t + 1 < nt) lstore(buf ^ 1);  // the other buffer: last read before the previous barrier
            __syncthreads();
        }
        // (the loop's last barrier also fences the staging buffers against the epilogue's reuse of this LDS)
    } else {
    // row pointers (rows clamped: they only feed outputs the epilogue never stores)
    const u16* pa[TM];
    const u16* pb[TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        int row = am + 16 * m + r;
        row = row < M ? row : M - 1;
        pa[m] = A + (size_t)row * d.lda + 8 * q;
    }
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        int row = bn + 16 * n + r;
        row = row < N ? row : N - 1;
        pb[n] = B + (size_t)row * d.ldb + 8 * q;
    }
    const int nc = K >> 5;  // 32-k chunks
    const int per = (nc + KS - 1) / KS;
    const int cbeg = ks * per;
    const int cend = (cbeg + per < nc) ? cbeg + per : nc;

    bf16x8 fa[D][TM], fb[D][TN];
    auto load = [&](int s, int c) {
#pragma unroll
        for (int m = 0; m < TM; ++m) fa[s][m] = ldgb<bf16x8>(pa[m] + 32 * c);
#pragma unroll
        for (int n = 0; n < TN; ++n) fb[s][n] = ldgb<bf16x8>(pb[n] + 32 * c);
    };
    auto mma = [&](int s) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s][m], fb[s][n], acc[m][n], 0, 0, 0);
        if (do_colsum) {
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) csum[m] += bf2f((u16)fa[s][m][e]);
        }
    };
    // software pipeline over a ring of D fragment slots, pinned with sched_barriers (see gemm_f32.hip: left alone the
    // compiler sinks all loads of a group below its last MFMA)
    int c = cbeg;
    const int nfull = cend > cbeg ? (cend - cbeg) / D : 0;
    if (nfull > 0) {
#pragma unroll
        for (int s = 0; s < D - 1; ++s) load(s, cbeg + s);
        for (int g = 0; g < nfull - 1; ++g) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                load((s + D - 1) % D, cbeg + g * D + s + D - 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(s);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        load(D - 1, cbeg + nfull * D - 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < D; ++s) mma(s);
        c = cbeg + nfull * D;
    }
    for (; c < cend; c += D) {
#pragma unroll
        for (int s = 0; s < D; ++s)
            if (c + s < cend) load(s, c + s);
#pragma unroll
        for (int s = 0; s < D; ++s)
            if (c + s < cend) mma(s);
    }

    }

    // ---- epilogue ----
    // CT: the conditional table tile [6][BN] of this workgroup's columns, one 16-byte piece per thread, in flight while
    // the accumulators go to LDS; parked in LDS ([7][BN], row 6 = zeros for unknown labels) and read by label in the row
    // loop.  (The register version -- six preloaded vectors and two select chains per row -- cost 6.4 us on the first
    // encoder layer and 4.5 us on decoder.0; this one 3.3 / 1.5 us.  Same additions in the same order: same bits.)
    constexpr int CTQ = BN / 4;
    f32x4 ctpre = {0.f, 0.f, 0.f, 0.f};
    if constexpr (CT) {
        if (d.ctab && tid < 6 * CTQ) {
            const int colj = n0 + 4 * (tid % CTQ);
            const float* tp = d.ctab + (size_t)(tid / CTQ) * N;
            if (colj + 3 < N && ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.ctab) & 15) == 0)) {
                ctpre = ldgb<f32x4>(tp + colj);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) ctpre[e] = ldgb<float>(tp + (colj + e < N ? colj + e : N - 1));
            }
        }
    }
    constexpr int SC = BN + 4;
    float* __restrict__ Cs = smem + ks * (BM * SC);
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lr = wm * 16 * TM + 16 * m + 4 * q + j;
#pragma unroll
            for (int n = 0; n < TN; ++n) Cs[lr * SC + wn * 16 * TN + 16 * n + r] = acc[m][n][j];
        }
    float* __restrict__ csum_s = smem + KS * (BM * SC);
    if (do_colsum) {
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            float s = csum[m];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (q == 0) csum_s[ks * BM + wm * 16 * TM + 16 * m + r] = s;
        }
    }
    float* __restrict__ ct_s = smem + KS * (BM * SC) + KS * BM + 8;
    if constexpr (CT) {
        if (d.ctab) {
            if (tid < 6 * CTQ) *reinterpret_cast<f32x4*>(ct_s + (tid / CTQ) * BN + 4 * (tid % CTQ)) = ctpre;
            if (tid < CTQ) *reinterpret_cast<f32x4*>(ct_s + 6 * BN + 4 * tid) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    const int epi = d.epi;
    constexpr int CQ = BN / 4;
    constexpr int RSTEP = NT / CQ;
    const int c4 = tid % CQ, r0 = tid / CQ;
    const int col = n0 + 4 * c4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (d.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col + e < N) bias4[e] = ldgb<float>(d.bias + col + e);
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ct[6] = {zero4, zero4, zero4, zero4, zero4, zero4};  // this thread's 4 columns of the conditional table
    if constexpr (!CT) {
        if (d.ctab) {
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < N) ct[c][e] = ldgb<float>(d.ctab + (size_t)c * N + col + e);
        }
    }
    float sq = 0.f;
    const float gsc = (epi == BEPI_RECON) ? d.scale * (d.go ? ldgb<float>(d.go) : 1.0f) : 0.f;
    const bool colfull = col + 3 < N;
#pragma unroll
    for (int rr = 0; rr < BM; rr += RSTEP) {
        const int lr = rr + r0;
        const int row = m0 + lr;
        if (lr >= BM) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(smem + lr * SC + 4 * c4);
#pragma unroll
        for (int k2 = 1; k2 < KS; ++k2) v += *reinterpret_cast<const f32x4*>(smem + k2 * (BM * SC) + lr * SC + 4 * c4);
        if (row < M && col < N) {
            v += bias4;
            if (d.ctab) {  // one-hot conditioning: + table[gender] + table[3 + age], table columns preloaded above
                const int g = d.gender[row], a = d.age[row];
                if constexpr (CT) {
                    const int gi = (unsigned)g < 3u ? g : 6, ai = (unsigned)a < 3u ? 3 + a : 6;
                    v += *reinterpret_cast<const f32x4*>(ct_s + gi * BN + 4 * c4);
                    v += *reinterpret_cast<const f32x4*>(ct_s + ai * BN + 4 * c4);
                } else {
                    v += (g == 0 ? ct[0] : g == 1 ? ct[1] : g == 2 ? ct[2] : zero4);
                    v += (a == 0 ? ct[3] : a == 1 ? ct[4] : a == 2 ? ct[5] : zero4);
                }
            }
            if (epi == BEPI_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (epi == BEPI_MASK) {  // ReLU backward: mask = saved bf16 activation [M, ldmask]
                const u16* mp = d.mask + (size_t)row * d.ldmask + col;
                if (colfull && ((d.ldmask & 3) == 0)) {
                    const u32x2 mk = ldgb<u32x2>(mp);  // 4 bf16
                    const u16 m4[4] = {(u16)(mk.x & 0xffff), (u16)(mk.x >> 16), (u16)(mk.y & 0xffff), (u16)(mk.y >> 16)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = bf2f(m4[e]) > 0.f ? v[e] : 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < N) v[e] = bf2f(ldgb<u16>(mp + e)) > 0.f ? v[e] : 0.f;
                }
            }
            if (d.Cf) {  // fp32 output (final outputs, wgrads)
                float* p = d.Cf + (size_t)ksg * M * d.ldcf + (size_t)row * d.ldcf + col;  // ksplit: this slice's slab
                if (colfull && ((d.ldcf & 3) == 0)) {
                    stgb<f32x4>(p, v);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < N) stgb<float>(p + e, v[e]);
                }
            }
            if (epi == BEPI_RECON) {  // v = x_hat: emit the MSE gradient instead of v into the bf16 outputs
                const float* tp = d.target + (size_t)row * d.ldcf + col;
                f32x4 xt = zero4;
                if (colfull && ((d.ldcf & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.target) & 15) == 0)) {
                    xt = ldgb<f32x4>(tp);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < N) xt[e] = ldgb<float>(tp + e);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float df = (col + e < N) ? v[e] - xt[e] : 0.f;
                    sq += df * df;
                    v[e] = gsc * df;
                }
            }
            if (d.Cb) {
                u16* p = d.Cb + (size_t)row * d.ldcb + col;
                if (colfull && ((d.ldcb & 3) == 0)) {  // 4 bf16 = one 8-byte store
                    u32x2 pk;
                    pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                    pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                    stgb<u32x2>(p, pk);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < N) stgb<u16>(p + e, f2bf(v[e]));
                }
            }
        } else {
            v = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (d.CbT) *reinterpret_cast<f32x4*>(smem + lr * SC + 4 * c4) = v;  // final values for the transposed pass
    }
    if (d.CbT) {  // transposed bf16 copy [N][ldt]: column reads of the finished tile, 4 rows per lane
        __syncthreads();
        constexpr int MQ = BM / 4;            // row quads
        constexpr int QW = MQ >= 8 ? 8 : MQ;  // quads per 64-item block (BM = 16: 4)
        constexpr int CW = 64 / QW;           // columns per block (BN is a multiple)
        constexpr int QB = MQ / QW;
        static_assert(BN % CW == 0 && MQ % QW == 0, "tile shape vs transposed-store blocks");
        for (int i = tid; i < BN * MQ; i += NT) {
            // lane order inside 64 items: QW consecutive row quads (64-B contiguous stores) x CW consecutive columns
            // (LDS banks 16*mq + n: at most 2-way conflicts, the plain n-major order was 8-way)
            const int blk = i >> 6, l = i & 63;
            const int mq = (blk % QB) * QW + (l % QW), n = (blk / QB) * CW + (l / QW);
            const int gn = n0 + n, gm = m0 + 4 * mq;
            if (gn >= N) continue;
            u16* p = d.CbT + (size_t)gn * d.ldt + gm;
            if (gm + 3 < M && ((d.ldt & 3) == 0)) {
                u32x2 pk;
                pk.x = (unsigned)f2bf(smem[(4 * mq + 0) * SC + n]) | ((unsigned)f2bf(smem[(4 * mq + 1) * SC + n]) << 16);
                pk.y = (unsigned)f2bf(smem[(4 * mq + 2) * SC + n]) | ((unsigned)f2bf(smem[(4 * mq + 3) * SC + n]) << 16);
                stgb<u32x2>(p, pk);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (gm + e < M) stgb<u16>(p + e, f2bf(smem[(4 * mq + e) * SC + n]));
            }
        }
    }
    if (epi == BEPI_RECON) {
        float* __restrict__ red = smem + KS * (BM * SC) + KS * BM;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) sq += __shfl_xor(sq, o2);
        __syncthreads();
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (tid == 0) stgb<float>(d.partials + tm * d.tiles_n + tn, (red[0] + red[1]) + (red[2] + red[3]));
    }
    if (d.colsum != nullptr && tn == 0 && tid < BM) {
        float s = csum_s[tid];
#pragma unroll
        for (int k2 = 1; k2 < KS; ++k2) s += csum_s[k2 * BM + tid];
        if (m0 + tid < M) stgb<float>(d.colsum + (size_t)ksg * M + m0 + tid, s);
    }
}

template <int TM, int TN, int WM, int WN, int KS, int D, bool STAGED, bool CT>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmBatchBf batch) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tile = blockIdx.x;
    int di = 0;
    for (int i = 1; i < batch.n; ++i)
        if (tile >= batch.d[i].tile_begin) di = i;
    const GemmDescBf& d = batch.d[di];
    int lt = tile - d.tile_begin;
    const int tiles_mn = d.tiles_m * d.tiles_n;
    const int ksg = STAGED ? lt / tiles_mn : 0;  // cross-workgroup k slice (staged kernel only)
    lt -= ksg * tiles_mn;
    const int tm = lt / d.tiles_n, tn = lt - tm * d.tiles_n;
    gemm_tile_bf16<TM, TN, WM, WN, KS, D, STAGED, CT>(d, tm, tn, ksg, smem);
}

template <int TM, int TN, int WM, int WN, int KS, int D, bool STAGED, bool CT>
static hipError_t launch_bf_ct(GemmBatchBf& b, hipStream_t stream) {
    constexpr int BM = WM * 16 * TM, BN = WN * 16 * TN;
    constexpr size_t lds_epi = (size_t)(KS * BM * (BN + 4) + KS * BM + 8 + (CT ? 7 * BN : 0)) * sizeof(float);  // CT: + the table tile
    constexpr size_t lds_stage = STAGED ? (size_t)2 * (BM + BN) * (64 + 8) * 2 : 0;  // two buffers of A and B panels
    constexpr size_t lds = lds_epi > lds_stage ? lds_epi : lds_stage;
    int t = 0;
    for (int i = 0; i < b.n; ++i) {
        b.d[i].tiles_m = (b.d[i].M + BM - 1) / BM;
        b.d[i].tiles_n = (b.d[i].N + BN - 1) / BN;
        b.d[i].tile_begin = t;
        const int S = (STAGED && b.d[i].ksplit > 1) ? b.d[i].ksplit : 1;
        if (!STAGED && b.d[i].ksplit > 1) return hipErrorInvalidValue;  // only the staged kernel splits k over workgroups
        t += b.d[i].tiles_m * b.d[i].tiles_n * S;
    }
    b.total_tiles = t;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<TM, TN, WM, WN, KS, D, STAGED, CT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds > 96 * 1024 ? lds : 96 * 1024));
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    // more workgroups than CUs: one workgroup per CU at a time (co-resident workgroups thrash the shared L1, see
    // gemm_f32.hip); the request only has to exceed half the CU's LDS
    static const bool cap_on = getenv("PSVAE_BF16_CAP") != nullptr && atoi(getenv("PSVAE_BF16_CAP")) != 0;  // measured: slower here
    size_t lds_req = lds;
    if (cap_on && t > 256 && lds_req < 82 * 1024) lds_req = 82 * 1024;
    hipLaunchKernelGGL((gemm_bf16_kernel<TM, TN, WM, WN, KS, D, STAGED, CT>), dim3(t), dim3(256), lds_req, stream, b);
    return hipGetLastError();
}

template <int TM, int TN, int WM, int WN, int KS, int D, bool STAGED = false>
static hipError_t launch_bf(GemmBatchBf& b, hipStream_t stream) {
    bool ct = false;
    for (int i = 0; i < b.n; ++i) ct = ct || (b.d[i].ctab != nullptr);
    return ct ? launch_bf_ct<TM, TN, WM, WN, KS, D, STAGED, true>(b, stream)
              : launch_bf_ct<TM, TN, WM, WN, KS, D, STAGED, false>(b, stream);
}

// config choice: wide register tiles (the bf16 pipe is 16x faster than fp32, the fragment loads are the
// bottleneck: 64x64 per wave halves the bytes per MFMA of 32x32); k split only when the tile count is low
hipError_t launch_gemm_batch_bf16(GemmBatchBf& b, hipStream_t stream) {
    if (b.n <= 0) return hipSuccess;
    long t44 = 0, t22 = 0, t88 = 0;
    int kmax = 0;
    for (int i = 0; i < b.n; ++i) {
        t88 += (long)((b.d[i].M + 127) / 128) * ((b.d[i].N + 127) / 128);
        t44 += (long)((b.d[i].M + 63) / 64) * ((b.d[i].N + 63) / 64);
        t22 += (long)((b.d[i].M + 31) / 32) * ((b.d[i].N + 31) / 32);
        kmax = b.d[i].K > kmax ? b.d[i].K : kmax;
    }
    const char* bm = getenv("PSVAE_BF16_BIG_MIN");  // tuning / tests (read per call: a test flips it in-process)
    const int big_min = bm ? atoi(bm) : 2048;
    if (b.force_staged) return launch_bf<4, 4, 2, 2, 1, 2, true>(b, stream);  // backward of a large batch (see the API)
    static const int t88_min = getenv("PSVAE_BF16_T88_MIN") ? atoi(getenv("PSVAE_BF16_T88_MIN")) : 256;
    if (t88_min > 0 && t88 >= t88_min && kmax <= 1024)  // a full machine of 128x128 tiles: half the operand bytes of 64x64
        return launch_bf<4, 4, 2, 2, 1, 2, true>(b, stream);
    if (t44 >= 192) {
        const char* st = getenv("PSVAE_BF16_STAGED");
        const int staged = st ? atoi(st) : 1;
        if (t44 >= big_min && kmax <= 1024)  // 128x128 tiles, no split
            return staged ? launch_bf<4, 4, 2, 2, 1, 2, true>(b, stream) : launch_bf<4, 4, 2, 2, 1, 2>(b, stream);
        return launch_bf<4, 4, 1, 1, 4, 2>(b, stream);                                    // 64x64, k split 4
    }
    if (t22 >= 128) return launch_bf<2, 2, 1, 1, 4, 4>(b, stream);                         // 32x32, k split 4
    return launch_bf<1, 2, 1, 1, 4, 4>(b, stream);                                         // 16x32, k split 4
}

// ---- fp32 [R,C] -> bf16 [R,C] and bf16 [C, ldt] (32x32 LDS tile), several tensors per launch -----------
struct CastArgs {
    const float* src[12];
    u16* dst[12];
    u16* dstT[12];
    int R[12], Cc[12], ld[12], ldt[12], tile_begin[12];
    int n;
    // optional rider: the transposed one-hot label matrix ohT[6][Bp] (blocks past the cast tiles)
    int cast_tiles;
    const int32_t* gender;
    const int32_t* age;
    u16* ohT;
    int B, Bp;
};

__global__ __launch_bounds__(256) void cast_transpose_kernel(const CastArgs a) {
    __shared__ float tile[32][33];
    if ((int)blockIdx.x >= a.cast_tiles) {  // one-hot rider blocks (one launch fewer per step)
        const int b = ((int)blockIdx.x - a.cast_tiles) * 256 + threadIdx.x;
        if (b < a.Bp) {
            const int g = b < a.B ? a.gender[b] : -1, ag = b < a.B ? a.age[b] : -1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                a.ohT[(size_t)c * a.Bp + b] = (g == c) ? (u16)0x3F80 : (u16)0;
                a.ohT[(size_t)(3 + c) * a.Bp + b] = (ag == c) ? (u16)0x3F80 : (u16)0;
            }
        }
        return;
    }
    int pi = 0;
    for (int i = 1; i < a.n; ++i)
        if ((int)blockIdx.x >= a.tile_begin[i]) pi = i;
    const int R = a.R[pi], Cc = a.Cc[pi];
    const int tiles_c = (Cc + 31) / 32;
    const int lt = blockIdx.x - a.tile_begin[pi];
    const int r0 = (lt / tiles_c) * 32, c0 = (lt % tiles_c) * 32;
    const float* __restrict__ src = a.src[pi];
    u16* __restrict__ dst = a.dst[pi];
    u16* __restrict__ dstT = a.dstT[pi];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = r0 + ty + 8 * i, cc = c0 + tx;
        const float v = (rr < R && cc < Cc) ? src[(size_t)rr * a.ld[pi] + cc] : 0.f;
        tile[ty + 8 * i][tx] = v;
        if (dst && rr < R && cc < Cc) dst[(size_t)rr * Cc + cc] = f2bf(v);
    }
    __syncthreads();
    if (dstT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cc = c0 + ty + 8 * i, rr = r0 + tx;
            if (cc < Cc && rr < R) dstT[(size_t)cc * a.ldt[pi] + rr] = f2bf(tile[tx][ty + 8 * i]);
        }
    }
}

hipError_t launch_cast_transpose(const float* const* src, u16* const* dst, u16* const* dstT, const int* R, const int* Cc,
                                 const int* ld, const int* ldt, int n, hipStream_t s, const int32_t* gender,
                                 const int32_t* age, u16* ohT, int B, int Bp) {
    CastArgs a{};
    int t = 0;
    for (int i = 0; i < n; ++i) {
        a.src[i] = src[i]; a.dst[i] = dst[i]; a.dstT[i] = dstT[i];
        a.R[i] = R[i]; a.Cc[i] = Cc[i]; a.ld[i] = ld[i]; a.ldt[i] = ldt[i];
        a.tile_begin[i] = t;
        t += ((R[i] + 31) / 32) * ((Cc[i] + 31) / 32);
    }
    a.n = n;
    a.cast_tiles = t;
    if (ohT) {
        a.gender = gender; a.age = age; a.ohT = ohT; a.B = B; a.Bp = Bp;
        t += (Bp + 255) / 256;
    }
    hipLaunchKernelGGL(cast_transpose_kernel, dim3(t), dim3(256), 0, s, a);
    return hipGetLastError();
}

// one-hot matrix of the labels, transposed: ohT[c][b] (bf16 0/1), c < 6, b < Bp (padding columns zero).
// With it the conditional-table gradient  gtab[c][n] = sum_b onehot[b][c] dY[b][n]  is one more wgrad GEMM.
__global__ __launch_bounds__(256) void onehot_t_kernel(const int32_t* __restrict__ gender, const int32_t* __restrict__ age,
                                                       u16* __restrict__ ohT, int B, int Bp) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= Bp) return;
    const int g = b < B ? gender[b] : -1, a = b < B ? age[b] : -1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ohT[(size_t)c * Bp + b] = (g == c) ? (u16)0x3F80 : (u16)0;
        ohT[(size_t)(3 + c) * Bp + b] = (a == c) ? (u16)0x3F80 : (u16)0;
    }
}
hipError_t launch_onehot_t(const int32_t* gender, const int32_t* age, u16* ohT, int B, int Bp, hipStream_t s) {
    hipLaunchKernelGGL(onehot_t_kernel, dim3((Bp + 255) / 256), dim3(256), 0, s, gender, age, ohT, B, Bp);
    return hipGetLastError();
}

// conditional-table gradients from bf16 dY (see cond_colsum_kernel in elementwise.hip)
struct CondSumArgsBf {
    const u16* dY[3];
    float* gtab[3];
};
__global__ __launch_bounds__(256) void cond_colsum_bf16_kernel(CondSumArgsBf a, const int32_t* __restrict__ gender,
                                                               const int32_t* __restrict__ age, int B, int N) {
    __shared__ float sh[4][6][64];
    const u16* __restrict__ dY = a.dY[blockIdx.y];
    float* __restrict__ gt = a.gtab[blockIdx.y];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        for (int b = rl; b < B; b += 4) {
            const float v = bf2f(dY[(size_t)b * N + col]);
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

hipError_t launch_cond_colsum_bf16(const u16* const* dY, float* const* gtab, int nprob, const int32_t* gender,
                                   const int32_t* age, int B, int N, hipStream_t s) {
    CondSumArgsBf a{};
    for (int i = 0; i < nprob; ++i) { a.dY[i] = dY[i]; a.gtab[i] = gtab[i]; }
    hipLaunchKernelGGL(cond_colsum_bf16_kernel, dim3((N + 63) / 64, nprob), dim3(256), 0, s, a, gender, age, B, N);
    return hipGetLastError();
}

}  // namespace psvae
