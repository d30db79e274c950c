// bf16 MFMA path (BASELINE config 3: bf16 operands, fp32 accumulate, fp32 master weights).
//
// v_mfma_f32_16x16x32_bf16 wants 8 consecutive k per lane for BOTH operands.  Activations and gradients exist in ONE
// bf16 layout, [rows][features]; weights in two ([out,in] and [in,out]: 2.5 MB, cast once per pass).  Every GEMM is
// C[m][n] = sum_k A[m][k] * B[n][k]:
//   forward  h' = h W^T        A = h  [B,K]        B = W   [N,K]          (both k-contiguous in memory)
//   dgrad    dX = dY W         A = dY [B,N]        B = W^T [K,N]          (both k-contiguous)
//   wgrad    dW = dY^T X       A = dY [B,out]      B = X [B,in]           (both k-MAJOR: the batch is the row index)
// The weight gradients therefore stage their [64 batch rows][128 features] operand tiles in LDS exactly as they lie in
// memory (whole 256-B row segments, coalesced) and read the fragments TRANSPOSED with ds_read_b64_tr_b16 (a 4 x 16
// block per 16 lanes, delivered column-major: cdna_hip_programming.md T10) -- no second copy of any activation is ever
// written (round 1 kept a [features][batch] copy of each: 48 MB of extra stores per step at B = 4096).
// Structure as the fp32 kernel (csrc/gemm_f32.hip): register-direct fragments (one 16-B load = 8 bf16 per
// lane per 16-row block per 32-k chunk, no permutation needed), D chunks in flight, split-K across the 4
// waves, one LDS meeting point for the k slices, fused epilogue writing fp32 and/or both bf16 layouts.
#include <hip/hip_bf16.h>

#include <type_traits>

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

#ifdef PSVAE_STAMP
    unsigned long long st_[16];
    int nst_ = 0;
#define STAMP() do { if (nst_ < 16) st_[nst_++] = __builtin_amdgcn_s_memtime(); } while (0)
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
    const bool do_colsum = (d.colsum != nullptr) && (tn == 0) && (wn == 0);

    if constexpr (STAGED) {
      static_assert(KS == 1 && WM == 2 && WN == 2 && TM == 4 && (TN == 4 || TN == 2), "staged loop: 128x128 or 128x64 tile, 2x2 waves");
      constexpr int PB = BN / 32;  // staging passes of the B operand (A: 4); the 128x64 tile lets two workgroups share a CU
      typedef std::integral_constant<int, 0> I0;
      typedef std::integral_constant<int, 1> I1;
      if (d.tr) {
        // ---- k-MAJOR operands (weight gradients): A = dY [K rows = batch][M], B = X [K][N], row-major as produced.
        // LDS image [64 k][128 + 8] per operand and buffer; fragments come out transposed (see the file header).
        constexpr int BK = 64, LSM = BM + 8, LSN = BN + 8;  // 272 / 144-B rows: every piece and tr address 8-B aligned
        typedef short v4s __attribute__((ext_vector_type(4)));
        u16* sA = reinterpret_cast<u16*>(smem);            // [2][BK][LSM]
        u16* sB = sA + 2 * BK * LSM;                       // [2][BK][LSN]
        const int krow = tid >> 4, mcol = (tid & 15) * 8;  // A piece (k row, feature) of pass i: k = krow + 16 i
        constexpr int NPB = BN / 8, KPB = 256 / NPB;       // B: pieces per k row, k rows per pass (PB passes)
        const int krowb = tid / NPB, ncol = (tid % NPB) * 8;
        // a piece is fetched when it lies inside the row pitch (pitches are multiples of 8; columns between the valid
        // extent and the pitch are zero by construction: the one-hot label matrix); batch rows >= K read as zeros
        const bool okA = m0 + mcol + 8 <= d.lda, okB = n0 + ncol + 8 <= d.ldb;
        const u16* ga = A + (size_t)krow * d.lda + (okA ? m0 + mcol : 0);
        const u16* gb = B + (size_t)krowb * d.ldb + (okB ? n0 + ncol : 0);
        // TWO k tiles in flight in registers (sets 0 / 1) on top of the two LDS buffers: with one the loop exposed a
        // full memory round trip per 64-row tile (the bf16 MFMAs of a tile take 0.2 us, the fetch ~2 us)
        bf16x8 ra_[2][4], rb_[2][PB];
        const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        auto gload = [&](auto P, int t) {
            constexpr int p = decltype(P)::value;
            const int k0 = t * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool in = k0 + krow + 16 * i < K;
                ra_[p][i] = (in && okA) ? ldgb<bf16x8>(ga + (size_t)(k0 + 16 * i) * d.lda) : zero8;
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const bool in = k0 + krowb + KPB * i < K;
                rb_[p][i] = (in && okB) ? ldgb<bf16x8>(gb + (size_t)(k0 + KPB * i) * d.ldb) : zero8;
            }
        };
        auto lstore = [&](auto P, int buf) {
            constexpr int p = decltype(P)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<bf16x8*>(sA + (buf * BK + krow + 16 * i) * LSM + mcol) = ra_[p][i];
#pragma unroll
            for (int i = 0; i < PB; ++i) *reinterpret_cast<bf16x8*>(sB + (buf * BK + krowb + KPB * i) * LSN + ncol) = rb_[p][i];
        };
        const int nt_all = (K + BK - 1) / BK;
        const int S = d.ksplit > 1 ? d.ksplit : 1;
        const int per = (nt_all + S - 1) / S;
        const int t_beg = ksg * per;
        const int nt = (t_beg + per < nt_all ? t_beg + per : nt_all);  // end (exclusive)
        if (t_beg < nt) gload(I0{}, t_beg);
        if (t_beg + 1 < nt) gload(I1{}, t_beg + 1);
        if (t_beg < nt) lstore(I0{}, 0);  // (waits for the first tile's loads only: vmcnt counts in issue order)
        __syncthreads();
        STAMP();
        // lane (r, q) of the MFMA wants k = 32 s2 + 8 q + 0..7 of feature column r of its block: two transposed reads
        // of 4 k rows each; inside its 16-lane group the lane SUPPLIES the address of (k row r >> 2, features 4 (r & 3)..)
        const int tr_off = (8 * q + (r >> 2)) * LSM + 4 * (r & 3), tr_offb = (8 * q + (r >> 2)) * LSN + 4 * (r & 3);
        auto trd = [&](const u16* p) {
            return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) v4s*)(reinterpret_cast<uintptr_t>(p)));
        };
        // iteration of tile t (parity p = (t - t_beg) & 1): LDS buffer p holds it, register set 1-p holds tile t+1 (in
        // flight since the previous iteration), register set p is free: tile t+2 is requested into it
        auto iter = [&](auto P, int t) {
            constexpr int p = decltype(P)::value;
            typedef std::integral_constant<int, 1 - p> Q;
            const int buf = p;
            if (t + 2 < nt) gload(P, t + 2);
#pragma unroll
            for (int s2 = 0; s2 < BK / 32; ++s2) {
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int m = 0; m < TM; ++m) {
                    const u16* pa = sA + (buf * BK + 32 * s2) * LSM + tr_off + wm * 64 + 16 * m;
                    const v4s lo = trd(pa), hi = trd(pa + 4 * LSM);
                    fa[m] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    const u16* pb = sB + (buf * BK + 32 * s2) * LSN + tr_offb + wn * 16 * TN + 16 * n;
                    const v4s lo = trd(pb), hi = trd(pb + 4 * LSN);
                    fb[n] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
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
            if (t + 1 < nt) lstore(Q{}, buf ^ 1);
            __syncthreads();
            STAMP();
        };
        for (int t = t_beg; t < nt; t += 2) {
            iter(I0{}, t);
            if (t + 1 < nt) iter(I1{}, t + 1);
        }
      } else {
        constexpr int BK = 64, LS = BK + 8;  // LDS row stride in bf16: 144 B keeps ds_read_b128 of 16 rows conflict-free
        u16* sA = reinterpret_cast<u16*>(smem);            // [2][BM][LS]
        u16* sB = sA + 2 * BM * LS;                        // [2][BN][LS]
        // global -> register -> LDS: 16-B pieces, 8 consecutive lanes = one 128-B line of one row
        const int prow = tid >> 3, pk = (tid & 7) * 8;     // piece (row, k) of pass i: row = prow + 32 i
        const u16* ga[4];
        const u16* gb[PB];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ra = m0 + prow + 32 * i;
            ra = ra < M ? ra : M - 1;  // clamped rows only feed outputs the epilogue never stores
            ga[i] = A + (size_t)ra * d.lda + pk;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            int rb = n0 + prow + 32 * i;
            rb = rb < N ? rb : N - 1;
            gb[i] = B + (size_t)rb * d.ldb + pk;
        }
        bf16x8 ra_[2][4], rb_[2][PB];  // two k tiles in flight in registers (see the k-major loop above)
        const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        auto gload = [&](auto P, int t) {
            constexpr int p = decltype(P)::value;
            const int k0 = t * BK;
            const bool in = k0 + pk < K;  // K % 32 == 0: the last tile may hold only its first half
#pragma unroll
            for (int i = 0; i < 4; ++i) ra_[p][i] = in ? ldgb<bf16x8>(ga[i] + k0) : zero8;
#pragma unroll
            for (int i = 0; i < PB; ++i) rb_[p][i] = in ? ldgb<bf16x8>(gb[i] + k0) : zero8;
        };
        auto lstore = [&](auto P, int buf) {
            constexpr int p = decltype(P)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<bf16x8*>(sA + (buf * BM + prow + 32 * i) * LS + pk) = ra_[p][i];
#pragma unroll
            for (int i = 0; i < PB; ++i) *reinterpret_cast<bf16x8*>(sB + (buf * BN + prow + 32 * i) * LS + pk) = rb_[p][i];
        };
        // k tiles of this workgroup: all of them, or slice ksg of a cross-workgroup split (weight gradients)
        const int nt_all = (K + BK - 1) / BK;
        const int S = d.ksplit > 1 ? d.ksplit : 1;
        const int per = (nt_all + S - 1) / S;
        const int t_beg = ksg * per;
        const int nt = (t_beg + per < nt_all ? t_beg + per : nt_all);  // end (exclusive)
        if (t_beg < nt) gload(I0{}, t_beg);
        if (t_beg + 1 < nt) gload(I1{}, t_beg + 1);
        if (t_beg < nt) lstore(I0{}, 0);
        __syncthreads();
        STAMP();
        auto iter = [&](auto P, int t) {
            constexpr int p = decltype(P)::value;
            typedef std::integral_constant<int, 1 - p> Q;
            const int buf = p;
            if (t + 2 < nt) gload(P, t + 2);  // two tiles ahead: in flight under this and the next tile's MFMAs
#pragma unroll
            for (int s2 = 0; s2 < BK / 32; ++s2) {
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int m = 0; m < TM; ++m)
                    fa[m] = *reinterpret_cast<const bf16x8*>(sA + (buf * BM + wm * 64 + 16 * m + r) * LS + 32 * s2 + 8 * q);
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    fb[n] = *reinterpret_cast<const bf16x8*>(sB + (buf * BN + wn * 16 * TN + 16 * n + r) * LS + 32 * s2 + 8 * q);
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
            if (t + 1 < nt) lstore(Q{}, buf ^ 1);  // the other buffer: last read before the previous barrier
            __syncthreads();
            STAMP();
        };
        for (int t = t_beg; t < nt; t += 2) {
            iter(I0{}, t);
            if (t + 1 < nt) iter(I1{}, t + 1);
        }
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
    STAMP();
    // Row loop: EIGHT columns per thread, so that the bf16 output leaves as one 16-byte store per thread and row (whole
    // 128-B lines per 8 lanes) and the fp32 outputs as two; there is no second (transposed) bf16 copy any more.
    const int epi = d.epi;
    constexpr int CQ = BN / 8;
    constexpr int RSTEP = NT / CQ;
    const int c8 = tid % CQ, r0 = tid / CQ;
    const int col = n0 + 8 * c8;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bias4[2] = {zero4, zero4};
    if (d.bias) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + 4 * h + e < N) bias4[h][e] = ldgb<float>(d.bias + col + 4 * h + e);
    }
    f32x4 ct[2][6];  // (!CT instantiations only) this thread's 8 columns of the conditional table
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 6; ++c) ct[h][c] = zero4;
    if constexpr (!CT) {
        if (d.ctab) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int c = 0; c < 6; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + 4 * h + e < N) ct[h][c][e] = ldgb<float>(d.ctab + (size_t)c * N + col + 4 * h + e);
        }
    }
    float sq = 0.f;
    const float gsc = (epi == BEPI_RECON) ? d.scale * (d.go ? ldgb<float>(d.go) : 1.0f) : 0.f;
    const bool colfull = col + 7 < N;
    // not unrolled: measured 211 / 213 / 213 / 218 us per cfg3 step at unroll 1 / 2 / 4 / 8 -- the loop is bound by the
    // store burst of all workgroups at once (in-kernel stamps: 9-17 k cycles of a 29-36 k cycle workgroup), not by latency
#pragma unroll 1
    for (int rr = 0; rr < BM; rr += RSTEP) {
        const int lr = rr + r0;
        const int row = m0 + lr;
        if (lr >= BM || row >= M || col >= N) continue;
        f32x4 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v[h] = *reinterpret_cast<const f32x4*>(smem + lr * SC + 8 * c8 + 4 * h);
#pragma unroll
            for (int k2 = 1; k2 < KS; ++k2) v[h] += *reinterpret_cast<const f32x4*>(smem + k2 * (BM * SC) + lr * SC + 8 * c8 + 4 * h);
            v[h] += bias4[h];
        }
        if (d.ctab) {  // one-hot conditioning: + table[gender] + table[3 + age]
            const int g = d.gender[row], a = d.age[row];
            if constexpr (CT) {
                const int gi = (unsigned)g < 3u ? g : 6, ai = (unsigned)a < 3u ? 3 + a : 6;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v[h] += *reinterpret_cast<const f32x4*>(ct_s + gi * BN + 8 * c8 + 4 * h);
                    v[h] += *reinterpret_cast<const f32x4*>(ct_s + ai * BN + 8 * c8 + 4 * h);
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v[h] += (g == 0 ? ct[h][0] : g == 1 ? ct[h][1] : g == 2 ? ct[h][2] : zero4);
                    v[h] += (a == 0 ? ct[h][3] : a == 1 ? ct[h][4] : a == 2 ? ct[h][5] : zero4);
                }
            }
        }
        if (epi == BEPI_RELU) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[h][e] = fmaxf(v[h][e], 0.f);
        } else if (epi == BEPI_MASK) {  // ReLU backward: mask = saved bf16 activation [M, ldmask]
            const u16* mp = d.mask + (size_t)row * d.ldmask + col;
            if (colfull && ((d.ldmask & 7) == 0)) {
                const bf16x8 mk = ldgb<bf16x8>(mp);  // 8 bf16 = 16 B
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[h][e] = bf2f((u16)mk[4 * h + e]) > 0.f ? v[h][e] : 0.f;
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + 4 * h + e < N) v[h][e] = bf2f(ldgb<u16>(mp + 4 * h + e)) > 0.f ? v[h][e] : 0.f;
            }
        }
        if (d.Cf) {  // fp32 output (final outputs, wgrads)
            float* p = d.Cf + (size_t)ksg * M * d.ldcf + (size_t)row * d.ldcf + col;  // ksplit: this slice's slab
            if (colfull && ((d.ldcf & 3) == 0)) {
                stgb<f32x4>(p, v[0]);  // (nontemporal stores for the split-K slabs were measured: no difference)
                stgb<f32x4>(p + 4, v[1]);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + 4 * h + e < N) stgb<float>(p + 4 * h + e, v[h][e]);
            }
        }
        if (epi == BEPI_RECON) {  // v = x_hat: emit the MSE gradient instead of v into the bf16 output
            const float* tp = d.target + (size_t)row * d.ldcf + col;
            f32x4 xt[2] = {zero4, zero4};
            if (colfull && ((d.ldcf & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.target) & 15) == 0)) {
                xt[0] = ldgb<f32x4>(tp);
                xt[1] = ldgb<f32x4>(tp + 4);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + 4 * h + e < N) xt[h][e] = ldgb<float>(tp + 4 * h + e);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float df = (col + 4 * h + e < N) ? v[h][e] - xt[h][e] : 0.f;
                    sq += df * df;
                    v[h][e] = gsc * df;
                }
        }
        if (d.Cb) {
            u16* p = d.Cb + (size_t)row * d.ldcb + col;
            if (colfull && ((d.ldcb & 7) == 0)) {  // 8 bf16 = one 16-byte store
                bf16x8 pk;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[4 * h + e] = (short)f2bf(v[h][e]);
                stgb<bf16x8>(p, pk);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + 4 * h + e < N) stgb<u16>(p + 4 * h + e, f2bf(v[h][e]));
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
#ifdef PSVAE_STAMP
    STAMP();
    if (STAGED && tid == 0 && tm == 0 && tn == 0 && ksg == 0) {
        printf("stamp tr=%d M=%d N=%d K=%d ks=%d:", d.tr, M, N, K, d.ksplit);
        for (int i = 1; i < nst_; ++i) printf(" %llu", st_[i] - st_[i - 1]);
        printf("\n");
    }
#endif
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
    if (batch.xcd_remap) {
        // XCD-aware order (speed only): workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  Give XCD
        // x the CONTIGUOUS tile range [x T/8, (x+1) T/8) of this problem: tiles that share an operand panel (the column
        // tiles of one 128-row panel; the 16 tiles of one batch slice of a weight gradient) then share one L2 instead
        // of every L2 pulling every panel from Infinity Cache -- at B = 4096 these launches are bound by exactly that
        // traffic.  Bijective for any T (cdna guide T1).
        const int T = tiles_mn * ((STAGED && d.ksplit > 1) ? d.ksplit : 1), q8 = T >> 3, r8 = T & 7;
        const int xcd = lt & 7, idx = lt >> 3;
        lt = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int ksg = STAGED ? lt / tiles_mn : 0;  // cross-workgroup k slice (staged kernel only)
    lt -= ksg * tiles_mn;
    const int tm = lt / d.tiles_n, tn = lt - tm * d.tiles_n;
    gemm_tile_bf16<TM, TN, WM, WN, KS, D, STAGED, CT>(d, tm, tn, ksg, smem);
}

template <int TM, int TN, int WM, int WN, int KS, int D, bool STAGED, bool CT>
static hipError_t launch_bf_ct(GemmBatchBf& b, hipStream_t stream) {
    constexpr int BM = WM * 16 * TM, BN = WN * 16 * TN;
    constexpr size_t lds_epi = (size_t)(KS * BM * (BN + 4) + KS * BM + 8 + (CT ? 7 * BN : 0)) * sizeof(float);  // CT: + the table tile
    // two buffers of A and B panels; the k-major image ([64 k][BM + 8] + [64 k][BN + 8]) is never larger than this
    constexpr size_t lds_stage = STAGED ? (size_t)2 * (BM + BN) * (64 + 8) * 2 : 0;
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
    {
        const char* e = getenv("PSVAE_BF16_XCD");  // tuning (read per call: tests flip it in-process)
        b.xcd_remap = e ? atoi(e) : 1;
    }
    static std::atomic<unsigned long long> lds_ok{0};  // per device: psvae_common.h lds_opt_in
    if (hipError_t attr_err = lds_opt_in(lds_ok, reinterpret_cast<const void*>(&gemm_bf16_kernel<TM, TN, WM, WN, KS, D, STAGED, CT>), (int)(lds > 96 * 1024 ? lds : 96 * 1024)))
        return attr_err;
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
    if (b.force_staged) {  // the backward launches (see the API)
        // a launch that gives a 128x128 tiling at most one workgroup per CU runs 128x64 tiles instead: two workgroups
        // share a CU, so one's store burst and LDS exchange run under the other's k loop
        long t88s = 0;
        for (int i = 0; i < b.n; ++i)
            t88s += (long)((b.d[i].M + 127) / 128) * ((b.d[i].N + 127) / 128) * (b.d[i].ksplit > 1 ? b.d[i].ksplit : 1);
        static const int half_max = getenv("PSVAE_BF16_HALF_MAX") ? atoi(getenv("PSVAE_BF16_HALF_MAX")) : 256;
        if (t88s <= half_max) return launch_bf<4, 2, 2, 2, 1, 2, true>(b, stream);
        return launch_bf<4, 4, 2, 2, 1, 2, true>(b, stream);
    }
    static const int t88_min = getenv("PSVAE_BF16_T88_MIN") ? atoi(getenv("PSVAE_BF16_T88_MIN")) : 128;
    if (t88_min > 0 && t88 >= t88_min && kmax <= 1024) {  // a full machine of 128x128 tiles: half the operand bytes of 64x64
        static const int fwd_half_max = getenv("PSVAE_BF16_FWD_HALF_MAX") ? atoi(getenv("PSVAE_BF16_FWD_HALF_MAX")) : 256;
        if (t88 <= fwd_half_max) return launch_bf<4, 2, 2, 2, 1, 2, true>(b, stream);  // (two workgroups per CU, as above)
        return launch_bf<4, 4, 2, 2, 1, 2, true>(b, stream);
    }
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
    // optional rider: the one-hot label matrix oh[B][8] (bf16 0/1; columns 6, 7 zero), blocks past the cast tiles
    int cast_tiles;
    const int32_t* gender;
    const int32_t* age;
    u16* oh;
    int B;
};

__global__ __launch_bounds__(256) void cast_transpose_kernel(const CastArgs a) {
    __shared__ float tile[32][33];
    if ((int)blockIdx.x >= a.cast_tiles) {  // one-hot rider blocks (one launch fewer per step): one 16-B row per thread
        const int b = ((int)blockIdx.x - a.cast_tiles) * 256 + threadIdx.x;
        if (b < a.B) {
            const int g = a.gender[b], ag = a.age[b];
            bf16x8 row = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                row[c] = (g == c) ? (short)0x3F80 : (short)0;
                row[3 + c] = (ag == c) ? (short)0x3F80 : (short)0;
            }
            *reinterpret_cast<bf16x8*>(a.oh + (size_t)b * 8) = row;
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
                                 const int32_t* age, u16* oh, int B) {
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
    if (oh) {
        a.gender = gender; a.age = age; a.oh = oh; a.B = B;
        t += (B + 255) / 256;
    }
    hipLaunchKernelGGL(cast_transpose_kernel, dim3(t), dim3(256), 0, s, a);
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
