// Grouped fp32 MFMA GEMM for gfx950 (MI355X): the building block of every Linear layer of the VAE
// (forward NT, dgrad NN, wgrad TN; reference: nn.Linear at /root/reference/ps_vae/model.py:14-36 and
// its autograd).  Exact fp32: v_mfma_f32_16x16x4_f32, which is a k-ordered fmaf chain.
//
// Design (see DESIGN.md "GEMM kernel"):
//   * one launch = a GROUP of independent problems (both encoders; a layer's wgrad + dgrad), so a
//     1024-row batch still fills 256 CUs; each 256-thread workgroup (4 waves, 2x2) owns one
//     (32*TM) x (32*TN) output tile, each wave a (16*TM) x (16*TN) sub-tile of 16x16 MFMA blocks;
//   * operands are staged global -> registers -> LDS in their NATIVE orientation, 16 B per lane,
//     full 128-B lines per row, double-buffered (loads of tile k+1 fly under the MFMAs of tile k):
//       "KC" operand (k contiguous in memory: activations / weights in NT, dY in NN): LDS tile
//            [rows][40] floats, fragment = one ds_read_b128 per 16-k chunk (4 k's per lane, used
//            by 4 consecutive MFMAs; the k permutation is applied identically to A and B);
//       "KM" operand (k-major: W in NN, both operands of the batch-contracting wgrad): LDS tile
//            [32][rows(+pad)], fragment = one ds_read_b{32,64,128} per k giving T adjacent rows
//            = the T MFMA blocks of the wave (row slot c of block e  <->  row T*c + e);
//     strides 40 / R+{4,8,0} make every fragment read bank-conflict-free (tools/lds_bank_check.py);
//   * epilogues fused: +bias, ReLU, ReLU-mask (dgrad), bias-gradient column sums (wgrad, from the
//     A fragments already in registers: VALU adds beside the MFMAs, no extra pass over dY).
#include "psvae_common.h"

namespace psvae {

constexpr int BK = 32;    // k per LDS tile
constexpr int S_KC = 40;  // KC tile row stride in floats (conflict-free ds_read_b128 along k)
constexpr int NTHR = 256;

__host__ __device__ constexpr int km_stride(int R, int T) { return T == 4 ? R : (T == 2 ? R + 8 : R + 4); }
__host__ __device__ constexpr int tile_floats(bool km, int R, int T) { return km ? BK * km_stride(R, T) : R * S_KC; }
__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }
// LDS floats for one (A,B) pair, worst case over the three kinds
__host__ __device__ constexpr int pair_floats(int TM, int TN) {
    return cmax(tile_floats(false, 32 * TM, TM), tile_floats(true, 32 * TM, TM)) +
           cmax(tile_floats(false, 32 * TN, TN), tile_floats(true, 32 * TN, TN));
}

template <int T> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<2> { typedef f32x2 type; };
template <> struct VecT<4> { typedef f32x4 type; };

template <int T> __device__ __forceinline__ float vget(const typename VecT<T>::type& v, int i);
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }
template <> __device__ __forceinline__ float vget<2>(const f32x2& v, int i) { return v[i]; }
template <> __device__ __forceinline__ float vget<4>(const f32x4& v, int i) { return v[i]; }

// guarded 4-float global load: (row_ok, c..c+3 < cmaxv) else zero fill
__device__ __forceinline__ f32x4 gload4(const float* __restrict__ p, int c, int cmaxv, bool row_ok, bool vec_ok) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row_ok) {
        if (vec_ok && c + 3 < cmaxv) {
            v = *reinterpret_cast<const f32x4*>(p);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < cmaxv) v[e] = p[e];
        }
    }
    return v;
}

// ---- staging: one operand tile, global -> registers ------------------------------------------
// KC: tile rows r0..r0+R-1 (bounded by rmax), k0..k0+31 (bounded by K); memory [row][k], ld
// KM: tile k rows k0..k0+31 (bounded by K), cols r0..r0+R-1 (bounded by rmax); memory [k][row], ld
template <int R, bool KM>
__device__ __forceinline__ void stage_load(f32x4 (&reg)[R / 32], const float* __restrict__ base, int ld, int r0,
                                           int rmax, int k0, int K, bool vec_ok, int tid) {
#pragma unroll
    for (int j = 0; j < R / 32; ++j) {
        const int i = tid + NTHR * j;
        if (!KM) {
            const int row = i >> 3, kq = i & 7;
            const int gr = r0 + row, gk = k0 + 4 * kq;
            reg[j] = gload4(base + (size_t)gr * ld + gk, gk, K, gr < rmax, vec_ok);
        } else {
            constexpr int Q = R / 4;
            const int krow = i / Q, cq = i % Q;
            const int gk = k0 + krow, gc = r0 + 4 * cq;
            reg[j] = gload4(base + (size_t)gk * ld + gc, gc, rmax, gk < K, vec_ok);
        }
    }
}

template <int R, int T, bool KM>
__device__ __forceinline__ void stage_store(const f32x4 (&reg)[R / 32], float* __restrict__ s, int tid) {
#pragma unroll
    for (int j = 0; j < R / 32; ++j) {
        const int i = tid + NTHR * j;
        if (!KM) {
            const int row = i >> 3, kq = i & 7;
            *reinterpret_cast<f32x4*>(s + row * S_KC + 4 * kq) = reg[j];
        } else {
            constexpr int Q = R / 4;
            constexpr int S = km_stride(R, T);
            const int krow = i / Q, cq = i % Q;
            *reinterpret_cast<f32x4*>(s + krow * S + 4 * cq) = reg[j];
        }
    }
}

// ---- fragment reads for one 16-k chunk: f[t][blk], t = MFMA step (k = 16c + 4q + t) ------------
template <int R, int T, bool KM>
__device__ __forceinline__ void read_frags(float (&f)[4][T], const float* __restrict__ s, int wave_off, int chunk,
                                           int r, int q) {
    if (!KM) {
#pragma unroll
        for (int b = 0; b < T; ++b) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(s + (wave_off + 16 * b + r) * S_KC + 16 * chunk + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) f[t][b] = v[t];
        }
    } else {
        constexpr int S = km_stride(R, T);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const typename VecT<T>::type v =
                *reinterpret_cast<const typename VecT<T>::type*>(s + (16 * chunk + 4 * q + t) * S + wave_off + T * r);
#pragma unroll
            for (int b = 0; b < T; ++b) f[t][b] = vget<T>(v, b);
        }
    }
}

// actual index (inside the wave's 16*T span) of MFMA slot `slot` (0..15) of block `blk`
template <int T, bool KM>
__device__ __forceinline__ int slot_index(int blk, int slot) { return KM ? T * slot + blk : 16 * blk + slot; }

template <int TM, int TN, bool AKM, bool BKM>
__device__ __forceinline__ void gemm_tile(const GemmDesc& d, int tm, int tn, float* __restrict__ smem) {
    constexpr int RA = 32 * TM, RB = 32 * TN;
    constexpr int FA = tile_floats(AKM, RA, TM), FB = tile_floats(BKM, RB, TN);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    const int M = d.M, N = d.N, K = d.K;
    const int m0 = tm * RA, n0 = tn * RB;
    const float* __restrict__ A = d.A;
    const float* __restrict__ B = d.B;
    const bool vecA = ((d.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool vecB = ((d.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);


    f32x4 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float csum[TM];
#pragma unroll
    for (int m = 0; m < TM; ++m) csum[m] = 0.f;
    const bool do_colsum = AKM && (d.colsum != nullptr) && (tn == 0) && (wn == 0);

    f32x4 ra[RA / 32], rb[RB / 32];
    const int nk = (K + BK - 1) / BK;
    stage_load<RA, AKM>(ra, A, d.lda, m0, M, 0, K, vecA, tid);
    stage_load<RB, BKM>(rb, B, d.ldb, n0, N, 0, K, vecB, tid);
    stage_store<RA, TM, AKM>(ra, smem, tid);
    stage_store<RB, TN, BKM>(rb, smem + FA, tid);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const float* __restrict__ curA = smem + (kt & 1) * (FA + FB);
        const float* __restrict__ curB = curA + FA;
        float* __restrict__ nxtA = smem + ((kt + 1) & 1) * (FA + FB);
        const bool more = kt + 1 < nk;
        if (more) {
            stage_load<RA, AKM>(ra, A, d.lda, m0, M, (kt + 1) * BK, K, vecA, tid);
            stage_load<RB, BKM>(rb, B, d.ldb, n0, N, (kt + 1) * BK, K, vecB, tid);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float fa[4][TM], fb[4][TN];
            read_frags<RA, TM, AKM>(fa, curA, wm * 16 * TM, c, r, q);
            read_frags<RB, TN, BKM>(fb, curB, wn * 16 * TN, c, r, q);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t][m], fb[t][n], acc[m][n], 0, 0, 0);
            }
            if (do_colsum) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int m = 0; m < TM; ++m) csum[m] += fa[t][m];
            }
        }
        if (more) {
            stage_store<RA, TM, AKM>(ra, nxtA, tid);
            stage_store<RB, TN, BKM>(rb, nxtA + FA, tid);
        }
        __syncthreads();
    }

    // ---- epilogue: acc[m][n][j] is (row slot 4q+j, col slot r) of block (m,n) ----
    const int epi = d.epi;
    float* __restrict__ C = d.C;
    const float* __restrict__ aux = d.aux;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = m0 + wm * 16 * TM + slot_index<TM, AKM>(m, 4 * q + j);
            if (row >= M) continue;
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const int col = n0 + wn * 16 * TN + slot_index<TN, BKM>(n, r);
                if (col >= N) continue;
                float v = acc[m][n][j];
                const size_t o = (size_t)row * d.ldc + col;
                if (epi == EPI_BIAS || epi == EPI_BIAS_RELU) {
                    v += aux[col];
                    if (d.ctab) {
                        const int g = d.gender[row], a = d.age[row];
                        if (g >= 0 && g < 3) v += d.ctab[(size_t)g * N + col];
                        if (a >= 0 && a < 3) v += d.ctab[(size_t)(3 + a) * N + col];
                    }
                    if (epi == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                } else if (epi == EPI_MASK) {
                    v = aux[o] > 0.f ? v : 0.f;
                } else if (epi == EPI_ACCUM) {
                    v += C[o];
                }
                C[o] = v;
            }
        }
    }
    if (do_colsum) {
        // lanes with equal r hold partial sums over k = 4q+t (q = lane>>4): reduce over q
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            float s = csum[m];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const int row = m0 + wm * 16 * TM + slot_index<TM, true>(m, r);
            if (q == 0 && row < M) d.colsum[row] = s;
        }
    }
}

template <int TM, int TN>
__global__ __launch_bounds__(NTHR) void gemm_grouped_kernel(const GemmBatch batch) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tile = blockIdx.x;
    int di = 0;
    for (int i = 1; i < batch.n; ++i)
        if (tile >= batch.d[i].tile_begin) di = i;
    const GemmDesc& d = batch.d[di];
    const int lt = tile - d.tile_begin;
    const int tm = lt / d.tiles_n, tn = lt - tm * d.tiles_n;
    if (d.kind == GK_NT)
        gemm_tile<TM, TN, false, false>(d, tm, tn, smem);
    else if (d.kind == GK_NN)
        gemm_tile<TM, TN, false, true>(d, tm, tn, smem);
    else
        gemm_tile<TM, TN, true, true>(d, tm, tn, smem);
}

// ---- host side ----------------------------------------------------------------------------------
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

static int count_tiles(const GemmBatch& b, int TM, int TN) {
    int t = 0;
    for (int i = 0; i < b.n; ++i) t += ((b.d[i].M + 32 * TM - 1) / (32 * TM)) * ((b.d[i].N + 32 * TN - 1) / (32 * TN));
    return t;
}

int pick_tile(const GemmBatch& b) {
    static const int min_tiles = env_int("PSVAE_MIN_TILES", 384);
    static const int forced = env_int("PSVAE_TILE", 0);
    if (forced) return forced;
    int maxM = 0, maxN = 0;
    for (int i = 0; i < b.n; ++i) {
        maxM = b.d[i].M > maxM ? b.d[i].M : maxM;
        maxN = b.d[i].N > maxN ? b.d[i].N : maxN;
    }
    static const int cand[][2] = {{4, 4}, {4, 2}, {2, 4}, {2, 2}, {2, 1}, {1, 2}, {1, 1}};
    for (auto& c : cand) {
        const int TM = c[0], TN = c[1];
        if (32 * TM > ((maxM + 31) / 32) * 32 || 32 * TN > ((maxN + 31) / 32) * 32) continue;
        if (count_tiles(b, TM, TN) >= min_tiles) return (TM << 4) | TN;
    }
    return (1 << 4) | 1;
}

void finalize_batch(GemmBatch& b, int tile) {
    const int TM = tile >> 4, TN = tile & 15;
    int t = 0;
    for (int i = 0; i < b.n; ++i) {
        b.d[i].tiles_m = (b.d[i].M + 32 * TM - 1) / (32 * TM);
        b.d[i].tiles_n = (b.d[i].N + 32 * TN - 1) / (32 * TN);
        b.d[i].tile_begin = t;
        t += b.d[i].tiles_m * b.d[i].tiles_n;
    }
    b.total_tiles = t;
}

template <int TM, int TN>
static hipError_t launch_cfg(const GemmBatch& b, hipStream_t stream) {
    constexpr size_t lds = 2 * pair_floats(TM, TN) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_grouped_kernel<TM, TN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_grouped_kernel<TM, TN>), dim3(b.total_tiles), dim3(NTHR), lds, stream, b);
    return hipGetLastError();
}

hipError_t launch_gemm_batch(GemmBatch& b, int tile, hipStream_t stream) {
    if (b.n <= 0) return hipSuccess;
    if (tile == 0) tile = pick_tile(b);
    finalize_batch(b, tile);
    if (b.total_tiles == 0) return hipSuccess;
    switch (tile) {
        case 0x44: return launch_cfg<4, 4>(b, stream);
        case 0x42: return launch_cfg<4, 2>(b, stream);
        case 0x24: return launch_cfg<2, 4>(b, stream);
        case 0x22: return launch_cfg<2, 2>(b, stream);
        case 0x21: return launch_cfg<2, 1>(b, stream);
        case 0x12: return launch_cfg<1, 2>(b, stream);
        case 0x11: return launch_cfg<1, 1>(b, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace psvae
