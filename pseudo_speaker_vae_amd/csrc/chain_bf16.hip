// Row-chain kernels of the bf16 training pass (BASELINE config 3): a 3-layer MLP -- one encoder (model.py:14-27) or the
// decoder (model.py:30-36,58-61) -- over a tile of 64 batch rows in ONE launch, organised like the fused bf16 decode
// (decode_fused.hip, second half): the four wavefronts of a workgroup own FEATURE slices, the tile's activations are
// shared through one LDS image [k step][row group][lane][8 bf16], weights go straight from L2 to registers.  The
// per-layer path (gemm_bf16.hip) runs each layer as a launch of 128 x 64 tiles: at B = 4096 that is one round of
// workgroups per launch whose fetch / MFMA / store-burst phases cannot overlap, and every hidden activation is written
// and immediately re-read through memory (the forward was 76 of the step's 192 us).  Here a hidden activation leaves the
// chip once -- as the row-major bf16 copy the backward's weight gradients and ReLU masks need, staged through LDS so
// that every store instruction writes whole rows -- and is never read back in the forward.
//
//   layer 0: in [B, K0] fp32 (x or z; rounded to bf16 on load exactly as the cast / reparam kernels do)
//            -> 512, + bias (+ table[gender] + table[3 + age]: the one-hot conditioning, fp32), ReLU -> save[0], image
//   layer 1: 512 -> 512, + bias, ReLU -> save[1], image
//   layer 2: 512 -> NOUT, + bias -> out [B, NOUT] fp32            (mu / log sigma^2 heads: NOUT = Z)
//            decoder: out = x_hat, and in the same pass over the staged rows the MSE gradient
//            gy = bf16(scale * go * (x_hat - x)) and the tile's sum of squared errors (lightning.py:113, BEPI_RECON's job)
// Same rounding points as the per-layer path; sums in a different order (the bias is the accumulator's initial value),
// so a hidden activation near a rounding boundary may land on the other bf16 neighbour.
// Further down: chain_pack_kernel (the weights in fragment order, plain and transposed, + the x cast and the one-hot matrix)
// and chain_bwd_bf16_kernel (the input-gradient path of an MLP, same organisation).  A training pass is then 7 launches:
// pack, encoders, decoder, decoder gradient chain, encoder gradient chains, all weight gradients, slab sums
// (psvae_api_bf16.hip).
#include <type_traits>

#include "psvae_common.h"
#include "psvae_device.h"

#ifndef PSVAE_CHAIN_NA
#define PSVAE_CHAIN_NA 4  // ring depth of the weight fragments at 16- / 32-row tiles: 2 / 3 / 4 / 5 measured 96.0 / 94.7 / 94.1 / 94.5 us per
                          // cfg3 step (round 2, 32-row tiles everywhere: 3 / 4 / 5 = 121.6 / 122.0 / 121.7)
#endif

namespace psvae {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 nbf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

namespace {

template <bool RELU>
__device__ __forceinline__ bf16x8 pack8(const f32x4& lo, const f32x4& hi) {
    union { bf16x8 v; s16x2 p[4]; } u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x4& s = h ? hi : lo;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            nbf16x2 b = __builtin_convertvector(f32x2{s[2 * t], s[2 * t + 1]}, nbf16x2);
            s16x2 v = *reinterpret_cast<s16x2*>(&b);
            if (RELU) v = __builtin_elementwise_max(v, s16x2{0, 0});
            u.p[2 * h + t] = v;
        }
    }
    return u.v;
}

constexpr int SROW = 1024 + 16;          // staged row: 512 bf16 (or 256 fp32) + 16 B so the 16 r lanes hit different banks

}  // namespace

// Wp[wavefront w][k step kb][block mi][lane][e] = bf16(W[16 (MBW w + mi) + r][32 kb + 16 (e / 4) + 4 q + (e % 4)]),
// W [OUT, IN] fp32 row-major, MBW = OUT / 64; grid.y = matrix
// T = 1: the packed matrix is W^T (the backward chains' A operand): Wp[..] = bf16(W[k slot][16 (MBW w + mi) + r]), MBW = IN / 64
struct ChainPackArgs {
    const float* W[16];
    u16* Wp[16];
    int OUT[16], IN[16], T[16];
    int n;
    // riders (grid.y = n, n + 1): x -> bf16 row-major (the weight gradients' operand) and the one-hot label matrix [B][8]
    const float* x;
    u16* xb;
    long nx;
    const int32_t* gender;
    const int32_t* age;
    u16* oh;
    int B;
};

__global__ __launch_bounds__(256) void chain_pack_kernel(const ChainPackArgs a) {
    if ((int)blockIdx.y == a.n) {  // x cast, 8 elements per thread and step
        for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < a.nx; i += (long)gridDim.x * 256 * 8)
            *reinterpret_cast<bf16x8*>(a.xb + i) =
                pack8<false>(*reinterpret_cast<const f32x4*>(a.x + i), *reinterpret_cast<const f32x4*>(a.x + i + 4));
        return;
    }
    if ((int)blockIdx.y == a.n + 1) {
        for (int b = blockIdx.x * 256 + threadIdx.x; b < a.B; b += gridDim.x * 256) {
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
    const float* __restrict__ W = a.W[blockIdx.y];
    u16* __restrict__ Wp = a.Wp[blockIdx.y];
    const int OUT = a.OUT[blockIdx.y], IN = a.IN[blockIdx.y];
    const bool T = a.T[blockIdx.y] != 0;
    const int N = T ? IN : OUT, K = T ? OUT : IN;  // the packed operand is [N, K]
    const int MBW = N / 64, KB = K / 32;
    const long np = (long)OUT * IN / 8;  // 16-byte pieces: one lane's fragment each
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < np; i += (long)gridDim.x * 256) {
        const int r = (int)(i & 15), q = (int)((i >> 4) & 3);
        const long t = i >> 6;
        const int mi = (int)(t % MBW), kb = (int)((t / MBW) % KB), w = (int)(t / ((long)MBW * KB));
        const int n = 16 * (MBW * w + mi) + r, k0 = 32 * kb + 4 * q;
        f32x4 lo, hi;
        if (!T) {  // two 16-byte reads of row n
            const float* src = W + (size_t)n * IN + k0;
            lo = *reinterpret_cast<const f32x4*>(src);
            hi = *reinterpret_cast<const f32x4*>(src + 16);
        } else {   // column n of W: 8 strided reads (the 16 r lanes of a fragment read 16 consecutive floats of a row)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[e] = W[(size_t)(k0 + e) * IN + n];
                hi[e] = W[(size_t)(k0 + 16 + e) * IN + n];
            }
        }
        *reinterpret_cast<bf16x8*>(Wp + 8 * i) = pack8<false>(lo, hi);
    }
}

hipError_t launch_chain_pack(const float* const* W, u16* const* Wp, const int* OUT, const int* IN, const int* T, int n,
                             const float* x, u16* xb, long nx, const int32_t* gender, const int32_t* age, u16* oh, int B,
                             hipStream_t s) {
    if (n > 16 || (nx & 7)) return hipErrorInvalidValue;
    ChainPackArgs a{};
    for (int k = 0; k < n; ++k) { a.W[k] = W[k]; a.Wp[k] = Wp[k]; a.OUT[k] = OUT[k]; a.IN[k] = IN[k]; a.T[k] = T ? T[k] : 0; }
    a.n = n; a.x = x; a.xb = xb; a.nx = nx; a.gender = gender; a.age = age; a.oh = oh; a.B = B;
    // (x == nullptr: weights only -- the training pass's encoder chain writes xb and the one-hot rows itself)
    hipLaunchKernelGGL(chain_pack_kernel, dim3(128, n + (x ? 1 + (oh ? 1 : 0) : 0)), dim3(256), 0, s, a);
    return hipGetLastError();
}

#ifdef PSVAE_STAMP
__device__ unsigned long long g_cstamp[2][8];
#define CSTAMP(K) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_[K]))
#else
#define CSTAMP(K) do {} while (0)
#endif
#if defined(PSVAE_STAMP) && defined(PSVAE_STAMP_PRO)  // diagnostic: slots 2..6 re-used for the forward chain's prologue
#define PSTAMP(K) CSTAMP(K)
#define LSTAMP(K) do {} while (0)
#else
#define PSTAMP(K) do {} while (0)
#define LSTAMP(K) CSTAMP(K)
#endif

template <int KB0, int MB2, bool RECON, int RG>  // K0 = 32 KB0 inputs, NOUT = 64 MB2 outputs, 16 RG rows per tile
__global__ __launch_bounds__(256, 1) void chain_fwd_bf16_kernel(const ChainBatch batch) {
    constexpr int TR = 16 * RG;                      // rows per tile
    constexpr int IMG = 16 * RG * 1024;              // activation image: 16 k steps of a 512-feature activation
    constexpr int STG = TR * SROW;
#ifdef PSVAE_STAMP
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    CSTAMP(0);
    constexpr int HK = H / 32;                       // 16 k steps of a hidden input
    constexpr int S = KB0 + HK + HK;                 // k steps of the weight stream
    constexpr int NA = RG <= 2 ? PSVAE_CHAIN_NA : 3;  // weight-fragment ring: this k step and NA - 1 ahead (registers allow more at 32 rows)
    const ChainArgs& a = batch.c[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* act = reinterpret_cast<char*>(lds);        // image [16][RG][64][16 B]
    char* stg = act + IMG;                           // staged rows
    float* sbias = reinterpret_cast<float*>(stg + STG);  // [512 + 512 + NOUT]
    float* stab = sbias + 2 * H + 64 * MB2;          // [7][512]: the conditional table, row 6 = 0 (label out of range)
    float* red = stab + 7 * H;                        // [4] (no static LDS: the dynamic request is sized against the 160 KB limit)
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long tile = blockIdx.x;
    const bool has_tab = a.ctab != nullptr;
    // ---- weights: this wavefront's fragments, ring of three k steps -------------------------------------------------
    auto mb_of = [&](int s) -> int { return s < KB0 + HK ? 8 : MB2; };
    typedef const __attribute__((address_space(1))) u16 gu16;
    typedef const __attribute__((address_space(1))) bf16x8 gbf16x8;
    const u16* wl0 = a.wp[0] + (size_t)wave * (KB0 * 8 * 512) + lane * 8;
    const u16* wl1 = a.wp[1] + (size_t)wave * (HK * 8 * 512) + lane * 8;
    const u16* wl2 = a.wp[2] + (size_t)wave * (HK * MB2 * 512) + lane * 8;
    bf16x8 wa[NA][8];
    gu16* wsrc = nullptr;
    auto wfetch = [&](int s) {  // in stream order
        if (s == 0) wsrc = (gu16*)wl0;
        if (s == KB0) wsrc = (gu16*)wl1;
        if (s == KB0 + HK) wsrc = (gu16*)wl2;
        asm volatile("" : "+v"(wsrc));
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
            if (mi < mb_of(s)) wa[s % NA][mi] = *(gbf16x8*)(wsrc + mi * 512);
        wsrc += mb_of(s) * 512;
    };
    bf16x8 bb[2][RG];
    auto bfetch = [&](int set, int kb) {
        const char* src = act + kb * (RG * 1024) + lane * 16;
#pragma unroll
        for (int g = 0; g < RG; ++g) bb[set][g] = *reinterpret_cast<const bf16x8*>(src + 1024 * g);
    };
#pragma unroll
    for (int s0 = 0; s0 < NA - 1; ++s0) wfetch(s0);

    // ---- prologue: every global read goes out before the first one is consumed (one memory round trip, not five) -----
    constexpr int NOUT = 64 * MB2;
    constexpr int OROW = NOUT * 4 + 16;                  // staged fp32 output row
    // output rows per staging pass (NOUT = 512: two passes of 16 / 32 rows at 32 / 64-row tiles; a 16-row tile stages its one
    // row group across the activation image -- dead by then -- and the staging buffer behind it: exactly 16 rows of 2,064 B)
    constexpr bool WIDE1 = RG == 1 && OROW * TR > STG;
    constexpr int RPP = (OROW * TR <= STG || WIDE1) ? TR : TR / 2;
    static_assert(OROW * RPP <= STG + (WIDE1 ? IMG : 0), "staging buffer too small for the output rows");
    static_assert(RPP % 16 == 0, "a staging pass holds whole row groups");
    constexpr int PPR = NOUT * 4 / 16;                   // 16-byte pieces per output row
    constexpr int NPC = (RPP * PPR + 255) / 256;         // pieces per thread and pass
    // decoder: the target rows, piece by piece as the output pass walks them -- requested FIRST: loads return in order, so
    // anything requested after an HBM read waits for it; up front that wait coincides with the input rows'
    f32x4 tv[RECON ? (TR / RPP) * NPC : 1];
    if (RECON && !a.raw_out) {
#pragma unroll
        for (int ps = 0; ps < TR / RPP; ++ps)
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const int pc = 256 * i + tid;
                long row = tile * TR + ps * RPP + pc / PPR;
                row = row < a.B ? row : a.B - 1;
                tv[ps * NPC + i] = *reinterpret_cast<const f32x4*>(a.target + (size_t)row * NOUT + (pc % PPR) * 4);
            }
    }
    int tg[RG], ta[RG];                                  // table rows of this lane's rows (one per row group)
    int lg[RG], la[RG];                                  // (the labels themselves, for the one-hot rows below)
    if (has_tab) {
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            long row = tile * TR + 16 * g + r;
            row = row < a.B ? row : a.B - 1;
            lg[g] = tg[g] = a.gender[row];
            la[g] = ta[g] = a.age[row];
        }
    }
    // input rows: the tile's RG x KB0 (row group, k step) pieces are dealt over the four wavefronts.  The decoder chain of
    // a training pass builds its input itself: z = mu + exp(log sigma^2 / 2) eps (model.py:56-57) with eps given or drawn
    // (Philox block = flat element index / 4, as reparam_kl_kernel), and leaves eps, z (fp32 and bf16) and the tile's KL sum
    constexpr int NIN = (RG * KB0 + 3) / 4;
    const bool rep = RECON && a.mu != nullptr;  // (decoder instantiations only: the encoders carry none of this)
    f32x4 vin[NIN][2], vlv[NIN][2], vep[NIN][2];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int pc = 4 * i + wave, g = pc % RG, kb = pc / RG;
        if (pc < RG * KB0) {
            long row = tile * TR + 16 * g + r;
            row = row < a.B ? row : a.B - 1;
            const size_t off = (size_t)row * a.K0 + 32 * kb + 4 * q;
            const float* ip = (rep ? a.mu : a.in) + off;
            vin[i][0] = *reinterpret_cast<const f32x4*>(ip);
            vin[i][1] = *reinterpret_cast<const f32x4*>(ip + 16);
            if (rep) {
                vlv[i][0] = *reinterpret_cast<const f32x4*>(a.lv + off);
                vlv[i][1] = *reinterpret_cast<const f32x4*>(a.lv + off + 16);
                if (a.eps_in) {
                    vep[i][0] = *reinterpret_cast<const f32x4*>(a.eps_in + off);
                    vep[i][1] = *reinterpret_cast<const f32x4*>(a.eps_in + off + 16);
                }
            }
        }
    }
    const f32x4 vb01 = *reinterpret_cast<const f32x4*>((tid < 128 ? a.bias[0] : a.bias[1] - H) + 4 * tid);
    f32x4 vb2 = {0.f, 0.f, 0.f, 0.f};
    if (tid < 16 * MB2) vb2 = *reinterpret_cast<const f32x4*>(a.bias[2] + 4 * tid);
    f32x4 vt[3];
    if (has_tab) {
#pragma unroll
        for (int i = 0; i < 3; ++i) vt[i] = *reinterpret_cast<const f32x4*>(a.ctab + 4 * (tid + 256 * i));
    }
    PSTAMP(2);
    *reinterpret_cast<f32x4*>(sbias + 4 * tid) = vb01;   // [bias0 | bias1]
    if (tid < 16 * MB2) *reinterpret_cast<f32x4*>(sbias + 2 * H + 4 * tid) = vb2;
    if (has_tab) {
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4*>(stab + 4 * (tid + 256 * i)) = vt[i];
        if (tid < 128) *reinterpret_cast<f32x4*>(stab + 6 * H + 4 * tid) = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            tg[g] = ((unsigned)tg[g] < 3u ? tg[g] : 6) * H;
            ta[g] = ((unsigned)ta[g] < 3u ? 3 + ta[g] : 6) * H;
        }
    }
    PSTAMP(3);
    float kl = 0.f;
    if (rep) {
        uint64_t seed = a.seed, offset = a.offset;
        if (a.rng && !a.eps_in) { seed = a.rng[0]; offset = a.rng[1]; }
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int pc = 4 * i + wave, g = pc % RG, kb = pc / RG;
            const long row = tile * TR + 16 * g + r;
            if (pc < RG * KB0 && row < a.B) {
                const size_t off = (size_t)row * a.K0 + 32 * kb + 4 * q;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 ep = a.eps_in ? vep[i][h] : philox_normal4(offset + (uint64_t)((off + 16 * h) >> 2), seed);
                    const f32x4 m = vin[i][h], l = vlv[i][h];
                    f32x4 zz;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        zz[e] = m[e] + expf(0.5f * l[e]) * ep[e];
                        kl += 1.0f + l[e] - m[e] * m[e] - expf(l[e]);
                    }
                    vin[i][h] = zz;
                    *reinterpret_cast<f32x4*>(a.eps_out + off + 16 * h) = ep;
                    *reinterpret_cast<f32x4*>(a.z_out + off + 16 * h) = zz;
                }
                const bf16x8 zp = pack8<false>(vin[i][0], vin[i][1]);
                *reinterpret_cast<bf16x4*>(a.zb + off) = bf16x4{zp[0], zp[1], zp[2], zp[3]};
                *reinterpret_cast<bf16x4*>(a.zb + off + 16) = bf16x4{zp[4], zp[5], zp[6], zp[7]};
            }
        }
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) kl += __shfl_xor(kl, o2);
        if (lane == 0) red[wave] = kl;
    }
    PSTAMP(4);
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int pc = 4 * i + wave, g = pc % RG, kb = pc / RG;
        if (pc < RG * KB0) {
            const bf16x8 pk = pack8<false>(vin[i][0], vin[i][1]);
            *reinterpret_cast<bf16x8*>(act + (kb * RG + g) * 1024 + lane * 16) = pk;
            if constexpr (!RECON) {
                // the row-major bf16 copy of the input rows the weight gradients read (was a rider of the pack launch)
                const long row = tile * TR + 16 * g + r;
                if (a.xb_out && row < a.B) {
                    u16* xp = a.xb_out + (size_t)row * a.K0 + 32 * kb + 4 * q;
                    *reinterpret_cast<bf16x4*>(xp) = bf16x4{pk[0], pk[1], pk[2], pk[3]};
                    *reinterpret_cast<bf16x4*>(xp + 16) = bf16x4{pk[4], pk[5], pk[6], pk[7]};
                }
            }
        }
    }
    if constexpr (!RECON) {
        if (a.oh_out && has_tab && wave == 0 && q == 0) {  // one-hot label rows [B][8] (the concatenated gender / age labels of config 3: SURVEY.md 8a row A9)
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                const long row = tile * TR + 16 * g + r;
                if (row < a.B) {
                    bf16x8 oh = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        oh[c] = (lg[g] == c) ? (short)0x3F80 : (short)0;
                        oh[3 + c] = (la[g] == c) ? (short)0x3F80 : (short)0;
                    }
                    *reinterpret_cast<bf16x8*>(a.oh_out + (size_t)row * 8) = oh;
                }
            }
        }
    }
    PSTAMP(5);
    __syncthreads();
    PSTAMP(6);
    if (rep && tid == 0) a.kl_partials[tile] = (red[0] + red[1]) + (red[2] + red[3]);

    f32x4 acc[8][RG];
    int s = 0;
    // one layer: KB k steps over MB blocks x RG row groups; the first step's C operand is the bias (+ table rows: layer 0
    // of a conditional chain starts from acc = bias + table[gender] + table[3 + age], summed in fp32)
    auto layer = [&](auto KBc, auto MBc, const float* bias, bool tab) {
        constexpr int KB = decltype(KBc)::value, MB = decltype(MBc)::value;
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 16 * (MB * wave + mi) + 4 * q);
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                acc[mi][g] = bv;
                if (tab) {
                    acc[mi][g] += *reinterpret_cast<const f32x4*>(stab + tg[g] + 16 * (MB * wave + mi) + 4 * q);
                    acc[mi][g] += *reinterpret_cast<const f32x4*>(stab + ta[g] + 16 * (MB * wave + mi) + 4 * q);
                }
            }
        }
        bfetch(0, 0);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb, ++s) {
            __builtin_amdgcn_sched_barrier(0);
            if (s + NA - 1 < S) wfetch(s + NA - 1);
            if (kb + 1 < KB) bfetch((kb + 1) & 1, kb + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                for (int g = 0; g < RG; ++g)
                    acc[mi][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[s % NA][mi], bb[kb & 1][g], acc[mi][g], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // a hidden activation: (+ table rows,) ReLU, bf16 -> the next layer's image AND the staged row-major copy, which then
    // leaves as whole rows (1 KB per store instruction) into save[]
    auto publish = [&](u16* save) {
        __syncthreads();  // every wavefront has read the previous image; the previous staged rows have left
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 t[RG];
#pragma unroll
            for (int g = 0; g < RG; ++g) t[g] = pack8<true>(acc[2 * kk][g], acc[2 * kk + 1][g]);
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                *reinterpret_cast<bf16x8*>(act + ((4 * wave + kk) * RG + g) * 1024 + lane * 16) = t[g];
                // row-major: features 32 kb + 4q .. +3 and 32 kb + 16 + 4q .. +3 of row 16 g + r  (kb = 4 wave + kk)
                char* sp = stg + (16 * g + r) * SROW + (32 * (4 * wave + kk) + 4 * q) * 2;
                const bf16x4 lo4 = {t[g][0], t[g][1], t[g][2], t[g][3]}, hi4 = {t[g][4], t[g][5], t[g][6], t[g][7]};
                *reinterpret_cast<bf16x4*>(sp) = lo4;
                *reinterpret_cast<bf16x4*>(sp + 32) = hi4;
            }
        }
        __syncthreads();
        // copy-out: wavefront w sends rows 4 RG w .. 4 RG (w + 1) - 1, one 1 KB row per instruction
        constexpr int RW = 4 * RG;
        constexpr int RB = RW < 8 ? RW : 8;  // rows per burst (16-row tiles: four rows per wavefront)
#pragma unroll
        for (int i0 = 0; i0 < RW; i0 += RB) {
            bf16x8 v[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) v[i] = *reinterpret_cast<const bf16x8*>(stg + (RW * wave + i0 + i) * SROW + lane * 16);
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const long row = tile * TR + RW * wave + i0 + i;
                if (row < a.B) *reinterpret_cast<bf16x8*>(save + row * H + lane * 8) = v[i];
            }
        }
    };
    CSTAMP(1);
    layer(std::integral_constant<int, KB0>{}, std::integral_constant<int, 8>{}, sbias, has_tab);
    LSTAMP(2);
    publish(a.save[0]);
    LSTAMP(3);
    layer(std::integral_constant<int, HK>{}, std::integral_constant<int, 8>{}, sbias + H, false);
    LSTAMP(4);
    publish(a.save[1]);
    LSTAMP(5);
    layer(std::integral_constant<int, HK>{}, std::integral_constant<int, MB2>{}, sbias + 2 * H, false);
    LSTAMP(6);

    // ---- output: fp32 rows through the staging buffer (the bf16 rows of save[1] have been read: barrier first) -------
    float sq = 0.f;
    const float gsc = RECON ? a.scale * (a.go ? *a.go : 1.0f) : 0.f;
    char* ostg = WIDE1 ? act : stg;
#pragma unroll
    for (int ps = 0; ps < TR / RPP; ++ps) {
        __syncthreads();
#pragma unroll
        for (int g = ps * (RPP / 16); g < (ps + 1) * (RPP / 16); ++g)
#pragma unroll
            for (int mi = 0; mi < MB2; ++mi)
                *reinterpret_cast<f32x4*>(ostg + (16 * (g - ps * (RPP / 16)) + r) * OROW + (16 * (MB2 * wave + mi) + 4 * q) * 4) = acc[mi][g];
        __syncthreads();
        // pieces of RPP rows, row-major: thread t takes pieces t, t + 256, ...
#pragma unroll
        for (int i0 = 0; i0 < NPC; i0 += 4) {
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pc = 256 * (i0 + i) + tid;
                if (i0 + i < NPC && pc < RPP * PPR) v[i] = *reinterpret_cast<const f32x4*>(ostg + (pc / PPR) * OROW + (pc % PPR) * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pc = 256 * (i0 + i) + tid;
                if (i0 + i >= NPC || pc >= RPP * PPR) continue;
                const long row = tile * TR + ps * RPP + pc / PPR;
                if (row >= a.B) continue;
                const size_t off = (size_t)row * NOUT + (pc % PPR) * 4;
                *reinterpret_cast<f32x4*>(a.out + off) = v[i];
                if (RECON && !a.raw_out) {
                    const f32x4 xt = tv[ps * NPC + i0 + i];
                    bf16x4 gq;
                    float df[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        df[e] = v[i][e] - xt[e];
                        sq += df[e] * df[e];
                    }
                    const nbf16x2 g0 = __builtin_convertvector(f32x2{gsc * df[0], gsc * df[1]}, nbf16x2);
                    const nbf16x2 g1 = __builtin_convertvector(f32x2{gsc * df[2], gsc * df[3]}, nbf16x2);
                    const s16x2 s0 = *reinterpret_cast<const s16x2*>(&g0), s1 = *reinterpret_cast<const s16x2*>(&g1);
                    gq[0] = s0[0]; gq[1] = s0[1]; gq[2] = s1[0]; gq[3] = s1[1];
                    *reinterpret_cast<bf16x4*>(a.gy + off) = gq;
                }
            }
        }
    }
    if (RECON && !a.raw_out) {
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) sq += __shfl_xor(sq, o2);
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (tid == 0) a.partials[tile] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    CSTAMP(7);
#ifdef PSVAE_STAMP
    if (tid == 0 && blockIdx.x == 5 && blockIdx.y == 0)
        for (int i = 0; i < 8; ++i) g_cstamp[RECON][i] = st_[i];
#endif
}

// ---- backward chains: the input-gradient (dgrad) path of one 3-layer MLP, same organisation ---------------------------
//   layer 0: in [B, K0] (decoder: the bf16 MSE gradient g_y; encoders: fp32 g_mu / g_logvar, rounded on load)
//            x W_last -> 512 hidden gradients, masked by the saved activation of that layer (ReLU backward) -> save[0], image
//   layer 1: 512 -> 512 through the middle layer's weights, masked by the first hidden activation -> save[1], image
//   layer 2 (decoder only, MB2 > 0): 512 -> Z through decoder.0's weights: g_z, fp32
// dX[b][i] = sum_o dY[b][o] W[o][i]: in the transposed scheme the A operand is W^T, so the weights arrive packed
// "transposed" (chain_pack_kernel, T = 1).  The saved activations are read as 8-byte pieces straight into each lane's C
// layout (16 rows x 32 B per instruction -- 32 KB per activation and tile, requested in the prologue with everything else).
template <int KB0, int MB2, bool IN_BF16, int RG>
__global__ __launch_bounds__(256, 1) void chain_bwd_bf16_kernel(const ChainBwdBatch batch) {
#ifdef PSVAE_STAMP
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    CSTAMP(0);
    constexpr int TR = 16 * RG;
    constexpr int IMG = 16 * RG * 1024;
    constexpr int STG = TR * SROW;
    constexpr int HK = H / 32;
    constexpr int S = KB0 + HK + (MB2 > 0 ? HK : 0);
    constexpr int NA = RG <= 2 ? PSVAE_CHAIN_NA : 3;  // weight-fragment ring: this k step and NA - 1 ahead (registers allow more at 32 rows)
    const ChainBwdArgs& a = batch.c[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* act = reinterpret_cast<char*>(lds);
    char* stg = act + IMG;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long tile = blockIdx.x;
    auto mb_of = [&](int s) -> int { return s < KB0 + HK ? 8 : MB2; };
    typedef const __attribute__((address_space(1))) u16 gu16;
    typedef const __attribute__((address_space(1))) bf16x8 gbf16x8;
    const u16* wl0 = a.wp[0] + (size_t)wave * (KB0 * 8 * 512) + lane * 8;
    const u16* wl1 = a.wp[1] + (size_t)wave * (HK * 8 * 512) + lane * 8;
    const u16* wl2 = MB2 > 0 ? a.wp[2] + (size_t)wave * (HK * MB2 * 512) + lane * 8 : nullptr;
    bf16x8 wa[NA][8];
    gu16* wsrc = nullptr;
    auto wfetch = [&](int s) {
        if (s == 0) wsrc = (gu16*)wl0;
        if (s == KB0) wsrc = (gu16*)wl1;
        if (s == KB0 + HK) wsrc = (gu16*)wl2;
        asm volatile("" : "+v"(wsrc));
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
            if (mi < mb_of(s)) wa[s % NA][mi] = *(gbf16x8*)(wsrc + mi * 512);
        wsrc += mb_of(s) * 512;
    };
    bf16x8 bb[2][RG];
    auto bfetch = [&](int set, int kb) {
        const char* src = act + kb * (RG * 1024) + lane * 16;
#pragma unroll
        for (int g = 0; g < RG; ++g) bb[set][g] = *reinterpret_cast<const bf16x8*>(src + 1024 * g);
    };
#pragma unroll
    for (int s0 = 0; s0 < NA - 1; ++s0) wfetch(s0);
    // ---- prologue: input rows and both layers' masks go out together ---------------------------------------------------
    constexpr int NIN = (RG * KB0 + 3) / 4;
    f32x4 vin[IN_BF16 ? 1 : NIN][2];
    bf16x4 vib[IN_BF16 ? NIN : 1][2];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int pc = 4 * i + wave, g = pc % RG, kb = pc / RG;
        if (pc < RG * KB0) {
            long row = tile * TR + 16 * g + r;
            row = row < a.B ? row : a.B - 1;
            if (IN_BF16) {
                const u16* ip = reinterpret_cast<const u16*>(a.in) + row * a.K0 + 32 * kb + 4 * q;
                vib[i][0] = *reinterpret_cast<const bf16x4*>(ip);
                vib[i][1] = *reinterpret_cast<const bf16x4*>(ip + 16);
            } else {
                const float* ip = reinterpret_cast<const float*>(a.in) + row * a.K0 + 32 * kb + 4 * q;
                vin[i][0] = *reinterpret_cast<const f32x4*>(ip);
                vin[i][1] = *reinterpret_cast<const f32x4*>(ip + 16);
            }
        }
    }
    // masks: this lane's features of its rows, as the accumulators hold them: [layer][kk][g][half] = 4 bf16
    bf16x4 mk[2][4][RG][2];
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                long row = tile * TR + 16 * g + r;
                row = row < a.B ? row : a.B - 1;
                const u16* mp = a.mask[l] + row * H + 32 * (4 * wave + kk) + 4 * q;
                mk[l][kk][g][0] = *reinterpret_cast<const bf16x4*>(mp);
                mk[l][kk][g][1] = *reinterpret_cast<const bf16x4*>(mp + 16);
            }
    // training pass: the rows of mu, log sigma^2 and eps the output pass will need, piece by piece as it walks them
    constexpr int NPCR = MB2 > 0 ? (TR * (64 * MB2 * 4 / 16) + 255) / 256 : 1;
    f32x4 rmu[NPCR], rlv[NPCR], rep_[NPCR];
    if (MB2 > 0 && a.mu != nullptr) {
#pragma unroll
        for (int i = 0; i < NPCR; ++i) {
            constexpr int PPR_ = MB2 > 0 ? 64 * MB2 * 4 / 16 : 1;
            const int pc = 256 * i + tid;
            long row = tile * TR + pc / PPR_;
            row = row < a.B ? row : a.B - 1;
            const size_t off = (size_t)row * (64 * MB2) + (pc % PPR_) * 4;
            rmu[i] = *reinterpret_cast<const f32x4*>(a.mu + off);
            rlv[i] = *reinterpret_cast<const f32x4*>(a.lv + off);
            rep_[i] = *reinterpret_cast<const f32x4*>(a.eps + off);
        }
    }
    // L2 warm-up.  The workgroups of an XCD stream the SAME packed weights in lock step, so a line is an L2 miss in flight
    // for all of them at once and the early layers run at the miss latency.  Each workgroup therefore touches its share of
    // EVERY layer's lines once, up front (one dword per 128-B line; lines k, k + P, k + 2P ... for the k-th of the chain's P
    // workgroups on this XCD): after one round trip the whole stream is in the XCD's L2.  Measured at cfg3: decoder gradient
    // chain 17.1 -> 15.5 us, encoder gradient chains unchanged; the forward chains, whose prologues already pull 100 KB of
    // inputs, do not gain (17.9 -> 18.1, 17.8 -> 18.1 us) and do not have it.
    unsigned warm[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
#ifndef PSVAE_CHAIN_NO_WARM
    {
        const int P = ((int)gridDim.x + 7) >> 3, k = (int)blockIdx.x >> 3;
        if (P >= 8) {
            const int nlines[3] = {KB0 * 256, HK * 256, HK * MB2 * 32};
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int line = k + P * (tid + 256 * rr);
                    if (line < nlines[l]) warm[l][rr] = *reinterpret_cast<const __attribute__((address_space(1))) unsigned*>(
                                              reinterpret_cast<uintptr_t>(a.wp[l]) + (size_t)line * 128);
                }
        }
    }
#endif
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int pc = 4 * i + wave, g = pc % RG, kb = pc / RG;
        if (pc < RG * KB0) {
            bf16x8 v;
            if (IN_BF16) v = bf16x8{vib[i][0][0], vib[i][0][1], vib[i][0][2], vib[i][0][3], vib[i][1][0], vib[i][1][1], vib[i][1][2], vib[i][1][3]};
            else v = pack8<false>(vin[i][0], vin[i][1]);
            *reinterpret_cast<bf16x8*>(act + (kb * RG + g) * 1024 + lane * 16) = v;
        }
    }
    __syncthreads();
    {   // (the warm-up loads' values are never used; this keeps them from being dropped: B is never negative)
        const unsigned w = (warm[0][0] ^ warm[0][1]) ^ (warm[1][0] ^ warm[1][1]) ^ (warm[2][0] ^ warm[2][1]);
        if (a.B < 0) a.save[0][tid] = (u16)w;
    }

    f32x4 acc[8][RG];
    int s = 0;
    auto layer = [&](auto KBc, auto MBc) {
        constexpr int KB = decltype(KBc)::value, MB = decltype(MBc)::value;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        bfetch(0, 0);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb, ++s) {
            __builtin_amdgcn_sched_barrier(0);
            if (s + NA - 1 < S) wfetch(s + NA - 1);
            if (kb + 1 < KB) bfetch((kb + 1) & 1, kb + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                for (int g = 0; g < RG; ++g)
                    acc[mi][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[s % NA][mi], bb[kb & 1][g], kb == 0 ? zero : acc[mi][g], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // a hidden gradient: zero where the saved activation is zero (its bf16 bits: ReLU outputs are +0 or positive), bf16
    // -> image + staged rows -> save[]
    auto publish = [&](u16* save, int l) {
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 t[RG];
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                union { bf16x8 v; unsigned u[4]; } p;
                p.v = pack8<false>(acc[2 * kk][g], acc[2 * kk + 1][g]);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    union { bf16x4 v; unsigned u[2]; } m;
                    m.v = mk[l][kk][g][h];
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                        union { unsigned u; u16x2 v; s16x2 sv; } w;
                        w.u = m.u[d];
                        w.v = __builtin_elementwise_min(w.v, u16x2{1, 1});  // 1 where the activation was positive
                        w.sv = s16x2{0, 0} - w.sv;                           // 0xFFFF there, 0 elsewhere
                        p.u[2 * h + d] &= w.u;
                    }
                }
                t[g] = p.v;
            }
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                *reinterpret_cast<bf16x8*>(act + ((4 * wave + kk) * RG + g) * 1024 + lane * 16) = t[g];
                char* sp = stg + (16 * g + r) * SROW + (32 * (4 * wave + kk) + 4 * q) * 2;
                const bf16x4 lo4 = {t[g][0], t[g][1], t[g][2], t[g][3]}, hi4 = {t[g][4], t[g][5], t[g][6], t[g][7]};
                *reinterpret_cast<bf16x4*>(sp) = lo4;
                *reinterpret_cast<bf16x4*>(sp + 32) = hi4;
            }
        }
        __syncthreads();
        constexpr int RW = 4 * RG;
        constexpr int RB = RW < 8 ? RW : 8;  // rows per burst (16-row tiles: four rows per wavefront)
#pragma unroll
        for (int i0 = 0; i0 < RW; i0 += RB) {
            bf16x8 v[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) v[i] = *reinterpret_cast<const bf16x8*>(stg + (RW * wave + i0 + i) * SROW + lane * 16);
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const long row = tile * TR + RW * wave + i0 + i;
                if (row < a.B) *reinterpret_cast<bf16x8*>(save + row * H + lane * 8) = v[i];
            }
        }
    };
    CSTAMP(1);
    layer(std::integral_constant<int, KB0>{}, std::integral_constant<int, 8>{});
    CSTAMP(2);
    publish(a.save[0], 0);
    CSTAMP(3);
    layer(std::integral_constant<int, HK>{}, std::integral_constant<int, 8>{});
    CSTAMP(4);
    publish(a.save[1], 1);
    CSTAMP(5);
    if constexpr (MB2 > 0) {
        layer(std::integral_constant<int, HK>{}, std::integral_constant<int, MB2>{});
        CSTAMP(6);
        constexpr int NOUT = 64 * MB2;
        constexpr int OROW = NOUT * 4 + 16;
        static_assert(OROW * TR <= STG, "staging buffer too small for the output rows");
        constexpr int PPR = NOUT * 4 / 16;
        constexpr int NPC = (TR * PPR + 255) / 256;
        __syncthreads();
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int mi = 0; mi < MB2; ++mi)
                *reinterpret_cast<f32x4*>(stg + (16 * g + r) * OROW + (16 * (MB2 * wave + mi) + 4 * q) * 4) = acc[mi][g];
        __syncthreads();
        // g_z leaves as whole rows; a training pass turns it into g_mu / g_logvar on the way out (the reparameterisation's
        // backward plus the KL term's gradient, lightning.py:115-120 through autograd; reparam_bwd_final_kernel's formulas)
        const bool rb = a.mu != nullptr;
        const float kbeta = rb ? (a.go ? *a.go : 1.0f) * a.beta / (float)a.B : 0.f;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int pc = 256 * i + tid;
            if (pc < TR * PPR) {
                const long row = tile * TR + pc / PPR;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + (pc / PPR) * OROW + (pc % PPR) * 16);
                if (row < a.B) {
                    const size_t off = (size_t)row * NOUT + (pc % PPR) * 4;
                    if (!rb) {
                        *reinterpret_cast<f32x4*>(a.out + off) = v;
                    } else {
                        const f32x4 m = rmu[i], l = rlv[i], ep = rep_[i];
                        f32x4 gm, gl;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            gm[e] = v[e] + kbeta * m[e];
                            gl[e] = v[e] * ep[e] * 0.5f * expf(0.5f * l[e]) + 0.5f * kbeta * (expf(l[e]) - 1.0f);
                        }
                        *reinterpret_cast<f32x4*>(a.g_mu + off) = gm;
                        *reinterpret_cast<f32x4*>(a.g_lv + off) = gl;
                        const bf16x8 pk = pack8<false>(gm, gl);
                        *reinterpret_cast<bf16x4*>(a.gmu_b + off) = bf16x4{pk[0], pk[1], pk[2], pk[3]};
                        *reinterpret_cast<bf16x4*>(a.glv_b + off) = bf16x4{pk[4], pk[5], pk[6], pk[7]};
                    }
                }
            }
        }
        if (rb && tile == 0) {  // the loss from the forward's partial sums, and the Philox state's advance (every reader ran
            float* sh4 = reinterpret_cast<float*>(stg);  // launches ago); `stg` is free once every thread's reads above are done
            __syncthreads();
            float k = 0.f, rr = 0.f;
            for (int i = tid; i < a.n_kl; i += 256) k += a.kl_partials[i];
            for (int i = tid; i < a.n_recon; i += 256) rr += a.recon_partials[i];
            const float ks = block_sum_256(k, sh4);
            __syncthreads();
            const float rs = block_sum_256(rr, sh4);
            if (tid == 0) {
                const float recon = a.recon_mode ? rs / (float)a.B : rs / ((float)a.B * (float)a.K0) / 10.0f;
                const float klv = -0.5f * ks / (float)a.B;
                a.loss3[0] = recon + a.beta * klv;
                a.loss3[1] = recon;
                a.loss3[2] = klv;
                if (a.rng) a.rng[1] += a.nblk;
            }
        }
    }
    CSTAMP(7);
#ifdef PSVAE_STAMP
    if (MB2 == 0) st_[6] = st_[5];  // (no third layer)
    if (tid == 0 && blockIdx.x == 5 && blockIdx.y == 0)
        for (int i = 0; i < 8; ++i) g_cstamp[MB2 > 0 ? 0 : 1][i] = st_[i];
#endif
}

template <int KB0, int MB2, bool IN_BF16, int RG>
static hipError_t launch_chain_bwd_rg(const ChainBwdBatch& b, hipStream_t s) {
    constexpr int TR = 16 * RG;
    constexpr size_t lds = (size_t)16 * RG * 1024 + (size_t)TR * SROW;
    static std::atomic<unsigned long long> lds_ok{0};  // per device: psvae_common.h lds_opt_in
    if (hipError_t attr_err = lds_opt_in(lds_ok, reinterpret_cast<const void*>(&chain_bwd_bf16_kernel<KB0, MB2, IN_BF16, RG>), 160 * 1024))
        return attr_err;
    const int ntiles = (b.c[0].B + TR - 1) / TR;
    hipLaunchKernelGGL((chain_bwd_bf16_kernel<KB0, MB2, IN_BF16, RG>), dim3(ntiles, b.n), dim3(256), lds, s, b);
#ifdef PSVAE_STAMP
    {
        unsigned long long h[2][8];
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cstamp), sizeof(h)) == hipSuccess) {
            const int w = MB2 > 0 ? 0 : 1;
            printf("chain gradient %s stamps (cycles): prologue | L0 | publish | L1 | publish | L2 | output:", w ? "encoder" : "decoder");
            for (int i = 1; i < 8; ++i) printf(" %llu", h[w][i] > h[w][i - 1] ? h[w][i] - h[w][i - 1] : 0ull);
            printf("\n");
        }
    }
#endif
    return hipGetLastError();
}

// decoder chain: in = g_y bf16 [B, D] (K0 = D), third layer -> g_z [B, Z]; encoder chains: in = fp32 [B, Z], two layers
hipError_t launch_chain_bwd_bf16(const ChainBwdBatch& b, hipStream_t s) {
    const ChainBwdArgs& c = b.c[0];
    const int rows = chain_rows_per_tile(c.B, b.n);
#define PSVAE_BD(KB0, MB2) if (c.in_bf16 && c.K0 == 32 * KB0 && c.NOUT == 64 * MB2) \
        return rows == 16 ? launch_chain_bwd_rg<KB0, MB2, true, 1>(b, s) : rows == 32 ? launch_chain_bwd_rg<KB0, MB2, true, 2>(b, s) : launch_chain_bwd_rg<KB0, MB2, true, 4>(b, s)
#define PSVAE_BE(KB0) if (!c.in_bf16 && c.K0 == 32 * KB0 && c.NOUT == 0) \
        return rows == 16 ? launch_chain_bwd_rg<KB0, 0, false, 1>(b, s) : rows == 32 ? launch_chain_bwd_rg<KB0, 0, false, 2>(b, s) : launch_chain_bwd_rg<KB0, 0, false, 4>(b, s)
    PSVAE_BD(8, 1); PSVAE_BE(2);  // cfg3
    PSVAE_BD(4, 1); PSVAE_BD(16, 1); PSVAE_BD(4, 2); PSVAE_BD(8, 2); PSVAE_BD(16, 2); PSVAE_BE(4);
#undef PSVAE_BD
#undef PSVAE_BE
    return hipErrorInvalidValue;
}

bool chain_fwd_ok(int B, int K0, int NOUT) {
    if (getenv("PSVAE_BF16_NO_CHAIN")) return false;
    // (measured faster than the launch-per-layer path at every batch from 128 to 16,384: 90 / 102 / 123 / 159 / 298 us per
    //  conditional step at B = 128 / 1024 / 4096 / 8192 / 16384 against 105 / 130 / 193 / 282 / 462)
    const int minb = getenv("PSVAE_BF16_CHAIN_MIN") ? atoi(getenv("PSVAE_BF16_CHAIN_MIN")) : 64;
    // D in {128, 256, 512}, Z in {64, 128}: encoders K0 = D, NOUT = Z; decoder K0 = Z, NOUT = D
    const bool enc = (K0 == 128 || K0 == 256 || K0 == 512) && (NOUT == 64 || NOUT == 128);
    const bool dec = (K0 == 64 || K0 == 128) && (NOUT == 128 || NOUT == 256 || NOUT == 512);
    return B >= minb && (enc || dec);
}

template <int KB0, int MB2, bool RECON, int RG>
static hipError_t launch_chain_rg(const ChainBatch& b, hipStream_t s) {
    constexpr int TR = 16 * RG;
    constexpr size_t lds = (size_t)16 * RG * 1024 + (size_t)TR * SROW + (size_t)(2 * H + 64 * MB2 + 7 * H + 4) * sizeof(float);
    const int ntiles = (b.c[0].B + TR - 1) / TR;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static std::atomic<unsigned long long> lds_ok{0};  // per device: psvae_common.h lds_opt_in
    if (hipError_t attr_err = lds_opt_in(lds_ok, reinterpret_cast<const void*>(&chain_fwd_bf16_kernel<KB0, MB2, RECON, RG>), 160 * 1024))
        return attr_err;
    hipLaunchKernelGGL((chain_fwd_bf16_kernel<KB0, MB2, RECON, RG>), dim3(ntiles, b.n), dim3(256), lds, s, b);
#ifdef PSVAE_STAMP
    {
        unsigned long long h[2][8];
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cstamp), sizeof(h)) == hipSuccess) {
            printf("chain %s stamps (cycles): prologue | L0 | publish | L1 | publish | L2 | output:", RECON ? "decoder" : "encoder");
            for (int i = 1; i < 8; ++i) printf(" %llu", h[RECON][i] - h[RECON][i - 1]);
            printf("\n");
        }
    }
#endif
    return hipGetLastError();
}

// rows per tile: 64, or 32 / 16 where that fills the chip better (PSVAE_CHAIN_RG = 1 | 2 | 4 overrides).  A chain
// workgroup streams its MLP's weights whatever its rows, so fewer rows per tile only pay while the tiles still make ONE round
// of workgroups: the decoder chains of cfg3 (one chain, 4,096 rows) run 256 tiles of 16 rows -- their row-proportional parts
// (input stage, publishes, output pass) halve: 19.3 -> 18.3 and 20.0 -> 17.2 us --, the two encoders 2 x 128 tiles of 32
// (2 x 256 of 16 would be two rounds: 17.2 -> 22.8 us)
int chain_rows_per_tile(int B, int nchains) {
    if (const char* e = getenv("PSVAE_CHAIN_RG")) return atoi(e) == 1 ? 16 : atoi(e) == 2 ? 32 : 64;
    if (((B + 63) / 64) * nchains >= 256) return 64;
    return ((B + 15) / 16) * nchains <= 256 ? 16 : 32;
}

template <int KB0, int MB2, bool RECON>
static hipError_t launch_chain_t(const ChainBatch& b, hipStream_t s) {
    const int rows = chain_rows_per_tile(b.c[0].B, b.n);
    return rows == 16 ? launch_chain_rg<KB0, MB2, RECON, 1>(b, s) : rows == 32 ? launch_chain_rg<KB0, MB2, RECON, 2>(b, s) : launch_chain_rg<KB0, MB2, RECON, 4>(b, s);
}

// every chain of a batch has the same B, K0 and NOUT (the two encoders; or the decoder alone)
hipError_t launch_chain_fwd_bf16(const ChainBatch& b, hipStream_t s) {
    const ChainArgs& c = b.c[0];
    const bool recon = c.gy != nullptr || c.mu != nullptr;  // the decoder of a training pass
#define PSVAE_ENC(KB0, MB2) if (!recon && c.K0 == 32 * KB0 && c.NOUT == 64 * MB2) return launch_chain_t<KB0, MB2, false>(b, s)
#define PSVAE_DEC(KB0, MB2) if (recon && c.K0 == 32 * KB0 && c.NOUT == 64 * MB2) return launch_chain_t<KB0, MB2, true>(b, s)
    PSVAE_ENC(8, 1); PSVAE_DEC(2, 4);  // cfg3: encoders 256 -> 64, decoder 64 -> 256
    PSVAE_ENC(4, 1); PSVAE_ENC(16, 1); PSVAE_ENC(4, 2); PSVAE_ENC(8, 2); PSVAE_ENC(16, 2);
    PSVAE_DEC(2, 2); PSVAE_DEC(2, 8); PSVAE_DEC(4, 2); PSVAE_DEC(4, 4); PSVAE_DEC(4, 8);
#undef PSVAE_ENC
#undef PSVAE_DEC
    return hipErrorInvalidValue;
}

}  // namespace psvae
