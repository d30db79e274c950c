// Fused decode (VAEModel.decode, /root/reference/ps_vae/model.py:65-69; the sampling path of inference.py:25,105 --
// BASELINE config 5): all three decoder layers in ONE kernel, hidden activations never leave the register file.
//
// The per-layer path (psvae_api.hip: three grouped-GEMM launches per 65,536-row chunk) writes d1 and d2 -- 2 KB per row
// each -- to memory and reads them back: 8 GB per 1 M rows, and each launch runs at the tile kernel's 66 % of the fp32
// MFMA peak.  Here one wavefront owns 16 rows for all three layers, everything TRANSPOSED (as in langevin.hip):
//     d^T [out x 16 rows] = W [out x in] . a^T [in x 16 rows]         A = weight fragments, B = activations, C init = bias
// v_mfma_f32_16x16x4_f32: lane (r, q) supplies A[i = r][k = q], B[k = q][n = r] and receives D[i = 4q + j][n = r]: a
// layer's output (lane (r, q): features 16 m + 4q + j of row r) is, register for register, the next layer's B operand,
// with k slot q of MFMA step (mb, j) = feature 16 mb + 4q + j.  The weights are PRE-PACKED once per call in exactly
// that consumption order -- Wp[layer][mb][j][m/4][q][r][m%4] = W[16 m + r][16 mb + 4q + j] -- so the four wavefronts of a
// workgroup stream them global -> registers -> LDS as plain 16-byte copies (32 KB chunks, three LDS buffers, one
// barrier per chunk of 128 MFMAs per wavefront) and read their fragments as contiguous ds_read_b128.
// Per 16 rows: 6,656 MFMAs (Z = 64, D = 256) against 96 KB of LDS fragment reads per wavefront -- MFMA-bound.
#include <mutex>
#include <type_traits>

#include "psvae_common.h"

namespace psvae {

// ---- weight packing: Wp[mb][j][m4][q][r][e] = W[16 (4 m4 + e) + r][16 mb + 4 q + j],  W is [OUT, IN] row-major --------
// (lane (r, q) reads output blocks 4 m4 .. 4 m4 + 3 of step (mb, j) as ONE 16-byte piece, and the 64 lanes' pieces are
//  contiguous: a ds_read_b128 of the wavefront covers 1 KB without a bank conflict.  With the m index innermost per
//  lane -- a 128-byte lane stride -- the 16 r lanes would all fall on the same four banks.)
__global__ __launch_bounds__(256) void pack_decode_weights_kernel(const float* __restrict__ W, float* __restrict__ Wp,
                                                                  int OUT, int IN) {
    const int M4 = OUT / 64;
    const long n = (long)OUT * IN;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int e = (int)(i & 3), r = (int)((i >> 2) & 15), q = (int)((i >> 6) & 3);
        long t = i >> 8;
        const int m4 = (int)(t % M4); t /= M4;
        const int j = (int)(t & 3);
        const int mb = (int)(t >> 2);
        Wp[i] = W[(size_t)(16 * (4 * m4 + e) + r) * IN + 16 * mb + 4 * q + j];
    }
}

constexpr int DF_MAXP = 8;  // pieces per thread of the largest chunk (MBO = 32: 32 KB)

struct DecodeFusedArgs {
    const float* z;       // [N, Z]
    float* x_hat;         // [N, D]
    const float* wp;      // packed weights: layer 0 | layer 1 | layer 2
    const float* bias0;   // [512]
    const float* bias1;   // [512]
    const float* bias2;   // [D]
    long N;
    int Z, D;
};

template <int ZB, int DB>  // Z = 16 ZB, D = 16 DB, hidden = 512
__global__ __launch_bounds__(256, 1) void decode_fused_kernel(const DecodeFusedArgs a) {
    constexpr int HB = H / 16;                      // 32 hidden blocks
    constexpr int CH_H = 4 * 4 * 16 * HB;           // floats per chunk of a layer with 512 outputs (32 KB)
    constexpr int CH_D = 4 * 4 * 16 * DB;           // ... of the output layer
    constexpr int NCH = ZB + HB + HB;               // chunks of the whole weight stream per tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wbuf = lds;                              // [3][CH_H]: chunk c lives in buffer c % 3
    float* sbias = lds + 3 * CH_H;                  // [512 + 512 + D]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    for (int i = tid; i < H; i += 256) { sbias[i] = a.bias0[i]; sbias[H + i] = a.bias1[i]; }
    for (int i = tid; i < 16 * DB; i += 256) sbias[2 * H + i] = a.bias2[i];

    // chunk c of the stream.  The loops below are fully unrolled (the activations are indexed statically), so every
    // chunk's address would be a loop invariant the compiler precomputes -- 68 address pairs, spilled.  One running
    // pointer, opaque to the optimiser, is advanced per fetch instead.
    auto ch_len = [&](int c) -> int { return c < ZB + HB ? CH_H : CH_D; };
    auto ch_mbo = [&](int c) -> int { return c < ZB + HB ? HB : DB; };
    f32x4 pf[DF_MAXP];  // the next chunk, in flight
    // (explicitly global: behind the asm the compiler would fall back to flat loads, which count on lgkmcnt too and
    //  force every LDS wait to lgkmcnt(0) -- the fragment lookahead would be lost)
    typedef const __attribute__((address_space(1))) float gfloat;
    typedef const __attribute__((address_space(1))) f32x4 gf32x4;
    gfloat* wsrc = nullptr;
    auto gfetch = [&](int c) {
        if (c == 0) wsrc = (gfloat*)(a.wp + 4 * tid);
        asm volatile("" : "+v"(wsrc));
        const int np = ch_len(c) / 4 / 256;
#pragma unroll
        for (int i = 0; i < DF_MAXP; ++i)
            if (i < np) pf[i] = *(const gf32x4*)(wsrc + 1024 * i);
        wsrc += ch_len(c);
    };
    // the stream is CYCLIC: chunk u (counted from the first tile on) lives in buffer u % 3 and NCH % 3 != 0, so a tile's
    // chunk c sits at base[c % 3] with the three bases rotated per tile -- the last chunks of a tile already stage and
    // read the first chunks of the next one, no restart at a tile boundary.
    int base[3] = {0, CH_H, 2 * CH_H};
    auto lstore = [&](int u) {  // u may run past NCH (next tile's chunks)
        float* dst = wbuf + base[u % 3];
        const int np = ch_len(u % NCH) / 4 / 256;
#pragma unroll
        for (int i = 0; i < DF_MAXP; ++i)
            if (i < np) *reinterpret_cast<f32x4*>(dst + 4 * (tid + 256 * i)) = pf[i];
    };
    // this lane's fragments of step j of chunk u: piece m4 at [j][m4][q][r][0..3]
    f32x4 w[2][HB / 4];  // two sets: the step being multiplied and the one after it
    auto frag = [&](int set, int u, int j) {
        const int mbo = ch_mbo(u % NCH);
        const float* wl = wbuf + base[u % 3] + j * (mbo / 4) * 256 + lane * 4;
#pragma unroll
        for (int m4 = 0; m4 < HB / 4; ++m4)
            if (m4 < mbo / 4) w[set][m4] = *reinterpret_cast<const f32x4*>(wl + 256 * m4);
    };

    const long ntiles = (a.N + 63) / 64;
    auto zrow = [&](long tile) -> long {
        const long row = tile * 64 + wave * 16 + r;
        return row < a.N ? row : a.N - 1;
    };
    f32x4 znext[ZB];  // the next tile's latent rows, requested a layer ahead
    auto zfetch = [&](long tile) {
        const float* zp = a.z + zrow(tile) * a.Z + 4 * q;
#pragma unroll
        for (int mb = 0; mb < ZB; ++mb) znext[mb] = *reinterpret_cast<const f32x4*>(zp + 16 * mb);
    };
    // stream start (first tile only): chunk 0 into its buffer, chunk 1 in flight
    gfetch(0);
    zfetch(blockIdx.x);
    __syncthreads();  // (publishes sbias)
    lstore(0);
    gfetch(1);
    __syncthreads();
    frag(0, 0, 0);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row = tile * 64 + wave * 16 + r;
        const bool live = row < a.N;
        int c = 0;  // current chunk of the stream

        // one chunk = 4 MFMA steps (input features 16 mb + 4q + j, j = 0..3) over MBO output blocks.
        //   step j:   request the NEXT step's fragments (next chunk's step 0 after j = 3: made visible by this chunk's
        //             barrier), multiply with the current set;
        //   step 0:   between its MFMAs the chunk in flight goes to the buffer of chunk c + 1 -- last read during chunk
        //             c - 2, which every wavefront finished before the barrier of chunk c - 1 -- and chunk c + 2 is
        //             requested;
        //   after 1:  the one barrier of the chunk.
        auto chunk = [&](auto MBOc, f32x4* acc, auto bsel) {
            constexpr int MBO = decltype(MBOc)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < 3) frag((j + 1) & 1, c, j + 1);
                else frag(0, c + 1, 0);
                if (j != 0) __builtin_amdgcn_sched_barrier(0);
                const float b = bsel(j);
#pragma unroll
                for (int m = 0; m < MBO; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j & 1][m >> 2][m & 3], b, acc[m], 0, 0, 0);
                if (j == 0) {
                    lstore(c + 1);
                    gfetch((c + 2) % NCH);
                    // issue order of this step: the fragment reads, then groups of {4 MFMA, 1 LDS write, address arithmetic, 1 global load}
                    constexpr int NP = MBO / 4;  // 16-byte pieces per thread of a chunk of this layer
                    __builtin_amdgcn_sched_group_barrier(0x100, MBO / 4, 0);
#pragma unroll
                    for (int g = 0; g < NP; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, MBO / NP, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (j == 1) __syncthreads();
            }
            ++c;
        };

        float zin[4 * ZB];
#pragma unroll
        for (int mb = 0; mb < ZB; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j) zin[4 * mb + j] = znext[mb][j];
        f32x4 d1[HB], d2[HB];
        // ---- layer 0: Z -> 512, ReLU ----
#pragma unroll
        for (int m = 0; m < HB; ++m) d1[m] = *reinterpret_cast<const f32x4*>(sbias + 16 * m + 4 * q);
#pragma unroll
        for (int mb = 0; mb < ZB; ++mb)
            chunk(std::integral_constant<int, HB>{}, d1, [&](int j) { return zin[4 * mb + j]; });
#pragma unroll
        for (int m = 0; m < HB; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) d1[m][j] = fmaxf(d1[m][j], 0.f);
        // ---- layer 1: 512 -> 512, ReLU ----
#pragma unroll
        for (int m = 0; m < HB; ++m) d2[m] = *reinterpret_cast<const f32x4*>(sbias + H + 16 * m + 4 * q);
#pragma unroll
        for (int mb = 0; mb < HB; ++mb)
            chunk(std::integral_constant<int, HB>{}, d2, [&](int j) { return d1[mb][j]; });
#pragma unroll
        for (int m = 0; m < HB; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) d2[m][j] = fmaxf(d2[m][j], 0.f);
        // ---- layer 2: 512 -> D, no activation; the accumulators reuse d1's registers ----
        zfetch(tile + gridDim.x);
        f32x4* out = d1;
#pragma unroll
        for (int m = 0; m < DB; ++m) out[m] = *reinterpret_cast<const f32x4*>(sbias + 2 * H + 16 * m + 4 * q);
#pragma unroll
        for (int mb = 0; mb < HB; ++mb)
            chunk(std::integral_constant<int, DB>{}, out, [&](int j) { return d2[mb][j]; });
        if (live) {
#pragma unroll
            for (int m = 0; m < DB; ++m) *reinterpret_cast<f32x4*>(a.x_hat + row * a.D + 16 * m + 4 * q) = out[m];
        }
        const int b0 = base[NCH % 3], b1 = base[(NCH + 1) % 3], b2 = base[(NCH + 2) % 3];
        base[0] = b0; base[1] = b1; base[2] = b2;
    }
}

size_t decode_fused_pack_floats(int Z, int D) { return (size_t)H * Z + (size_t)H * H + (size_t)D * H; }

bool decode_fused_ok(int64_t N, int Z, int D) {
    // (read per call: the parity tests run both paths in one process)
    const bool off = getenv("PSVAE_DECODE_UNFUSED") != nullptr;
    const long minrows = getenv("PSVAE_DECODE_FUSED_MIN") ? atol(getenv("PSVAE_DECODE_FUSED_MIN")) : 16384;
    return !off && N >= minrows && (Z == 64 || Z == 32 || Z == 128) && (D == 256 || D == 512 || D == 128);
}

template <int ZB, int DB>
static hipError_t launch_decode_fused_t(const DecodeFusedArgs& a, hipStream_t s) {
    constexpr size_t lds = (size_t)(3 * 4 * 4 * 16 * (H / 16) + 2 * H + 16 * DB) * sizeof(float);
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_fused_kernel<ZB, DB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (attr_err != hipSuccess) return attr_err;
    const long ntiles = (a.N + 63) / 64;
    const int grid = (int)(ntiles < 256 ? ntiles : 256);  // persistent: one workgroup per CU walks the row tiles
    hipLaunchKernelGGL((decode_fused_kernel<ZB, DB>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

// pack the three weight matrices into `wp` (decode_fused_pack_floats floats) and run the fused kernel
hipError_t launch_decode_fused(int64_t N, int Z, int D, const float* z, const float* const* dec6, float* x_hat, float* wp,
                               hipStream_t s) {
    float* wp0 = wp;
    float* wp1 = wp0 + (size_t)H * Z;
    float* wp2 = wp1 + (size_t)H * H;
    hipLaunchKernelGGL(pack_decode_weights_kernel, dim3(128), dim3(256), 0, s, dec6[0], wp0, H, Z);
    hipLaunchKernelGGL(pack_decode_weights_kernel, dim3(256), dim3(256), 0, s, dec6[2], wp1, H, H);
    hipLaunchKernelGGL(pack_decode_weights_kernel, dim3(256), dim3(256), 0, s, dec6[4], wp2, D, H);
    DecodeFusedArgs a{};
    a.z = z; a.x_hat = x_hat; a.wp = wp; a.bias0 = dec6[1]; a.bias1 = dec6[3]; a.bias2 = dec6[5];
    a.N = N; a.Z = Z; a.D = D;
#define PSVAE_DF(ZB, DB) if (Z == 16 * ZB && D == 16 * DB) return launch_decode_fused_t<ZB, DB>(a, s)
    PSVAE_DF(4, 16); PSVAE_DF(8, 32); PSVAE_DF(2, 16); PSVAE_DF(4, 8); PSVAE_DF(4, 32); PSVAE_DF(8, 16);
    PSVAE_DF(2, 8); PSVAE_DF(2, 32); PSVAE_DF(8, 8);
#undef PSVAE_DF
    return hipErrorInvalidValue;
}

// ===================================================================================================================
// bf16 variant (psvae_decode_bf16: bf16 operands and hidden activations, fp32 accumulation / biases / x_hat).
// Same transposed scheme on v_mfma_f32_16x16x32_bf16: lane (r, q) supplies 8 k slots of W row 16 m + r (A) and of
// activation row r (B) and receives outputs 16 m + 4q + j of row r.  The k slots of an MFMA can be assigned freely as long
// as both operands agree, so slot (q, e) of k step kb is feature 32 kb + 16 (e / 4) + 4 q + (e % 4): a lane's B operand is
// then its own accumulator blocks 2 kb and 2 kb + 1 of the previous layer, ReLU'd and rounded to bf16 in place
// (v_cvt_pk_bf16_f32 + v_pk_max_i16) -- again no shuffle between layers -- and the permutation is baked into the
// packed weights  Wp[kb][m][lane][e] = bf16(W[16 m + r][32 kb + 16 (e / 4) + 4 q + (e % 4)]).
// An MFMA here is 16 cycles for a 16-byte fragment read, so every fragment is used for TWO row groups (a wavefront owns
// 32 rows: 256 accumulator + 128 packed-activation registers): 128 B/clk/CU of LDS reads against the 256 the array
// delivers.  Chunk = one k step (32 KB for 512 outputs), three LDS buffers, one barrier per chunk as above.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 nbf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void pack_decode_weights_bf16_kernel(const float* __restrict__ W,
                                                                       unsigned short* __restrict__ Wp, int OUT, int IN) {
    const int MB = OUT / 16;
    const long n = (long)OUT * IN;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int e = (int)(i & 7), r = (int)((i >> 3) & 15), q = (int)((i >> 7) & 3);
        const long t = i >> 9;
        const int m = (int)(t % MB), kb = (int)(t / MB);
        const __bf16 b = (__bf16)W[(size_t)(16 * m + r) * IN + 32 * kb + 16 * (e >> 2) + 4 * q + (e & 3)];
        Wp[i] = *reinterpret_cast<const unsigned short*>(&b);
    }
}

// two fp32 accumulator blocks -> one B operand: (ReLU,) round to nearest even, pack
template <bool RELU>
__device__ __forceinline__ bf16x8 pack_bf16x8(const f32x4& lo, const f32x4& hi) {
    union { bf16x8 v; s16x2 p[4]; } u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x4& s = h ? hi : lo;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            nbf16x2 b = __builtin_convertvector(f32x2{s[2 * t], s[2 * t + 1]}, nbf16x2);
            s16x2 v = *reinterpret_cast<s16x2*>(&b);
            if (RELU) v = __builtin_elementwise_max(v, s16x2{0, 0});  // sign bit set <=> negative int16
            u.p[2 * h + t] = v;
        }
    }
    return u.v;
}

#ifdef PSVAE_STAMP
__device__ unsigned long long g_dstamp[8];
#endif

struct DecodeFusedBfArgs {
    const float* z;            // [N, Z] fp32 (rounded to bf16 on load, as the per-layer path's cast does)
    float* x_hat;              // [N, D]
    const unsigned short* wp;  // packed bf16 weights: layer 0 | layer 1 | layer 2
    const float* bias0;
    const float* bias1;
    const float* bias2;
    long N;
    int Z, D;
};

// s_waitcnt vmcnt(n) with n known after unrolling (the asm immediate must be a literal)
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
#define PSVAE_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        PSVAE_VM(0) PSVAE_VM(4) PSVAE_VM(8) PSVAE_VM(12) PSVAE_VM(16) PSVAE_VM(20) PSVAE_VM(24)
#undef PSVAE_VM
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// The weight stream of the bf16 kernel: a chunk lasts 64 MFMAs x 16 cycles, far less than an L2 round trip, and the
// register file has no room for a second chunk in flight -- the chunks go global -> LDS directly (LDS-DMA,
// global_load_lds_dwordx4: one 1 KB piece per wave-instruction, no VGPRs), THREE chunks ahead into a ring of four
// buffers.  hipcc does not see these loads (inline asm), so the waits are counted by hand: loads return in order, so
// "my pieces of chunk c + 1 have landed" is vmcnt(pieces issued after them); pending x_hat stores only make a counted wait
// conservative.  The latent rows of the next tile are asm loads too (anything hipcc counts would drain the DMA queue
// with a vmcnt(0)); they are ordered behind a DMA issue so that only the chunk issued AFTER them queues behind their HBM
// latency.
template <int ZK, int DB, int GS>  // Z = 32 ZK, D = 16 DB, hidden = 512; GS output blocks per fragment group
__global__ __launch_bounds__(256, 1) void decode_fused_bf16_kernel(const DecodeFusedBfArgs a) {
    constexpr int HB = H / 16, HK = H / 32;        // 32 output blocks, 16 k steps
    constexpr int CHB = HB * 1024;                 // bytes of a chunk with 512 outputs (one k step)
    constexpr int NCH = ZK + HK + HK;
    constexpr int NZ = 2 * ZK * 2;                 // z loads per lane and tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* wbuf = reinterpret_cast<char*>(lds);     // [4][CHB]
    float* sbias = lds + 4 * CHB / 4;              // [512 + 512 + D]
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < H; i += 256) { sbias[i] = a.bias0[i]; sbias[H + i] = a.bias1[i]; }
    for (int i = tid; i < 16 * DB; i += 256) sbias[2 * H + i] = a.bias2[i];

    auto ch_mbo = [&](int c) -> int { return c < ZK + HK ? HB : DB; };
    auto ch_off = [&](int c) -> long {  // byte offset of chunk c in the packed weights
        return c < ZK + HK ? (long)c * CHB : (long)(ZK + HK) * CHB + (long)(c - ZK - HK) * DB * 1024;
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wbuf;
    unsigned base[4] = {0u, (unsigned)CHB, 2u * CHB, 3u * CHB};  // chunk u of the cyclic stream lives at base[u % 4]
    const unsigned voff = 16u * tid;
    auto dma_piece = [&](int u, int i) {  // piece i of chunk u of the stream (u may run into the next tile)
        const int cc = u % NCH;
        if (i >= ch_mbo(cc) / 4) return;
        const char* src = reinterpret_cast<const char*>(a.wp) + ch_off(cc);
        const unsigned dst = lds0 + base[u % 4] + 1024u * wave + 4096u * i;
        const unsigned vo = voff + 4096u * i;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo), "s"(src), "s"(dst) : "memory");
    };
    auto dma = [&](int u) {
#pragma unroll
        for (int i = 0; i < HB / 4; ++i) dma_piece(u, i);
    };
    bf16x8 w[2][GS];  // fragment groups of GS output blocks: the one being multiplied and the one after it
    auto fragg = [&](int set, int u, int g) {
        const char* wl = wbuf + base[u % 4] + g * (GS * 1024) + lane * 16;
#pragma unroll
        for (int i = 0; i < GS; ++i) w[set][i] = *reinterpret_cast<const bf16x8*>(wl + 1024 * i);
    };

    const long ntiles = (a.N + 127) / 128;
    auto zrow = [&](long tile, int g) -> long {
        const long row = tile * 128 + wave * 32 + 16 * g + r;
        return row < a.N ? row : a.N - 1;
    };
    f32x4 zn[2][ZK][2];  // the next tile's latent rows (fp32)
    auto zfetch = [&](long tile) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float* zp = a.z + zrow(tile, g) * a.Z + 4 * q;
#pragma unroll
            for (int kb = 0; kb < ZK; ++kb)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(zn[g][kb][h]) : "v"(zp + 32 * kb + 16 * h) : "memory");
        }
    };
    auto zfence = [&]() {  // no consumer of zn above this point
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int kb = 0; kb < ZK; ++kb) asm volatile("" : "+v"(zn[g][kb][0]), "+v"(zn[g][kb][1]));
    };
    zfetch(blockIdx.x);
    dma(0);
    dma(1);
    dma(2);
    wait_vmcnt(0);
    __syncthreads();  // (also publishes sbias; nothing is in flight here)
    fragg(0, 0, 0);
#ifdef PSVAE_STAMP
    // cycle stamps of the workgroup's last tile, unconditional (a branch per stamp cuts the block and costs ~100 spills), kept in SGPRs (the kernel has no vector registers to spare: a spill would break the
    // asm loads' contract)
    unsigned long long st_[7] = {0, 0, 0, 0, 0, 0, 0};
#define DSTAMP(K) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_[K]))
#else
#define DSTAMP(K) do {} while (0)
#endif
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int c = 0;
        f32x4 acc[2][HB];
        bf16x8 ap[2][HK];
        DSTAMP(0);
        // one chunk = one k step over MBO output blocks, both row groups, in fragment groups of 8 blocks:
        //   group g: request the next group's fragments (the next chunk's group 0 after the last), multiply;
        //   after group 0: wait for MY pieces of chunk c + 1, the chunk's one barrier (every wavefront's pieces of c + 1
        //   are in LDS, every wavefront is done with chunk c - 1), then chunk c + 3 is requested into c - 1's buffer.
        // `zc`: the chunk after whose DMA issue the next tile's latent rows are requested
        auto chunk = [&](auto MBOc, const bf16x8& b0, const bf16x8& b1, int zc) {
            constexpr int MBO = decltype(MBOc)::value;
            constexpr int G = MBO / GS;
            static_assert(G % 2 == 0, "the fragment sets alternate by group parity");
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g + 1 < G) fragg((g + 1) & 1, c, g + 1);
                else fragg(0, c + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    acc[0][GS * g + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[g & 1][i], b0, acc[0][GS * g + i], 0, 0, 0);
                    acc[1][GS * g + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[g & 1][i], b1, acc[1][GS * g + i], 0, 0, 0);
                    if (g >= 1 && (g - 1) * GS + i < HB / 4) {  // chunk c + 3's pieces go out between the MFMAs after the barrier
                        dma_piece(c + 3, (g - 1) * GS + i);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (g == G - 1) {  // a short chunk (D = 128) has fewer slots than a 512-output chunk has pieces
#pragma unroll
                    for (int pce = (G - 1) * GS; pce < HB / 4; ++pce) dma_piece(c + 3, pce);
                }
                if (c == zc && g == (HB / 4 + GS - 1) / GS) zfetch(tile + gridDim.x);  // (right behind the last piece)
                __builtin_amdgcn_sched_barrier(0);
                if (g == 0) {
                    // outstanding behind chunk c + 1's pieces: chunk c + 2's, and the latent rows if they were requested
                    // after chunk c + 1's or c + 2's issue
                    wait_vmcnt(ch_mbo((c + 2) % NCH) / 4 + ((c == zc + 1 || c == zc + 2) ? NZ : 0));
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            ++c;
        };
        constexpr int ZC = ZK;  // first chunk of layer 1
        // ---- layer 0: Z -> 512 ----
        zfence();
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int kb = 0; kb < ZK; ++kb) ap[g][kb] = pack_bf16x8<false>(zn[g][kb][0], zn[g][kb][1]);
#pragma unroll
        for (int m = 0; m < HB; ++m) acc[0][m] = acc[1][m] = *reinterpret_cast<const f32x4*>(sbias + 16 * m + 4 * q);
#pragma unroll
        for (int kb = 0; kb < ZK; ++kb) chunk(std::integral_constant<int, HB>{}, ap[0][kb], ap[1][kb], ZC);
        DSTAMP(1);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int kb = 0; kb < HK; ++kb) ap[g][kb] = pack_bf16x8<true>(acc[g][2 * kb], acc[g][2 * kb + 1]);
        DSTAMP(2);
        // ---- layer 1: 512 -> 512 ----
#pragma unroll
        for (int m = 0; m < HB; ++m) acc[0][m] = acc[1][m] = *reinterpret_cast<const f32x4*>(sbias + H + 16 * m + 4 * q);
#pragma unroll
        for (int kb = 0; kb < HK; ++kb) chunk(std::integral_constant<int, HB>{}, ap[0][kb], ap[1][kb], ZC);
        DSTAMP(3);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int kb = 0; kb < HK; ++kb) ap[g][kb] = pack_bf16x8<true>(acc[g][2 * kb], acc[g][2 * kb + 1]);
        DSTAMP(4);
        // ---- layer 2: 512 -> D, fp32 out ----
#pragma unroll
        for (int m = 0; m < DB; ++m) acc[0][m] = acc[1][m] = *reinterpret_cast<const f32x4*>(sbias + 2 * H + 16 * m + 4 * q);
#pragma unroll
        for (int kb = 0; kb < HK; ++kb) chunk(std::integral_constant<int, DB>{}, ap[0][kb], ap[1][kb], ZC);
        DSTAMP(5);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const long row = tile * 128 + wave * 32 + 16 * g + r;
            if (row < a.N) {
#pragma unroll
                for (int m = 0; m < DB; ++m) *reinterpret_cast<f32x4*>(a.x_hat + row * a.D + 16 * m + 4 * q) = acc[g][m];
            }
        }
        const unsigned b0 = base[NCH % 4], b1 = base[(NCH + 1) % 4], b2 = base[(NCH + 2) % 4], b3 = base[(NCH + 3) % 4];
        base[0] = b0; base[1] = b1; base[2] = b2; base[3] = b3;
        DSTAMP(6);
#ifdef PSVAE_STAMP
#endif
    }
#ifdef PSVAE_STAMP
    if (tid == 0 && blockIdx.x == 7)
        for (int i = 0; i < 7; ++i) g_dstamp[i] = st_[i];
#endif
    wait_vmcnt(0);  // the stream ran ahead into a tile that does not exist: its DMA must not outlive the workgroup's LDS
}

bool decode_fused_bf16_ok(int64_t N, int Z, int D) {
    const bool off = getenv("PSVAE_DECODE_UNFUSED") != nullptr;
    const long minrows = getenv("PSVAE_DECODE_FUSED_MIN") ? atol(getenv("PSVAE_DECODE_FUSED_MIN")) : 32768;
    // Z = 128 does not fit: 16 more registers of latent rows in flight spill, and a spilled asm-load destination is
    // garbage (hipcc stores it before the data has landed)
    return !off && N >= minrows && (Z == 64 || Z == 32) && (D == 256 || D == 512 || D == 128);
}

template <int ZK, int DB, int GS>
static hipError_t launch_decode_fused_bf16_t(const DecodeFusedBfArgs& a, hipStream_t s) {
    constexpr size_t lds = (size_t)4 * (H / 16) * 1024 + (size_t)(2 * H + 16 * DB) * sizeof(float);
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_fused_bf16_kernel<ZK, DB, GS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (attr_err != hipSuccess) return attr_err;
    const long ntiles = (a.N + 127) / 128;
    const int grid = (int)(ntiles < 256 ? ntiles : 256);
    hipLaunchKernelGGL((decode_fused_bf16_kernel<ZK, DB, GS>), dim3(grid), dim3(256), lds, s, a);
#ifdef PSVAE_STAMP
    {
        unsigned long long h[8];
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dstamp), sizeof(h)) == hipSuccess) {
            printf("decode bf16 stamps (cycles): L0 | relu+pack | L1 | relu+pack | L2 | stores:");
            for (int i = 1; i < 7; ++i) printf(" %llu", h[i] - h[i - 1]);
            printf("\n");
        }
    }
#endif
    return hipGetLastError();
}

// pack (cast + permute) the three weight matrices into `wp` ((H Z + H H + D H) bf16) and run the fused kernel
hipError_t launch_decode_fused_bf16(int64_t N, int Z, int D, const float* z, const float* const* dec6, float* x_hat,
                                    unsigned short* wp, hipStream_t s) {
    unsigned short* wp0 = wp;
    unsigned short* wp1 = wp0 + (size_t)H * Z;
    unsigned short* wp2 = wp1 + (size_t)H * H;
    hipLaunchKernelGGL(pack_decode_weights_bf16_kernel, dim3(128), dim3(256), 0, s, dec6[0], wp0, H, Z);
    hipLaunchKernelGGL(pack_decode_weights_bf16_kernel, dim3(256), dim3(256), 0, s, dec6[2], wp1, H, H);
    hipLaunchKernelGGL(pack_decode_weights_bf16_kernel, dim3(256), dim3(256), 0, s, dec6[4], wp2, D, H);
    DecodeFusedBfArgs a{};
    a.z = z; a.x_hat = x_hat; a.wp = wp; a.bias0 = dec6[1]; a.bias1 = dec6[3]; a.bias2 = dec6[5];
    a.N = N; a.Z = Z; a.D = D;
    static const int gs = getenv("PSVAE_DECODE_BF16_GS") ? atoi(getenv("PSVAE_DECODE_BF16_GS")) : 8;
#define PSVAE_DFB(ZK, DB)                                                                        \
    if (Z == 32 * ZK && D == 16 * DB)                                                            \
        return (gs == 4 || DB == 8) ? launch_decode_fused_bf16_t<ZK, DB, 4>(a, s)                \
                                    : launch_decode_fused_bf16_t<ZK, (DB == 8 ? 16 : DB), 8>(a, s)
    PSVAE_DFB(2, 16); PSVAE_DFB(1, 16); PSVAE_DFB(2, 8); PSVAE_DFB(2, 32); PSVAE_DFB(1, 8); PSVAE_DFB(1, 32);
#undef PSVAE_DFB
    return hipErrorInvalidValue;
}

}  // namespace psvae
