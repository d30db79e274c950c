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
#include <type_traits>

#include "psvae_common.h"
#include "psvae_device.h"

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
    const float* z;       // [N, Z]; NULL = draw z ~ N(0, I) here: element e of the flat [N, Z] array is normal (e % 4) of
                          // Philox block offset + e / 4 under `seed` -- exactly what psvae_philox_normal_f32 would have
                          // written (inference.py:22-23: torch.randn((num_samples, dim))), without the launch and the round trip
    float* z_out;         // nullable: receives the z that was read or drawn
    unsigned long long seed, offset;
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
    const bool draw = a.z == nullptr;
    auto zfetch = [&](long tile) {
        const long zr = zrow(tile);
        if (draw) {  // ZB Philox blocks per lane and tile (~1 % of the tile's cycles, issued ahead of layer 2's MFMAs)
            const long row = tile * 64 + wave * 16 + r;
#pragma unroll
            for (int mb = 0; mb < ZB; ++mb) {
                const unsigned long long e = (unsigned long long)zr * a.Z + 16 * mb + 4 * q;
                znext[mb] = philox_normal4_hw(a.offset + (e >> 2), a.seed);
                if (a.z_out && row < a.N) *reinterpret_cast<f32x4*>(a.z_out + e) = znext[mb];
            }
            return;
        }
        const float* zp = a.z + zr * a.Z + 4 * q;
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
    static std::atomic<unsigned long long> lds_ok{0};  // per device: psvae_common.h lds_opt_in
    if (hipError_t attr_err = lds_opt_in(lds_ok, reinterpret_cast<const void*>(&decode_fused_kernel<ZB, DB>), 160 * 1024))
        return attr_err;
    const long ntiles = (a.N + 63) / 64;
    const int grid = (int)(ntiles < 256 ? ntiles : 256);  // persistent: one workgroup per CU walks the row tiles
    hipLaunchKernelGGL((decode_fused_kernel<ZB, DB>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

// pack the three weight matrices into `wp` (decode_fused_pack_floats floats) and run the fused kernel
hipError_t launch_decode_fused(int64_t N, int Z, int D, const float* z, const float* const* dec6, float* x_hat, float* wp,
                               hipStream_t s, unsigned long long seed, unsigned long long offset, float* z_out) {
    float* wp0 = wp;
    float* wp1 = wp0 + (size_t)H * Z;
    float* wp2 = wp1 + (size_t)H * H;
    hipLaunchKernelGGL(pack_decode_weights_kernel, dim3(128), dim3(256), 0, s, dec6[0], wp0, H, Z);
    hipLaunchKernelGGL(pack_decode_weights_kernel, dim3(256), dim3(256), 0, s, dec6[2], wp1, H, H);
    hipLaunchKernelGGL(pack_decode_weights_kernel, dim3(256), dim3(256), 0, s, dec6[4], wp2, D, H);
    DecodeFusedArgs a{};
    a.z = z; a.x_hat = x_hat; a.wp = wp; a.bias0 = dec6[1]; a.bias1 = dec6[3]; a.bias2 = dec6[5];
    a.N = N; a.Z = Z; a.D = D; a.seed = seed; a.offset = offset; a.z_out = z ? nullptr : z_out;
#define PSVAE_DF(ZB, DB) if (Z == 16 * ZB && D == 16 * DB) return launch_decode_fused_t<ZB, DB>(a, s)
    PSVAE_DF(4, 16); PSVAE_DF(8, 32); PSVAE_DF(2, 16); PSVAE_DF(4, 8); PSVAE_DF(4, 32); PSVAE_DF(8, 16);
    PSVAE_DF(2, 8); PSVAE_DF(2, 32); PSVAE_DF(8, 8);
#undef PSVAE_DF
    return hipErrorInvalidValue;
}

// ===================================================================================================================
// bf16 variant (psvae_decode_bf16: bf16 operands and hidden activations, fp32 accumulation / biases / x_hat), on
// v_mfma_f32_16x16x32_bf16, transposed as above: lane (r, q) supplies 8 k slots of W row 16 m + r (A) and of activation
// row r (B) and receives outputs 16 m + 4q + j of row r.  The k slots of an MFMA can be assigned freely as long as both
// operands agree, so slot (q, e) of k step kb is feature 32 kb + 16 (e / 4) + 4 q + (e % 4): a lane's B operand is then
// its own accumulator blocks 2 kb and 2 kb + 1 of the previous layer, ReLU'd and rounded to bf16 in place
// (v_cvt_pk_bf16_f32 + v_pk_max_i16) -- again no shuffle between layers -- and the permutation is baked into the
// packed weights.
//
// An MFMA here is 16 cycles, so the fp32 kernel's organisation (a wavefront owns rows, every weight byte goes global ->
// LDS -> registers) does not carry over: built that way (LDS-DMA three chunks ahead into a ring of four 32 KB buffers,
// hand-counted vmcnt, two row groups per fragment read) it ran 1 M rows in 0.84 ms, paying one DMA issue (60-185 cycles)
// per KB of weights and one barrier per k step.  Here a wavefront owns a FEATURE slice instead: the 128 rows of a tile
// are shared by the workgroup's four wavefronts, wavefront w computes output blocks [8w, 8w + 8) of each hidden layer
// (D / 64 blocks of the output layer) for ALL 128 rows.  Its weight fragments are needed by no other wavefront and go
// straight from L2 into registers (plain 16-byte loads, two k steps ahead, each used for 8 row groups); the ACTIVATIONS
// are what is shared: one 128 KB LDS image [k step][row group][lane][8 bf16], written by the wavefront that produced
// those features -- the C-to-B register identity makes the write lane-linear -- and read as B fragments (8 ds_read_b128
// per 64 MFMAs).  Eight barriers per tile, no DMA, no asm loads; x_hat leaves through the (then dead) image as whole rows.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 nbf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

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
    const float* z;            // [N, Z] fp32 (rounded to bf16 on load, as the per-layer path's cast does); NULL = drawn
                               // in the kernel (see DecodeFusedArgs::z)
    float* z_out;              // nullable (drawing kernels): receives the fp32 z
    unsigned long long seed, offset;
    float* x_hat;              // [N, D]
    const unsigned short* wp;  // packed bf16 weights: layer 0 | layer 1 | layer 2
    const float* bias0;
    const float* bias1;
    const float* bias2;
    long N;
    int Z, D;
};

// Wp[layer][wavefront w][k step kb][block mi][lane][e] = bf16(W[16 (MBW w + mi) + r][32 kb + 16 (e / 4) + 4 q + (e % 4)])
__global__ __launch_bounds__(256) void pack_decode_weights_bf16w_kernel(const float* __restrict__ W,
                                                                        unsigned short* __restrict__ Wp, int OUT, int IN) {
    const int MBW = OUT / 64, KB = IN / 32;  // blocks per wavefront, k steps
    const long n = (long)OUT * IN;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int e = (int)(i & 7), r = (int)((i >> 3) & 15), q = (int)((i >> 7) & 3);
        const long t = i >> 9;
        const int mi = (int)(t % MBW), kb = (int)((t / MBW) % KB), w = (int)(t / ((long)MBW * KB));
        const int m = MBW * w + mi;
        const __bf16 b = (__bf16)W[(size_t)(16 * m + r) * IN + 32 * kb + 16 * (e >> 2) + 4 * q + (e & 3)];
        Wp[i] = *reinterpret_cast<const unsigned short*>(&b);
    }
}

// RGD: row groups of 16 per tile -- 8 (128 rows: one workgroup per CU) or 4 (64 rows: a 64 KB image, two workgroups per CU;
// one's publishes and stores then run under the other's MFMAs, at twice the weight intake per row)
template <int ZK, int DBW, bool DRAW = false, int RGD = 8>  // Z = 32 ZK, D = 64 DBW, hidden = 512; DRAW: z comes from Philox, not memory
__global__ __launch_bounds__(256, RGD == 4 ? 2 : 1) void decode_rows_bf16_kernel(const DecodeFusedBfArgs a) {
    static_assert(RGD == 8 || RGD == 4, "128- or 64-row tiles");
    constexpr int TR = 16 * RGD;                     // rows per tile
    constexpr int RGW = RGD / 4;                     // row groups whose latent rows a wavefront loads / draws
    constexpr int HK = H / 32;                       // 16 k steps of a hidden input
    constexpr int S = ZK + HK + HK;                  // k steps of the whole weight stream per tile
    constexpr int NA = RGD == 8 ? 3 : 2;             // weight-fragment ring: this step and two (one: 64-row tiles, 256 registers) ahead
    constexpr int RPP = DBW == 8 ? TR / 2 : TR;      // rows per x_hat staging pass (D = 512: two passes)
    constexpr int RS = 256 * DBW + 16;               // bytes of a staged row
    constexpr int IMG = RPP * RS > TR * 1024 ? RPP * RS : TR * 1024;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* act = reinterpret_cast<char*>(lds);        // [16 k steps][8 row groups][64 lanes][16 B] = 128 KB; x_hat staging
    float* sbias = lds + IMG / 4;                    // [512 + 512 + D]
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < H; i += 256) { sbias[i] = a.bias0[i]; sbias[H + i] = a.bias1[i]; }
    for (int i = tid; i < 64 * DBW; i += 256) sbias[2 * H + i] = a.bias2[i];

    // this wavefront's weight stream: step s has MB(s) fragments of 1 KB (16 B per lane), wave-major inside a layer
    auto mb_of = [&](int s) -> int { return s < ZK + HK ? 8 : DBW; };
    const unsigned short* wl0 = a.wp + (size_t)wave * (ZK * 8 * 512) + lane * 8;
    const unsigned short* wl1 = a.wp + (size_t)H * a.Z + (size_t)wave * (HK * 8 * 512) + lane * 8;
    const unsigned short* wl2 = a.wp + (size_t)H * a.Z + (size_t)H * H + (size_t)wave * (HK * DBW * 512) + lane * 8;
    bf16x8 wa[NA][8];
    // (one running pointer, opaque to the optimiser and typed global: per-fragment addresses are loop invariants the
    //  compiler would otherwise precompute and spill by the hundred -- see the fp32 kernel)
    typedef const __attribute__((address_space(1))) unsigned short gu16;
    typedef const __attribute__((address_space(1))) bf16x8 gbf16x8;
    gu16* wsrc = nullptr;
    auto wfetch = [&](int s) {  // s in [0, S), called in stream order: fragments of k step s into ring slot s % NA
        if (s == 0) wsrc = (gu16*)wl0;
        if (s == ZK) wsrc = (gu16*)wl1;
        if (s == ZK + HK) wsrc = (gu16*)wl2;
        asm volatile("" : "+v"(wsrc));
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
            if (mi < mb_of(s)) wa[s % NA][mi] = *(gbf16x8*)(wsrc + mi * 512);
        wsrc += mb_of(s) * 512;
    };
    bf16x8 bb[2][RGD];
    auto bfetch = [&](int set, int kb) {  // B fragments of k step kb: all row groups
        const char* src = act + kb * (RGD * 1024) + lane * 16;
#pragma unroll
        for (int g = 0; g < RGD; ++g) bb[set][g] = *reinterpret_cast<const bf16x8*>(src + 1024 * g);
    };

    const long ntiles = (a.N + TR - 1) / TR;
    f32x4 zn[RGW][ZK][2];  // the next tile's latent rows of this wavefront's row groups (fp32)
    auto zfetch = [&](long tile) {
#pragma unroll
        for (int g = 0; g < RGW; ++g) {
            long row = tile * TR + 16 * RGW * wave + 16 * g + r;
            row = row < a.N ? row : a.N - 1;
            const float* zp = a.z + row * a.Z + 4 * q;
#pragma unroll
            for (int kb = 0; kb < ZK; ++kb) {
                zn[g][kb][0] = *reinterpret_cast<const f32x4*>(zp + 32 * kb);
                zn[g][kb][1] = *reinterpret_cast<const f32x4*>(zp + 32 * kb + 16);
            }
        }
    };
    // DRAW: the same rows' z as Philox normals, one block of four per 16-byte piece -- 4 ZK pieces per lane and tile.  The
    // first tile's are made here; the next tile's are made PIECE BY PIECE inside layer 2's k loop (piece p between the
    // MFMAs of k step p), where the vector ALU is otherwise idle: the separate draw launch (94 us per 1 M rows) and the
    // 256 MB round trip of z disappear, the tile pays nothing visible.
    auto zpiece = [&](long tile, auto Pc) {
        constexpr int p = decltype(Pc)::value;
        constexpr int g = p / (2 * ZK), kbz = (p % (2 * ZK)) / 2, h = p % 2;
        const long row = tile * TR + 16 * RGW * wave + 16 * g + r;
        const unsigned long long e = (unsigned long long)(row < a.N ? row : a.N - 1) * a.Z + 32 * kbz + 16 * h + 4 * q;
        zn[g][kbz][h] = philox_normal4_hw(a.offset + (e >> 2), a.seed);
        if (a.z_out && row < a.N) *reinterpret_cast<f32x4*>(a.z_out + e) = zn[g][kbz][h];
    };
    constexpr int NPZ = 2 * RGW * ZK;  // pieces (Philox blocks) per lane and tile: at most 16
    if constexpr (DRAW) {
#define PSVAE_ZP0(P) if constexpr (P < NPZ) zpiece(blockIdx.x, std::integral_constant<int, (P < NPZ ? P : 0)>{})
        PSVAE_ZP0(0); PSVAE_ZP0(1); PSVAE_ZP0(2); PSVAE_ZP0(3); PSVAE_ZP0(4); PSVAE_ZP0(5); PSVAE_ZP0(6); PSVAE_ZP0(7);
        PSVAE_ZP0(8); PSVAE_ZP0(9); PSVAE_ZP0(10); PSVAE_ZP0(11); PSVAE_ZP0(12); PSVAE_ZP0(13); PSVAE_ZP0(14); PSVAE_ZP0(15);
#undef PSVAE_ZP0
    } else {
        zfetch(blockIdx.x);
    }
    constexpr int PD = NA - 1;  // k steps of weight fragments in flight ahead of the MFMAs
#pragma unroll
    for (int i = 0; i < PD; ++i) wfetch(i);
    __syncthreads();  // sbias
#ifdef PSVAE_STAMP
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RSTAMP(K) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_[K]))
#else
#define RSTAMP(K) do {} while (0)
#endif
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        RSTAMP(0);
        // ---- the tile's latent rows -> B image (k steps 0 .. ZK-1 of `act`); every wavefront is past the previous tile's
        //      last read of `act` (barrier at the end of the loop body)
#pragma unroll
        for (int g = 0; g < RGW; ++g)
#pragma unroll
            for (int kb = 0; kb < ZK; ++kb)
                *reinterpret_cast<bf16x8*>(act + (kb * RGD + RGW * wave + g) * 1024 + lane * 16) =
                    pack_bf16x8<false>(zn[g][kb][0], zn[g][kb][1]);
        __syncthreads();
        f32x4 acc[8][RGD];  // [block of this wavefront][row group]
        int s = 0;        // k step of the weight stream
        // one layer: KB k steps over MB blocks x 8 row groups; the bias is the first step's C operand
        auto layer = [&](auto KBc, auto MBc, const float* bias, auto between) {
            constexpr int KB = decltype(KBc)::value, MB = decltype(MBc)::value;
            f32x4 bv[MB];
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) bv[mi] = *reinterpret_cast<const f32x4*>(bias + 16 * (MB * wave + mi) + 4 * q);
            bfetch(0, 0);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb, ++s) {
                __builtin_amdgcn_sched_barrier(0);  // (keeps the compiler from hoisting every step's loads to the top)
                // the stream is cyclic (the tile's last two steps fetch the next tile's first two) when the ring closes,
                // S % NA == 0; otherwise those two fetches wait until the tile's last MFMA has issued (below)
                if (s + PD < S) wfetch(s + PD);
                else if (S % NA == 0) wfetch(s + PD - S);
                if (kb + 1 < KB) bfetch((kb + 1) & 1, kb + 1);
                __builtin_amdgcn_sched_barrier(0);
                between(kb);  // (vector-ALU work the scheduler may interleave with this step's MFMAs)
#pragma unroll
                for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                    for (int g = 0; g < RGD; ++g)
                        acc[mi][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[s % NA][mi], bb[kb & 1][g],
                                                                            kb == 0 ? bv[mi] : acc[mi][g], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // hidden activations: ReLU, round to bf16, into the B image of the next layer (this wavefront produced the features
        // of k steps 4 wave .. 4 wave + 3)
        auto publish = [&]() {
            __syncthreads();  // every wavefront has read the previous image
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 t[RGD];  // (independent convert chains, then their writes: one chain at a time is latency-bound)
#pragma unroll
                for (int g = 0; g < RGD; ++g) t[g] = pack_bf16x8<true>(acc[2 * kk][g], acc[2 * kk + 1][g]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < RGD; ++g)
                    *reinterpret_cast<bf16x8*>(act + ((4 * wave + kk) * RGD + g) * 1024 + lane * 16) = t[g];
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        };
        RSTAMP(1);
        auto nothing = [](int) {};
        layer(std::integral_constant<int, ZK>{}, std::integral_constant<int, 8>{}, sbias, nothing);
        RSTAMP(2);
        publish();
        RSTAMP(3);
        layer(std::integral_constant<int, HK>{}, std::integral_constant<int, 8>{}, sbias + H, nothing);
        RSTAMP(4);
        publish();
        RSTAMP(5);
        if constexpr (DRAW) {
            const long nxt = tile + gridDim.x;
            layer(std::integral_constant<int, HK>{}, std::integral_constant<int, DBW>{}, sbias + 2 * H, [&](int kb) {
                // (kb is a constant after unrolling; 4 ZK <= 16 pieces over the 16 k steps)
#define PSVAE_ZP(P) if (kb == P && P < NPZ) zpiece(nxt, std::integral_constant<int, (P < NPZ ? P : 0)>{})
                PSVAE_ZP(0); PSVAE_ZP(1); PSVAE_ZP(2); PSVAE_ZP(3); PSVAE_ZP(4); PSVAE_ZP(5); PSVAE_ZP(6); PSVAE_ZP(7);
                PSVAE_ZP(8); PSVAE_ZP(9); PSVAE_ZP(10); PSVAE_ZP(11); PSVAE_ZP(12); PSVAE_ZP(13); PSVAE_ZP(14); PSVAE_ZP(15);
#undef PSVAE_ZP
            });
        } else {
            zfetch(tile + gridDim.x);
            layer(std::integral_constant<int, HK>{}, std::integral_constant<int, DBW>{}, sbias + 2 * H, nothing);
        }
        RSTAMP(6);
        if (S % NA != 0) {  // (their latency runs under the stores, the barrier and the latent rows' write)
#pragma unroll
            for (int i = 0; i < PD; ++i) wfetch(i);
        }
        // x_hat: a lane holds 4 consecutive features of 16 different rows per block -- stored as it stands, every store
        // instruction would touch 16 rows with 64 bytes each (store-issue bound: 8.8k of the tile's 48k cycles).  The tile goes
        // through LDS instead (the image is dead once every wavefront has left layer 2; rows padded by 16 B so the 16 r lanes
        // hit different banks) and leaves as whole rows: 1 KB contiguous per store instruction.
#pragma unroll
        for (int ps = 0; ps < TR / RPP; ++ps) {
            __syncthreads();  // image (or the previous pass) no longer read
#pragma unroll
            for (int g = ps * (RPP / 16); g < (ps + 1) * (RPP / 16); ++g)
#pragma unroll
                for (int mi = 0; mi < DBW; ++mi)
                    *reinterpret_cast<f32x4*>(act + (16 * (g - ps * (RPP / 16)) + r) * RS + (16 * (DBW * wave + mi) + 4 * q) * 4) =
                        acc[mi][g];
            __syncthreads();
            // (reads in batches of 8 ahead of their stores: row by row, every store would wait out an LDS round trip)
            constexpr int NI = DBW >= 4 ? (RPP / 4) * (DBW / 4) : RPP / 8;  // store instructions per wavefront and pass
#pragma unroll
            for (int i0 = 0; i0 < NI; i0 += 8) {
                f32x4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int it = i0 + i;
                    const int rl = DBW >= 4 ? (RPP / 4) * wave + it / (DBW >= 4 ? DBW / 4 : 1) : (RPP / 4) * wave + 2 * it + (lane >> 5);
                    const int pc = DBW >= 4 ? it % (DBW >= 4 ? DBW / 4 : 1) : 0;
                    v[i] = *reinterpret_cast<const f32x4*>(act + rl * RS + pc * 1024 + (DBW >= 4 ? lane : (lane & 31)) * 16);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int it = i0 + i;
                    const int rl = DBW >= 4 ? (RPP / 4) * wave + it / (DBW >= 4 ? DBW / 4 : 1) : (RPP / 4) * wave + 2 * it + (lane >> 5);
                    const int pc = DBW >= 4 ? it % (DBW >= 4 ? DBW / 4 : 1) : 0;
                    const long row = tile * TR + ps * RPP + rl;
                    if (row < a.N)
                        *reinterpret_cast<f32x4*>(a.x_hat + row * a.D + pc * 256 + (DBW >= 4 ? lane : (lane & 31)) * 4) = v[i];
                }
            }
        }
        __syncthreads();  // the staged rows have been read before the next tile's latent rows overwrite them
        RSTAMP(7);
    }
#ifdef PSVAE_STAMP
    if (tid == 0 && blockIdx.x == 7)
        for (int i = 0; i < 8; ++i) g_dstamp[i] = st_[i];
#endif
}

template <int ZK, int DBW, bool DRAW, int RGD>
static hipError_t launch_decode_rows_bf16_rg(const DecodeFusedBfArgs& a, hipStream_t s) {
    constexpr int TR = 16 * RGD, RPP = DBW == 8 ? TR / 2 : TR, RS = 256 * DBW + 16;
    constexpr size_t lds = (size_t)(RPP * RS > TR * 1024 ? RPP * RS : TR * 1024) + (size_t)(2 * H + 64 * DBW) * sizeof(float);
    static std::atomic<unsigned long long> lds_ok{0};  // per device: psvae_common.h lds_opt_in
    if (hipError_t attr_err = lds_opt_in(lds_ok, reinterpret_cast<const void*>(&decode_rows_bf16_kernel<ZK, DBW, DRAW, RGD>), 160 * 1024))
        return attr_err;
    const long ntiles = (a.N + TR - 1) / TR;
    const int slots = 256 * (lds <= 80 * 1024 ? 2 : 1);  // persistent: one workgroup per LDS slot of every CU
    const int grid = (int)(ntiles < slots ? ntiles : slots);
    hipLaunchKernelGGL((decode_rows_bf16_kernel<ZK, DBW, DRAW, RGD>), dim3(grid), dim3(256), lds, s, a);
#ifdef PSVAE_STAMP
    {
        unsigned long long h[8];
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dstamp), sizeof(h)) == hipSuccess) {
            printf("decode bf16 (feature-slice form) stamps (cycles): z image | L0 | publish | L1 | publish | L2 | stores:");
            for (int i = 1; i < 8; ++i) printf(" %llu", h[i] - h[i - 1]);
            printf("\n");
        }
    }
#endif
    return hipGetLastError();
}

template <int ZK, int DBW, bool DRAW>
static hipError_t launch_decode_rows_bf16_t(const DecodeFusedBfArgs& a, hipStream_t s) {
    // rows per tile / 16: 4 (default: 64-row tiles, two workgroups per CU: 0.82 ms per 1 M rows) or 8 (128-row tiles, one
    // workgroup per CU, weight ring one k step deeper: 0.85 ms); read per call: a test runs both
    const char* e = getenv("PSVAE_DECODE_BF16_RG");
    if (e && atoi(e) == 8) return launch_decode_rows_bf16_rg<ZK, DBW, DRAW, 8>(a, s);
    return launch_decode_rows_bf16_rg<ZK, DBW, DRAW, 4>(a, s);
}

hipError_t launch_decode_rows_bf16(int64_t N, int Z, int D, const float* z, const float* const* dec6, float* x_hat,
                                   unsigned short* wp, hipStream_t s, unsigned long long seed, unsigned long long offset,
                                   float* z_out) {
    unsigned short* wp0 = wp;
    unsigned short* wp1 = wp0 + (size_t)H * Z;
    unsigned short* wp2 = wp1 + (size_t)H * H;
    hipLaunchKernelGGL(pack_decode_weights_bf16w_kernel, dim3(128), dim3(256), 0, s, dec6[0], wp0, H, Z);
    hipLaunchKernelGGL(pack_decode_weights_bf16w_kernel, dim3(256), dim3(256), 0, s, dec6[2], wp1, H, H);
    hipLaunchKernelGGL(pack_decode_weights_bf16w_kernel, dim3(256), dim3(256), 0, s, dec6[4], wp2, D, H);
    DecodeFusedBfArgs a{};
    a.z = z; a.x_hat = x_hat; a.wp = wp; a.bias0 = dec6[1]; a.bias1 = dec6[3]; a.bias2 = dec6[5];
    a.N = N; a.Z = Z; a.D = D; a.seed = seed; a.offset = offset; a.z_out = z ? nullptr : z_out;
#define PSVAE_DRB(ZK, DBW) if (Z == 32 * ZK && D == 64 * DBW) \
        return z ? launch_decode_rows_bf16_t<ZK, DBW, false>(a, s) : launch_decode_rows_bf16_t<ZK, DBW, true>(a, s)
    PSVAE_DRB(2, 4); PSVAE_DRB(1, 4); PSVAE_DRB(4, 4); PSVAE_DRB(2, 2); PSVAE_DRB(1, 2); PSVAE_DRB(4, 2);
    PSVAE_DRB(2, 8); PSVAE_DRB(1, 8); PSVAE_DRB(4, 8);
#undef PSVAE_DRB
    return hipErrorInvalidValue;
}

bool decode_fused_bf16_ok(int64_t N, int Z, int D) {
    const bool off = getenv("PSVAE_DECODE_UNFUSED") != nullptr;
    const long minrows = getenv("PSVAE_DECODE_FUSED_MIN") ? atol(getenv("PSVAE_DECODE_FUSED_MIN")) : 32768;
    return !off && N >= minrows && (Z == 64 || Z == 32 || Z == 128) && (D == 256 || D == 512 || D == 128);
}

// pack (cast + permute) the three weight matrices into `wp` ((H Z + H H + D H) bf16) and run the fused kernel
hipError_t launch_decode_fused_bf16(int64_t N, int Z, int D, const float* z, const float* const* dec6, float* x_hat,
                                    unsigned short* wp, hipStream_t s, unsigned long long seed, unsigned long long offset,
                                    float* z_out) {
    return launch_decode_rows_bf16(N, Z, D, z, dec6, x_hat, wp, s, seed, offset, z_out);
}

}  // namespace psvae
