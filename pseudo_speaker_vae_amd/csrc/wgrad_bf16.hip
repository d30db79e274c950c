// All weight gradients of a bf16 pass as ONE launch of 128 x 128 tiles: the .weight.grad / .bias.grad that autograd leaves in
// the nine nn.Linear layers of model.py:14-36 after lightning.py:131, and the gradients of the conditional tables (config 3's
// one-hot gender / age inputs: no reference code).
//
//     dW[o][i] = sum_b dY[b][o] X[b][i]          both operands batch-major ("k-major"), as the row chains leave them
//
// What the round-2 kernel (gemm_bf16.hip, `tr` branch) cost per 64-row k tile was the SUM of its phases -- global
// loads into registers, ds_write_b128, barrier, transposed fragment reads with 2-way bank conflicts, MFMAs -- because
// the four wavefronts of a workgroup move in lock step (2.1-2.7 k cycles per tile for 0.5 k cycles of MFMA).
// This kernel removes the register hop and the conflicts and overlaps the rest:
//   * operand tiles (32 or 64 batch rows) go global -> LDS by LDS-DMA (global_load_lds_dwordx4), NBUF tiles deep, counted
//     s_waitcnt vmcnt(n) and ONE raw s_barrier per tile; no VGPR staging, no ds_write.
//   * the LDS image is not rows of the matrix: a DMA instruction writes 1 KB = 64 lanes x 16 B lane-linearly, but every
//     lane's SOURCE address is free, so the image is laid out as the fragment reads want it.  One
//     ds_read_b64_tr_b16 of the 32x32x16 MFMA operand touches 8 batch rows x 64 B (32 features): that block is made 512
//     CONTIGUOUS bytes of LDS, ordered so that the 32 lanes of a half read 256 consecutive bytes (every bank once:
//     conflict-free by construction) and every address is lane part + immediates.  Two neighbouring blocks (64
//     features = one 128-B line per batch row, 8 rows) are one DMA instruction: whole lines from memory.
//   * fragment reads run one 16-deep k step ahead of the MFMAs (two fragment sets); the tile hand-over (wait, barrier,
//     next DMA, first reads of the next tile) sits in front of the last k step's MFMAs.
//   * accumulators leave straight from registers (a 32x32 block's rows are 128-B runs); no epilogue LDS.
// A ragged last tile (batch not a multiple of the tile's rows) takes the same image through registers with zero fill.
#include <atomic>

#include "psvae_common.h"

namespace psvae {

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

#define PSVAE_GLOBAL __attribute__((address_space(1)))
#define PSVAE_LDS __attribute__((address_space(3)))

constexpr int WG_BM = 128, WG_BN = 128;
// (the loader-wavefront form's fixed tile: 64 batch rows; wgrad_tile takes its rows per tile as a template parameter)
constexpr int WG_BK = 64;
constexpr int OP_BYTES = WG_BK * 128 * 2;  // one operand's tile image: 16 KB = 16 DMA instructions
constexpr int TILE_BYTES = 2 * OP_BYTES;   // A image, then B image

// LDS-DMA: 64 lanes x 16 B land at lds_dst + 16 * lane (M0 = wave-uniform LDS byte address); asm, so that hipcc's own
// waitcnt bookkeeping does not see it (it would drain vmcnt to 0 in front of every LDS read): counted by hand below
__device__ __forceinline__ void glds16(const u16* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ v4s trd(unsigned addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((PSVAE_LDS v4s*)(uintptr_t)addr);
}
__device__ __forceinline__ float bf2f_(short u) { return __uint_as_float(((unsigned)(u16)u) << 16); }

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `tiles` of this wave's tiles (PER DMA instructions each) are still in flight; `tiles` is wave-uniform
template <int MAXT, int PER>
__device__ __forceinline__ void wait_vm_tiles(int tiles) {
    if (MAXT >= 3 && tiles >= 3) wait_vm<3 * PER>();
    else if (MAXT >= 2 && tiles == 2) wait_vm<2 * PER>();
    else if (MAXT >= 1 && tiles == 1) wait_vm<PER>();
    else wait_vm<0>();
}

// BK: batch rows per tile (64, or 32: half the LDS per buffer -- three buffers in 48 KB)
template <int NBUF, int BK>
__device__ __forceinline__ void wgrad_tile(const GemmDescBf& d, int tm, int tn, int ksg, unsigned lds0) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int M = d.M, N = d.N, K = d.K;
    const int m0 = tm * WG_BM, n0 = tn * WG_BN;
    constexpr int KS = BK / 16;              // 16-deep k steps per tile
    constexpr int OPB = BK * 256;            // one operand's tile image (bytes)
    constexpr int TILEB = 2 * OPB;           // A image, then B image
    constexpr int NJ = BK / 16;              // DMA instructions per wavefront, operand and tile

    // ---- k tiles of this workgroup (slice ksg of the batch split)
    const int nt_all = (K + BK - 1) / BK;
    const int S = d.ksplit > 1 ? d.ksplit : 1;
    const int per = (nt_all + S - 1) / S;
    const int t_beg = ksg * per;
    const int t_end = (t_beg + per < nt_all ? t_beg + per : nt_all);

    // ---- DMA side: wave w owns the four instructions J = 4 w + j of each operand (k step w of the tile; j = (hi, pair))
    const int cbl = lane >> 5, kh = (lane >> 4) & 1, qq = (lane >> 2) & 3, c16 = (lane >> 1) & 1, ph = lane & 1;
    const u16* pa[NJ];
    const u16* pb[NJ];
    int krow[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int J = NJ * wave + j, hi = (J >> 1) & 1, pair = J & 1;  // instruction J of 4 NJ: k step J >> 2
        krow[j] = 16 * (J >> 2) + 8 * kh + 4 * hi + qq;
        const int col = 64 * pair + 32 * cbl + 16 * c16 + 8 * ph;
        // pieces beyond the row pitch re-read piece 0 of the row: they only feed outputs that are never stored
        const int ca = (m0 + col + 8 <= d.lda) ? m0 + col : m0;
        const int cb = (n0 + col + 8 <= d.ldb) ? n0 + col : n0;
        pa[j] = d.A + ((size_t)t_beg * BK + krow[j]) * d.lda + ca;
        pb[j] = d.B + ((size_t)t_beg * BK + krow[j]) * d.ldb + cb;
    }
    const size_t stepa = (size_t)BK * d.lda, stepb = (size_t)BK * d.ldb;
    const unsigned dma0 = lds0 + 1024u * NJ * (unsigned)wave;  // + buffer + 1024 j (+ OPB for B)
    // tile t -> buffer at buf_off, in four parts (two DMA instructions each) so that the k loop can put them between MFMAs:
    // a wavefront that issues all eight at once sits in the vector-memory issue queue behind the other three wavefronts'
    // instructions (32 KB through a 64 B/clk path) with its MFMAs waiting behind them in program order
    auto issue_part = [&](auto PART, int t, unsigned buf_off) {
        constexpr int part = decltype(PART)::value;
        if ((t + 1) * BK <= K) {
            if constexpr (NJ == 4) {  // 64-row tiles: two instructions per part
                if constexpr (part < 2) {
                    glds16(pa[2 * part], __builtin_amdgcn_readfirstlane(dma0 + buf_off + 2048u * part));
                    glds16(pa[2 * part + 1], __builtin_amdgcn_readfirstlane(dma0 + buf_off + 2048u * part + 1024u));
                } else {
                    glds16(pb[2 * part - 4], __builtin_amdgcn_readfirstlane(dma0 + buf_off + OPB + 2048u * (part - 2)));
                    glds16(pb[2 * part - 3], __builtin_amdgcn_readfirstlane(dma0 + buf_off + OPB + 2048u * (part - 2) + 1024u));
                }
            } else {                  // 32-row tiles: one instruction per part
                if constexpr (part < 2) glds16(pa[part], __builtin_amdgcn_readfirstlane(dma0 + buf_off + 1024u * part));
                else glds16(pb[part - 2], __builtin_amdgcn_readfirstlane(dma0 + buf_off + OPB + 1024u * (part - 2)));
            }
        } else if (part == 0) {
            // ragged tile (the last one of the batch): same image through registers, rows past the batch as zeros
            const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bool in = t * BK + krow[j] < K;
                const bf16x8 va = in ? *reinterpret_cast<const PSVAE_GLOBAL bf16x8*>(reinterpret_cast<uintptr_t>(pa[j])) : zero8;
                const bf16x8 vb = in ? *reinterpret_cast<const PSVAE_GLOBAL bf16x8*>(reinterpret_cast<uintptr_t>(pb[j])) : zero8;
                *(PSVAE_LDS bf16x8*)(uintptr_t)(dma0 + buf_off + 1024u * j + 16u * lane) = va;
                *(PSVAE_LDS bf16x8*)(uintptr_t)(dma0 + buf_off + OPB + 1024u * j + 16u * lane) = vb;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the ds_writes, before the barrier that publishes them)
        }
        if constexpr (part == 3) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                pa[j] += stepa;
                pb[j] += stepb;
            }
        }
    };
    auto issue = [&](int t, unsigned buf_off) {
        issue_part(std::integral_constant<int, 0>{}, t, buf_off);
        issue_part(std::integral_constant<int, 1>{}, t, buf_off);
        issue_part(std::integral_constant<int, 2>{}, t, buf_off);
        issue_part(std::integral_constant<int, 3>{}, t, buf_off);
    };

    // ---- fragment side: lane (g = lane >> 4, i = lane & 15) of a transposed read
    const int g = lane >> 4, i = lane & 15;
    const unsigned lanepart = 16u * (16u * (g >> 1) + 4u * (i >> 2) + 2u * (g & 1) + ((i & 3) >> 1)) + 8u * (i & 1);
    const unsigned aoff = lds0 + lanepart + 1024u * wm;
    const unsigned boff = lds0 + lanepart + OPB + 1024u * wn;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    float csum[2] = {0.f, 0.f};
    const bool do_colsum = (d.colsum != nullptr) && (tn == 0) && (wn == 0);

    bf16x8 fa[2][2], fb[2][2];  // [set][block]
    auto rda = [&](auto SET, unsigned buf_off, int ks, int mb) {
        constexpr int s = decltype(SET)::value;
        const v4s lo = trd(aoff + buf_off + 4096u * ks + 512u * mb), hi = trd(aoff + buf_off + 4096u * ks + 2048u + 512u * mb);
        fa[s][mb] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto rdb = [&](auto SET, unsigned buf_off, int ks, int nb) {
        constexpr int s = decltype(SET)::value;
        const v4s lo = trd(boff + buf_off + 4096u * ks + 512u * nb), hi = trd(boff + buf_off + 4096u * ks + 2048u + 512u * nb);
        fb[s][nb] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto rd = [&](auto SET, unsigned buf_off, int ks) {
        rda(SET, buf_off, ks, 0); rda(SET, buf_off, ks, 1); rdb(SET, buf_off, ks, 0); rdb(SET, buf_off, ks, 1);
    };
    auto mm = [&](auto SET, int mb, int nb) {
        constexpr int s = decltype(SET)::value;
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][mb], fb[s][nb], acc[mb][nb], 0, 0, 0);
    };
    auto colsum_add = [&](auto SET) {
        constexpr int s = decltype(SET)::value;
        if (do_colsum) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int e = 0; e < 8; ++e) csum[mb] += bf2f_(fa[s][mb][e]);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
#define SB() __builtin_amdgcn_sched_barrier(0)

#ifdef PSVAE_STAMP
    unsigned long long st_[64];
    int nst_ = 0;
#define STAMP() do { if (nst_ < 64) st_[nst_++] = __builtin_amdgcn_s_memtime(); } while (0)
    STAMP();
#else
#define STAMP() do {} while (0)
#endif

    if (t_beg < t_end) {
        // prologue: up to NBUF tiles in flight, wait for the first
        int issued = 0;
#pragma unroll
        for (int b = 0; b < NBUF; ++b)
            if (t_beg + b < t_end) { issue(t_beg + b, (unsigned)b * TILEB); ++issued; }
        wait_vm_tiles<NBUF - 1, 2 * NJ>(issued - 1);
        __builtin_amdgcn_s_barrier();
        SB();
        rd(I0{}, 0u, 0);
        SB();
    }
    STAMP();
    // One 16-deep k step: the four MFMAs of fragment set CUR with the eight transposed reads of the next step's set
    // between them (two per MFMA: they issue while the MFMA before them runs; with all reads in front of all MFMAs the
    // four wavefronts' reads -- 16 KB -- held the MFMA pipe idle for as long as the MFMAs take).
    auto kstep = [&](auto CUR, auto NXT, unsigned buf_off, int ksn) {
        mm(CUR, 0, 0); SB(); rda(NXT, buf_off, ksn, 0); SB();
        mm(CUR, 0, 1); SB(); rda(NXT, buf_off, ksn, 1); SB();
        mm(CUR, 1, 0); SB(); rdb(NXT, buf_off, ksn, 0); SB();
        mm(CUR, 1, 1); SB(); rdb(NXT, buf_off, ksn, 1); colsum_add(CUR); SB();
    };
    // iteration of tile t in buffer BUF: fragment set 0 holds (or is receiving) its k step 0
    auto iter = [&](auto BUFC, int t) {
        constexpr int BUF = decltype(BUFC)::value;
        constexpr unsigned bo = (unsigned)BUF * TILEB, bnext = (unsigned)((BUF + 1) % NBUF) * TILEB;
        kstep(I0{}, I1{}, bo, 1);
        if constexpr (KS == 4) {
            kstep(I1{}, I0{}, bo, 2);
            kstep(I0{}, I1{}, bo, 3);
        }
        STAMP();
        // last k step = hand-over: behind its first MFMA all of tile t is in registers (lgkmcnt 0) and tile t + 1 has landed
        // for everybody (counted vmcnt, barrier); the DMA of tile t + NBUF into this buffer and the first reads of tile
        // t + 1 go between the remaining MFMAs
        mm(I1{}, 0, 0); SB();
#ifdef PSVAE_WG_NODMA  // diagnostic: no DMA inside the loop (stale tiles: wrong results, the loop's own time)
        const bool more = t + 1 < t_end, load = false;
#else
        const bool more = t + 1 < t_end, load = t + NBUF < t_end;
#endif
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int after = t_end - 2 - t;  // tiles behind t + 1; at most NBUF - 2 of them have been issued
            wait_vm_tiles<NBUF - 2, 2 * NJ>(after < NBUF - 2 ? after : NBUF - 2);
            STAMP();
            __builtin_amdgcn_s_barrier();
            STAMP();
            if (load) issue_part(std::integral_constant<int, 0>{}, t + NBUF, bo);
            rda(I0{}, bnext, 0, 0);
        }
        SB(); mm(I1{}, 0, 1); SB();
        if (more) {
            if (load) issue_part(std::integral_constant<int, 1>{}, t + NBUF, bo);
            rda(I0{}, bnext, 0, 1);
        }
        SB(); mm(I1{}, 1, 0); SB();
        if (more) {
            if (load) issue_part(std::integral_constant<int, 2>{}, t + NBUF, bo);
            rdb(I0{}, bnext, 0, 0);
        }
        SB(); mm(I1{}, 1, 1); SB();
        if (more) {
            if (load) issue_part(std::integral_constant<int, 3>{}, t + NBUF, bo);
            rdb(I0{}, bnext, 0, 1);
        }
        colsum_add(I1{});
        SB();
        STAMP();
    };
    for (int t = t_beg; t < t_end; t += NBUF) {
        iter(I0{}, t);
        if constexpr (NBUF >= 2) { if (t + 1 < t_end) iter(std::integral_constant<int, 1 % NBUF>{}, t + 1); }
        if constexpr (NBUF >= 3) { if (t + 2 < t_end) iter(std::integral_constant<int, 2 % NBUF>{}, t + 2); }
        if constexpr (NBUF >= 4) { if (t + 3 < t_end) iter(std::integral_constant<int, 3 % NBUF>{}, t + 3); }
    }

    // ---- epilogue: accumulators straight to memory.  Element v of lane l of a 32x32 block: row 8 (v >> 2) + 4 (l >> 5) +
    // (v & 3), column l & 31: one store instruction writes two 128-B runs.  Slice ksg writes its own slab.
    float* __restrict__ C = d.Cf + (size_t)ksg * M * d.ldcf;
    const int colb = n0 + wn * 64 + (lane & 31);
    const int rowb = m0 + wm * 64 + 4 * (lane >> 5);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = colb + 32 * nb;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = rowb + 32 * mb + 8 * (v >> 2) + (v & 3);
                if (row < M && col < N)
                    *reinterpret_cast<PSVAE_GLOBAL float*>(reinterpret_cast<uintptr_t>(C + (size_t)row * d.ldcf + col)) = acc[mb][nb][v];
            }
        }
    if (do_colsum) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const float s = csum[mb] + __shfl_xor(csum[mb], 32);
            const int row = m0 + wm * 64 + 32 * mb + (lane & 31);
            if (lane < 32 && row < M) d.colsum[(size_t)ksg * M + row] = s;
        }
    }
#ifdef PSVAE_STAMP
    STAMP();
    if ((tid & 63) == 0 && (blockIdx.x == 203 || blockIdx.x == 40)) {
        for (int k = 1; k < nst_; ++k) st_[k - 1] = st_[k] - st_[k - 1];
        printf("wgrad stamp wg %d wave %d M=%d N=%d nbuf=%d: pro %llu | %llu %llu %llu %llu | %llu %llu %llu %llu | %llu %llu %llu %llu | %llu %llu %llu %llu | %llu %llu %llu %llu | ... last %llu epi %llu (n=%d)\n",
               (int)blockIdx.x, wave, M, N, NBUF, st_[0], st_[1], st_[2], st_[3], st_[4], st_[5], st_[6], st_[7], st_[8], st_[9], st_[10],
               st_[11], st_[12], st_[13], st_[14], st_[15], st_[16], st_[17], st_[18], st_[19], st_[20], st_[nst_ - 3], st_[nst_ - 2], nst_);
    }
#endif
#undef SB
#undef STAMP
}

// ---- loader-wavefront form ---------------------------------------------------------------------------------------------
// (PSVAE_WGRAD_LW=1; not the default -- see launch_wgrad_batch_bf16 for what it measured.)
// A fifth wavefront does nothing but issue the LDS-DMA of the tiles; the four MFMA wavefronts never issue a vector-memory
// instruction.  In the form above a wavefront that reaches its DMA instructions waits in the vector-memory issue queue
// behind the other wavefronts' (a CU takes in ~50 B/clk: the 32 KB of a tile are ~650 cycles of issue), and its MFMAs and
// fragment reads wait behind them in program order: the hand-over step of a tile measured 750-900 cycles with the DMA
// and 300 without.  Same LDS image, same fragment reads, same barrier per tile (five wavefronts take part in it).
__device__ __forceinline__ void glds16_s(unsigned voff, const u16* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

template <int NBUF>
__device__ __forceinline__ void wgrad_tile_lw(const GemmDescBf& d, int tm, int tn, int ksg, unsigned lds0) {
    static_assert(NBUF == 2 || NBUF == 3, "the loader counts at most one tile (32 instructions) behind the awaited one");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..3: MFMA wavefronts (2 x 2 blocks of 64 x 64), 4: loader
    const int M = d.M, N = d.N, K = d.K;
    const int m0 = tm * WG_BM, n0 = tn * WG_BN;
    const int nt_all = (K + WG_BK - 1) / WG_BK;
    const int S = d.ksplit > 1 ? d.ksplit : 1;
    const int per = (nt_all + S - 1) / S;
    const int t_beg = ksg * per;
    const int t_end = (t_beg + per < nt_all ? t_beg + per : nt_all);
    if (t_beg >= t_end && wave == 4) return;  // (an empty slice: the MFMA wavefronts store zeros, no barrier is executed)

    if (wave == 4) {
        // ===== loader: instruction J (0..15) of an operand = k step J >> 2, half (J >> 1) & 1, feature pair J & 1 (see the
        // image in the file header); lane -> (block of the pair, k half, k row, 16-feature half, 8-feature piece)
        const int cbl = lane >> 5, kh = (lane >> 4) & 1, qq = (lane >> 2) & 3, c16 = (lane >> 1) & 1, ph = lane & 1;
        unsigned va[2], vb[2];  // per-lane byte offsets inside a tile, by feature pair
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
            const int col = 64 * pair + 32 * cbl + 16 * c16 + 8 * ph;
            const int ca = (m0 + col + 8 <= d.lda) ? m0 + col : m0;  // (pieces beyond the pitch: see wgrad_tile)
            const int cb = (n0 + col + 8 <= d.ldb) ? n0 + col : n0;
            va[pair] = (unsigned)(((8 * kh + qq) * d.lda + ca) * 2);
            vb[pair] = (unsigned)(((8 * kh + qq) * d.ldb + cb) * 2);
        }
        const u16* ta = d.A + (size_t)t_beg * WG_BK * d.lda;  // tile bases (wave-uniform), advanced per issued tile
        const u16* tb = d.B + (size_t)t_beg * WG_BK * d.ldb;
        auto issue = [&](int t, unsigned buf_off) {
            if ((t + 1) * WG_BK <= K) {
#pragma unroll
                for (int J = 0; J < 16; ++J) {
                    const int rows = 16 * (J >> 2) + 4 * ((J >> 1) & 1);
                    glds16_s(va[J & 1], ta + (size_t)rows * d.lda, __builtin_amdgcn_readfirstlane(lds0 + buf_off + 1024u * J));
                }
#pragma unroll
                for (int J = 0; J < 16; ++J) {
                    const int rows = 16 * (J >> 2) + 4 * ((J >> 1) & 1);
                    glds16_s(vb[J & 1], tb + (size_t)rows * d.ldb, __builtin_amdgcn_readfirstlane(lds0 + buf_off + OP_BYTES + 1024u * J));
                }
            } else {  // ragged tile (the last one of the batch): the same image through registers, rows past the batch as zeros
                const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
                for (int J = 0; J < 16; ++J) {
                    const int rows = 16 * (J >> 2) + 4 * ((J >> 1) & 1);
                    const bool in = t * WG_BK + rows + 8 * kh + qq < K;
                    const char* ga = reinterpret_cast<const char*>(ta + (size_t)rows * d.lda) + va[J & 1];
                    const char* gb = reinterpret_cast<const char*>(tb + (size_t)rows * d.ldb) + vb[J & 1];
                    const bf16x8 xa = in ? *reinterpret_cast<const PSVAE_GLOBAL bf16x8*>(reinterpret_cast<uintptr_t>(ga)) : zero8;
                    const bf16x8 xb = in ? *reinterpret_cast<const PSVAE_GLOBAL bf16x8*>(reinterpret_cast<uintptr_t>(gb)) : zero8;
                    *(PSVAE_LDS bf16x8*)(uintptr_t)(lds0 + buf_off + 1024u * J + 16u * lane) = xa;
                    *(PSVAE_LDS bf16x8*)(uintptr_t)(lds0 + buf_off + OP_BYTES + 1024u * J + 16u * lane) = xb;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            ta += (size_t)WG_BK * d.lda;
            tb += (size_t)WG_BK * d.ldb;
        };
        int issued = 0;
#pragma unroll
        for (int b = 0; b < NBUF; ++b)
            if (t_beg + b < t_end) { issue(t_beg + b, (unsigned)b * TILE_BYTES); ++issued; }
        // the first tile has landed when at most the tiles behind it are in flight (32 instructions each; NBUF = 3: the
        // counter holds 63, so "two tiles behind" is waited down to one)
        if (issued >= 2) wait_vm<32>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        for (int t = t_beg; t + 1 < t_end; ++t) {
            const int after = t_end - 2 - t;
            if (NBUF >= 3 && after >= 1) wait_vm<32>(); else wait_vm<0>();  // tile t + 1 has landed
            __builtin_amdgcn_s_barrier();                                        // ... and everybody is done with tile t
            if (t + NBUF < t_end) issue(t + NBUF, (unsigned)((t - t_beg) % NBUF) * TILE_BYTES);
        }
        return;
    }

    // ===== MFMA wavefronts
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, i = lane & 15;
    const unsigned lanepart = 16u * (16u * (g >> 1) + 4u * (i >> 2) + 2u * (g & 1) + ((i & 3) >> 1)) + 8u * (i & 1);
    const unsigned aoff = lds0 + lanepart + 1024u * wm;
    const unsigned boff = lds0 + lanepart + OP_BYTES + 1024u * wn;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    float csum[2] = {0.f, 0.f};
    const bool do_colsum = (d.colsum != nullptr) && (tn == 0) && (wn == 0);
    bf16x8 fa[2][2], fb[2][2];  // [set][block]
    auto rda = [&](auto SET, unsigned buf_off, int ks, int mb) {
        constexpr int s = decltype(SET)::value;
        const v4s lo = trd(aoff + buf_off + 4096u * ks + 512u * mb), hi = trd(aoff + buf_off + 4096u * ks + 2048u + 512u * mb);
        fa[s][mb] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto rdb = [&](auto SET, unsigned buf_off, int ks, int nb) {
        constexpr int s = decltype(SET)::value;
        const v4s lo = trd(boff + buf_off + 4096u * ks + 512u * nb), hi = trd(boff + buf_off + 4096u * ks + 2048u + 512u * nb);
        fb[s][nb] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto mm = [&](auto SET, int mb, int nb) {
        constexpr int s = decltype(SET)::value;
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][mb], fb[s][nb], acc[mb][nb], 0, 0, 0);
    };
    auto colsum_add = [&](auto SET) {
        constexpr int s = decltype(SET)::value;
        if (do_colsum) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int e = 0; e < 8; ++e) csum[mb] += bf2f_(fa[s][mb][e]);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
#define SB() __builtin_amdgcn_sched_barrier(0)
#ifdef PSVAE_STAMP
    unsigned long long st_[64];
    int nst_ = 0;
#define STAMP() do { if (nst_ < 64) st_[nst_++] = __builtin_amdgcn_s_memtime(); } while (0)
    STAMP();
#else
#define STAMP() do {} while (0)
#endif
    // one 16-deep k step: MFMAs of set CUR in the order (0,0) (1,0) (0,1) (1,1), the next step's reads between them in the
    // order A0 B0 A1 B1 -- the order in which the next step's MFMAs first need them
    auto kstep = [&](auto CUR, auto NXT, unsigned buf_off, int ksn) {
        mm(CUR, 0, 0); SB(); rda(NXT, buf_off, ksn, 0); SB();
        mm(CUR, 1, 0); SB(); rdb(NXT, buf_off, ksn, 0); SB();
        mm(CUR, 0, 1); SB(); rda(NXT, buf_off, ksn, 1); SB();
        mm(CUR, 1, 1); SB(); rdb(NXT, buf_off, ksn, 1); colsum_add(CUR); SB();
    };
    if (t_beg < t_end) {
        __builtin_amdgcn_s_barrier();  // the loader's first tile
        SB();
        rda(I0{}, 0u, 0, 0); rdb(I0{}, 0u, 0, 0); rda(I0{}, 0u, 0, 1); rdb(I0{}, 0u, 0, 1);
        SB();
    }
    STAMP();
    auto iter = [&](auto BUFC, int t) {
        constexpr int BUF = decltype(BUFC)::value;
        constexpr unsigned bo = (unsigned)BUF * TILE_BYTES, bnext = (unsigned)((BUF + 1) % NBUF) * TILE_BYTES;
        kstep(I0{}, I1{}, bo, 1);
        kstep(I1{}, I0{}, bo, 2);
        kstep(I0{}, I1{}, bo, 3);
        STAMP();
        // last k step = hand-over: behind its first MFMA every read of tile t has returned (lgkmcnt 0); after the barrier the
        // loader may overwrite this buffer and tile t + 1 is there
        mm(I1{}, 0, 0); SB();
        if (t + 1 < t_end) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            STAMP();
            __builtin_amdgcn_s_barrier();
            STAMP();
            rda(I0{}, bnext, 0, 0); SB();
            mm(I1{}, 1, 0); SB(); rdb(I0{}, bnext, 0, 0); SB();
            mm(I1{}, 0, 1); SB(); rda(I0{}, bnext, 0, 1); SB();
            mm(I1{}, 1, 1); SB(); rdb(I0{}, bnext, 0, 1); SB();
        } else {
            mm(I1{}, 1, 0); mm(I1{}, 0, 1); mm(I1{}, 1, 1);
        }
        colsum_add(I1{});
        SB();
        STAMP();
    };
    for (int t = t_beg; t < t_end; t += NBUF) {
        iter(I0{}, t);
        if (t + 1 < t_end) iter(std::integral_constant<int, 1 % NBUF>{}, t + 1);
        if constexpr (NBUF >= 3) { if (t + 2 < t_end) iter(std::integral_constant<int, 2 % NBUF>{}, t + 2); }
    }
    // ---- epilogue (as wgrad_tile)
    float* __restrict__ C = d.Cf + (size_t)ksg * M * d.ldcf;
    const int colb = n0 + wn * 64 + (lane & 31);
    const int rowb = m0 + wm * 64 + 4 * (lane >> 5);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = colb + 32 * nb;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = rowb + 32 * mb + 8 * (v >> 2) + (v & 3);
                if (row < M && col < N)
                    *reinterpret_cast<PSVAE_GLOBAL float*>(reinterpret_cast<uintptr_t>(C + (size_t)row * d.ldcf + col)) = acc[mb][nb][v];
            }
        }
    if (do_colsum) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const float s = csum[mb] + __shfl_xor(csum[mb], 32);
            const int row = m0 + wm * 64 + 32 * mb + (lane & 31);
            if (lane < 32 && row < M) d.colsum[(size_t)ksg * M + row] = s;
        }
    }
#ifdef PSVAE_STAMP
    STAMP();
    if ((tid & 63) == 0 && wave == 0 && (blockIdx.x == 203 || blockIdx.x == 40)) {
        for (int k = 1; k < nst_; ++k) st_[k - 1] = st_[k] - st_[k - 1];
        printf("wgrad-lw stamp wg %d M=%d N=%d nbuf=%d: pro %llu | %llu %llu %llu %llu | %llu %llu %llu %llu | %llu %llu %llu %llu | %llu %llu %llu %llu | ... last %llu epi %llu (n=%d)\n",
               (int)blockIdx.x, M, N, NBUF, st_[0], st_[1], st_[2], st_[3], st_[4], st_[5], st_[6], st_[7], st_[8], st_[9], st_[10],
               st_[11], st_[12], st_[13], st_[14], st_[15], st_[16], st_[nst_ - 3], st_[nst_ - 2], nst_);
    }
#endif
#undef SB
#undef STAMP
}

}  // namespace

template <int NBUF>
__global__ __launch_bounds__(320) void wgrad_bf16_lw_kernel(const GemmBatchBf batch) {
    extern __shared__ __attribute__((aligned(16))) char wg_smem[];
    int L = blockIdx.x;
    if (batch.xcd_remap) {
        const int T = batch.total_tiles, q8 = T >> 3, r8 = T & 7;
        const int xcd = L & 7, idx = L >> 3;
        L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int P = batch.tiles_per_slice;
    const int ksg = L / P;
    const int tile = L - ksg * P;
    int di = 0;
    for (int i = 1; i < batch.n; ++i)
        if (tile >= batch.d[i].tile_begin) di = i;
    const GemmDescBf& d = batch.d[di];
    const int lt = tile - d.tile_begin;
    const int tm = lt / d.tiles_n, tn = lt - tm * d.tiles_n;
    wgrad_tile_lw<NBUF>(d, tm, tn, ksg, (unsigned)reinterpret_cast<uintptr_t>(wg_smem));
}

namespace {

template <int NBUF>
hipError_t launch_wgrad_lw(GemmBatchBf& b, hipStream_t stream) {
    constexpr int lds = NBUF * TILE_BYTES;
    static std::atomic<unsigned long long> lds_ok{0};
    if (hipError_t e = lds_opt_in(lds_ok, reinterpret_cast<const void*>(&wgrad_bf16_lw_kernel<NBUF>), lds > 64 * 1024 ? lds : 64 * 1024))
        return e;
    hipLaunchKernelGGL((wgrad_bf16_lw_kernel<NBUF>), dim3(b.total_tiles), dim3(320), lds, stream, b);
    return hipGetLastError();
}

}  // namespace

// Work list of a launch: (batch slice, problem, tile row, tile column), slice-major; every problem has the same batch
// and the same split, so every item costs the same.  Workgroups are dealt round-robin over the 8 XCDs, each with its own
// L2: XCD x takes a CONTIGUOUS eighth of the list, so the 16 tiles of one slice of a 512 x 512 gradient (4 + 4 operand
// panels) run under one L2, and so do problems that read the same array (the table gradients next to the layer whose dY
// they share; both first encoder layers, which share x).  With a contiguous range per PROBLEM (round 2's mapping) the
// launch pulled 90 MB through the L2s for 53 MB of operands, and the fabric, not the CU, set its time.
template <int NBUF, int BK>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const GemmBatchBf batch) {
    extern __shared__ __attribute__((aligned(16))) char wg_smem[];
    int L = blockIdx.x;
    if (batch.xcd_remap) {
        const int T = batch.total_tiles, q8 = T >> 3, r8 = T & 7;
        const int xcd = L & 7, idx = L >> 3;
        L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;  // bijective for any T
    }
    const int P = batch.tiles_per_slice;
    const int ksg = L / P;
    const int tile = L - ksg * P;
    int di = 0;
    for (int i = 1; i < batch.n; ++i)
        if (tile >= batch.d[i].tile_begin) di = i;
    const GemmDescBf& d = batch.d[di];
    const int lt = tile - d.tile_begin;
    const int tm = lt / d.tiles_n, tn = lt - tm * d.tiles_n;
    wgrad_tile<NBUF, BK>(d, tm, tn, ksg, (unsigned)reinterpret_cast<uintptr_t>(wg_smem));
}

namespace {

template <int NBUF, int BK>
hipError_t launch_wgrad_nbuf(GemmBatchBf& b, hipStream_t stream) {
    constexpr int lds = NBUF * 2 * BK * 256;
    static std::atomic<unsigned long long> lds_ok{0};  // per device: psvae_common.h lds_opt_in
    if (hipError_t e = lds_opt_in(lds_ok, reinterpret_cast<const void*>(&wgrad_bf16_kernel<NBUF, BK>), lds > 64 * 1024 ? lds : 64 * 1024))
        return e;
    hipLaunchKernelGGL((wgrad_bf16_kernel<NBUF, BK>), dim3(b.total_tiles), dim3(256), lds, stream, b);
    return hipGetLastError();
}

}  // namespace

// every problem of the batch: tr = 1 (A = dY [K, lda], B = X [K, ldb], pitches % 8 == 0, bases 16-B aligned), fp32 output
// (slabs when ksplit > 1), optional column sums; no bias / mask / table epilogue
hipError_t launch_wgrad_batch_bf16(GemmBatchBf& b, hipStream_t stream) {
    if (b.n <= 0) return hipSuccess;
    int t = 0;
    const int S = b.d[0].ksplit > 1 ? b.d[0].ksplit : 1;
    for (int i = 0; i < b.n; ++i) {
        GemmDescBf& d = b.d[i];
        if (!d.tr || !d.Cf || d.Cb || d.bias || d.ctab || d.epi != 0) return hipErrorInvalidValue;
        if ((d.lda & 7) || (d.ldb & 7) || (reinterpret_cast<uintptr_t>(d.A) & 15) || (reinterpret_cast<uintptr_t>(d.B) & 15))
            return hipErrorInvalidValue;
        if ((d.ksplit > 1 ? d.ksplit : 1) != S || d.K != b.d[0].K) return hipErrorInvalidValue;  // one batch, one split
        d.tiles_m = (d.M + WG_BM - 1) / WG_BM;
        d.tiles_n = (d.N + WG_BN - 1) / WG_BN;
        d.tile_begin = t;  // (within one slice)
        t += d.tiles_m * d.tiles_n;
    }
    b.tiles_per_slice = t;
    t *= S;
    b.total_tiles = t;
    {
        const char* e = getenv("PSVAE_BF16_XCD");  // tuning (read per call: tests flip it in-process)
        b.xcd_remap = e ? atoi(e) : 1;
    }
    const char* nbe = getenv("PSVAE_WGRAD_NBUF");  // (read per call: a test runs the other depths)
    const int nbuf = nbe ? atoi(nbe) : 2;
    // loader-wavefront form: measured slower at cfg3 (29.0 vs 22.0 us): its hand-over step is 360 cycles instead of 750-900, but
    // five wavefronts of 144 registers do not pack two workgroups onto a CU's four SIMDs reliably (3 wavefronts per SIMD),
    // and with two LDS buffers one wavefront's 32 DMA instructions (>= 500 cycles of issue) plus their latency exceed the
    // tile they must hide under (barrier wait 700-800 cycles per tile); with three buffers it is one workgroup per CU.
    // Kept selectable (read per call: a test runs it) as the measured alternative.
    const char* lwe = getenv("PSVAE_WGRAD_LW");
    if (lwe && atoi(lwe)) return nbuf == 3 ? launch_wgrad_lw<3>(b, stream) : launch_wgrad_lw<2>(b, stream);
    // batch rows per tile: 32 (default; two buffers = 32 KB per workgroup) or 64.  Measured at cfg3, per step, interleaved runs:
    // 32 rows x 2 buffers 95.1-95.6 us, 32 x 3 96.1-96.3, 64 x 2 96.4-96.9, 64 x 3 / 64 x 4 (one workgroup per CU) +10 / +12
    const char* bke = getenv("PSVAE_WGRAD_BK");
    if (bke && atoi(bke) == 64) {
        if (nbuf == 3) return launch_wgrad_nbuf<3, 64>(b, stream);
        if (nbuf == 4) return launch_wgrad_nbuf<4, 64>(b, stream);
        return launch_wgrad_nbuf<2, 64>(b, stream);
    }
    if (nbuf == 3) return launch_wgrad_nbuf<3, 32>(b, stream);
    if (nbuf == 4) return launch_wgrad_nbuf<4, 32>(b, stream);
    return launch_wgrad_nbuf<2, 32>(b, stream);
}

}  // namespace psvae
