// Direct gradient exchange for the data-parallel step (the reference's DDPStrategy, /root/reference/ps_vae/training.py:78:
// g <- mean over ranks of g, 5.13 MB of fp32 per step at cfg2) over xGMI WITHOUT a ring: one-shot reduce-scatter + all-gather
// through buffers every rank maps from every other rank (hipIpcMemHandle), two hops in all, each a kernel whose remote reads
// use all of a GPU's links at once (rank r pulls 1/W of the payload from each of its W-1 peers).
//
//   publish     rank r copies its flat gradients into its exported staging buffer      stage_r[0 .. n)
//   reduce      rank r sums chunk r of every rank's staging buffer in rank order       red_r[chunk r] = (1/W) sum_p stage_p[chunk r]
//               (fixed order: every rank ends with bit-identical gradients) and keeps it as its own result too
//   gather      rank r reads every other chunk from the rank that reduced it           g[chunk p] = red_p[chunk p]
//
// Synchronisation is stream-ordered end to end -- NO kernel ever spins on a flag: after `publish` and after `reduce` a rank
// records an interprocess event (hipIpcEventHandle) on its stream, and the consumers' streams wait for it
// (hipStreamWaitEvent).  What an event wait cannot express by itself is WHICH record it refers to (a wait enqueued before
// the peer's host has called hipEventRecord for this step would bind to last step's record), so the hosts hand each other
// one sequence number per phase through a page of POSIX shared memory: a host enqueues its wait for (peer, phase, step t)
// only after the peer's host has enqueued that record.  That is a host-side handshake between processes that are about to
// block on each other's GPUs anyway; the GPUs never poll.
//   Reuse hazards: stage_r is rewritten by publish(t+1), after r's own gather(t), which waited for every peer's reduce(t) --
// the last reader of stage_r(t).  red_r is rewritten by reduce(t+1), which waits for every peer's publish(t+1), enqueued on
// the peer's stream behind its gather(t) -- the last reader of red_r(t).
#include <fcntl.h>
#include <sched.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <new>

#include "psvae_common.h"

namespace psvae {

constexpr int IPC_MAX_WORLD = 16;

struct IpcBlob {  // what a rank hands to every other rank at setup (psvae_ipc_blob_bytes() bytes)
    hipIpcMemHandle_t mem;
    hipIpcEventHandle_t ev[2];
    unsigned long long nfloats;
    int rank, world;
};

struct IpcShared {  // one page of POSIX shared memory per communicator: seq[rank][phase] = last step whose record is enqueued
    std::atomic<unsigned long long> seq[IPC_MAX_WORLD][2];
    std::atomic<int> attached;
};

struct IpcPtrs {
    const float* stage[IPC_MAX_WORLD];
    const float* red[IPC_MAX_WORLD];
};

}  // namespace psvae

struct psvae_ipc_comm {
    int rank, world, dev;
    size_t n, chunk;              // floats in the payload; floats per rank's chunk (a multiple of 4)
    float* base;                  // hipMalloc: [stage n | red n]
    void* peer_base[psvae::IPC_MAX_WORLD];
    hipEvent_t ev[2];             // mine: 0 = published, 1 = reduced
    hipEvent_t peer_ev[psvae::IPC_MAX_WORLD][2];
    psvae::IpcShared* shm;
    char shm_name[64];
    unsigned long long step;
    bool connected;
    psvae::IpcPtrs ptrs;
};

namespace psvae {

// red_mine[chunk] = g[chunk] = (1/W) sum_p stage_p[chunk], p in rank order (16-byte pieces; remote reads over xGMI)
__global__ __launch_bounds__(256) void ipc_reduce_kernel(IpcPtrs P, float* __restrict__ red_mine, float* __restrict__ g,
                                                         size_t off, size_t len4, int world, float inv) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < len4; i += (size_t)gridDim.x * 256) {
        f32x4 s = *reinterpret_cast<const f32x4*>(P.stage[0] + off + 4 * i);
        for (int p = 1; p < world; ++p) s += *reinterpret_cast<const f32x4*>(P.stage[p] + off + 4 * i);
        s *= inv;
        *reinterpret_cast<f32x4*>(red_mine + off + 4 * i) = s;
        *reinterpret_cast<f32x4*>(g + off + 4 * i) = s;
    }
}

// g[chunk p] = red_p[chunk p] for every p != rank; blockIdx.y walks the peers
__global__ __launch_bounds__(256) void ipc_gather_kernel(IpcPtrs P, float* __restrict__ g, size_t chunk, size_t n, int rank,
                                                         int world) {
    int p = blockIdx.y;
    if (p >= rank) ++p;  // skip myself
    if (p >= world) return;
    const size_t off = (size_t)p * chunk;
    const size_t len4 = (off >= n ? 0 : (n - off < chunk ? n - off : chunk)) / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < len4; i += (size_t)gridDim.x * 256)
        *reinterpret_cast<f32x4*>(g + off + 4 * i) = *reinterpret_cast<const f32x4*>(P.red[p] + off + 4 * i);
}

static bool wait_seq(std::atomic<unsigned long long>& s, unsigned long long want, double timeout_s) {
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0; s.load(std::memory_order_acquire) < want; ++spins) {
        if ((spins & 1023) == 1023) {
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > timeout_s) return false;
            sched_yield();
        }
    }
    return true;
}

}  // namespace psvae

using namespace psvae;

extern "C" {

size_t psvae_ipc_blob_bytes(void) { return sizeof(IpcBlob); }

// Step 1 of 2: allocate this rank's exchange buffers and events; `blob_out` (psvae_ipc_blob_bytes() bytes) is what the other
// ranks need -- the caller moves the blobs between the ranks (torch.distributed all_gather: setup only).  `tag` names the
// communicator's shared-memory page: the same string on every rank, unique per communicator (e.g. "<master port>-<id>").
int psvae_ipc_comm_create(int rank, int world, size_t nfloats, const char* tag, psvae_ipc_comm** out, void* blob_out) {
    if (!out || !blob_out || !tag) return PSVAE_ERR_NULL;
    if (world < 1 || world > IPC_MAX_WORLD || rank < 0 || rank >= world || nfloats == 0 || (nfloats & 3)) return PSVAE_ERR_SHAPE;
    psvae_ipc_comm* c = new (std::nothrow) psvae_ipc_comm();
    if (!c) return PSVAE_ERR_STATE;
    memset(c, 0, sizeof(*c));
    c->rank = rank; c->world = world; c->n = nfloats;
    c->chunk = ((nfloats + world - 1) / world + 3) / 4 * 4;
    hipError_t e = hipGetDevice(&c->dev);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->base), 2 * nfloats * sizeof(float));
    if (e == hipSuccess) e = hipMemset(c->base, 0, 2 * nfloats * sizeof(float));
    for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&c->ev[k], hipEventDisableTiming | hipEventInterprocess);
    IpcBlob b{};
    if (e == hipSuccess) e = hipIpcGetMemHandle(&b.mem, c->base);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipIpcGetEventHandle(&b.ev[k], c->ev[k]);
    if (e != hipSuccess) { delete c; return (int)e; }
    b.nfloats = nfloats; b.rank = rank; b.world = world;
    memcpy(blob_out, &b, sizeof(b));
    snprintf(c->shm_name, sizeof(c->shm_name), "/psvae_ipc_%.40s", tag);
    // rank 0 creates (and zeroes) the page, the others attach in psvae_ipc_comm_connect, after the blobs have gone round
    if (rank == 0) {
        shm_unlink(c->shm_name);
        const int fd = shm_open(c->shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(IpcShared)) != 0) { if (fd >= 0) close(fd); delete c; return PSVAE_ERR_STATE; }
        c->shm = reinterpret_cast<IpcShared*>(mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
        close(fd);
        if (c->shm == MAP_FAILED) { c->shm = nullptr; delete c; return PSVAE_ERR_STATE; }
        memset(reinterpret_cast<void*>(c->shm), 0, sizeof(IpcShared));
    }
    *out = c;
    return 0;
}

// Step 2 of 2: `all_blobs` = world blobs in rank order.  Maps every peer's buffers and events.
int psvae_ipc_comm_connect(psvae_ipc_comm* c, const void* all_blobs) {
    if (!c || !all_blobs) return PSVAE_ERR_NULL;
    const IpcBlob* B = reinterpret_cast<const IpcBlob*>(all_blobs);
    for (int p = 0; p < c->world; ++p) {
        if (B[p].rank != p || B[p].world != c->world || B[p].nfloats != c->n) return PSVAE_ERR_ARG;
        if (p == c->rank) {
            c->peer_base[p] = c->base;
            c->peer_ev[p][0] = c->ev[0]; c->peer_ev[p][1] = c->ev[1];
        } else {
            hipError_t e = hipIpcOpenMemHandle(&c->peer_base[p], B[p].mem, hipIpcMemLazyEnablePeerAccess);
            for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipIpcOpenEventHandle(&c->peer_ev[p][k], B[p].ev[k]);
            if (e != hipSuccess) return (int)e;
        }
        c->ptrs.stage[p] = reinterpret_cast<const float*>(c->peer_base[p]);
        c->ptrs.red[p] = reinterpret_cast<const float*>(c->peer_base[p]) + c->n;
    }
    if (c->rank != 0) {
        const int fd = shm_open(c->shm_name, O_RDWR, 0600);
        if (fd < 0) return PSVAE_ERR_STATE;
        c->shm = reinterpret_cast<IpcShared*>(mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
        close(fd);
        if (c->shm == MAP_FAILED) { c->shm = nullptr; return PSVAE_ERR_STATE; }
    }
    c->shm->attached.fetch_add(1);
    c->connected = true;
    return 0;
}

// g[0 .. n) <- mean over the ranks of g, in place, on `stream`; every rank must call it once per step, in step order.
// Returns PSVAE_ERR_STATE if a peer's host does not show up within `timeout_s` (its process died): never hangs the GPU.
int psvae_ipc_allreduce_avg_f32(psvae_ipc_comm* c, float* g, double timeout_s, void* stream) {
    if (!c || !g) return PSVAE_ERR_NULL;
    if (!c->connected) return PSVAE_ERR_STATE;
    if (reinterpret_cast<uintptr_t>(g) & 15) return PSVAE_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const unsigned long long t = ++c->step;
    const int W = c->world, r = c->rank;
    float* stage = c->base;
    float* red = c->base + c->n;
    // ---- publish ----
#define PSVAE_TRY(e) do { hipError_t _e = (e); if (_e != hipSuccess) return (int)_e; } while (0)
    PSVAE_TRY(hipMemcpyAsync(stage, g, c->n * sizeof(float), hipMemcpyDeviceToDevice, s));
    PSVAE_TRY(hipEventRecord(c->ev[0], s));
    c->shm->seq[r][0].store(t, std::memory_order_release);
    // ---- reduce my chunk: needs every rank's publish of this step ----
    for (int p = 0; p < W; ++p) {
        if (p == r) continue;
        if (!wait_seq(c->shm->seq[p][0], t, timeout_s)) return PSVAE_ERR_STATE;
        PSVAE_TRY(hipStreamWaitEvent(s, c->peer_ev[p][0], 0));
    }
    {
        const size_t off = (size_t)r * c->chunk;
        const size_t len4 = (off >= c->n ? 0 : (c->n - off < c->chunk ? c->n - off : c->chunk)) / 4;
        if (len4 > 0) {
            const int grid = (int)((len4 + 255) / 256 < 512 ? (len4 + 255) / 256 : 512);
            hipLaunchKernelGGL(ipc_reduce_kernel, dim3(grid), dim3(256), 0, s, c->ptrs, red, g, off, len4, W, 1.0f / (float)W);
            PSVAE_TRY(hipGetLastError());
        }
    }
    PSVAE_TRY(hipEventRecord(c->ev[1], s));
    c->shm->seq[r][1].store(t, std::memory_order_release);
    // ---- gather the other chunks: needs every rank's reduce of this step ----
    if (W > 1) {
        for (int p = 0; p < W; ++p) {
            if (p == r) continue;
            if (!wait_seq(c->shm->seq[p][1], t, timeout_s)) return PSVAE_ERR_STATE;
            PSVAE_TRY(hipStreamWaitEvent(s, c->peer_ev[p][1], 0));
        }
        const size_t len4 = c->chunk / 4;
        const int gx = (int)((len4 + 255) / 256 < 256 ? (len4 + 255) / 256 : 256);
        hipLaunchKernelGGL(ipc_gather_kernel, dim3(gx, W - 1), dim3(256), 0, s, c->ptrs, g, c->chunk, c->n, r, W);
        PSVAE_TRY(hipGetLastError());
    }
#undef PSVAE_TRY
    return 0;
}

int psvae_ipc_comm_destroy(psvae_ipc_comm* c) {
    if (!c) return 0;
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank || !c->connected) continue;
        if (c->peer_base[p]) (void)hipIpcCloseMemHandle(c->peer_base[p]);
        for (int k = 0; k < 2; ++k)
            if (c->peer_ev[p][k]) (void)hipEventDestroy(c->peer_ev[p][k]);
    }
    for (int k = 0; k < 2; ++k)
        if (c->ev[k]) (void)hipEventDestroy(c->ev[k]);
    if (c->base) (void)hipFree(c->base);
    if (c->shm) {
        munmap(reinterpret_cast<void*>(c->shm), sizeof(IpcShared));
        if (c->rank == 0) shm_unlink(c->shm_name);
    }
    delete c;
    return 0;
}

}  // extern "C"
