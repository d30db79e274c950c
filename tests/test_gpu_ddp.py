"""The gradient exchange (DDPStrategy, /root/reference/ps_vae/training.py:78) on the GPU.

(1) TrainEngine with a 1-rank RCCL group on the box's single GPU: every data-parallel mode -- eager phase split with
    two bucketed all-reduces, captured step + one all-reduce, two captured halves around the collectives, collective(s)
    captured inside the step's graph -- must leave gradients, loss scalars and the Philox state bit-identical to the
    engine without a group (ncclAvg over one rank is the identity; what is under test is that every bucket is reduced
    after the kernels that write it, and that nothing is skipped or reordered).
(2) TWO ranks sharing the one GPU over gloo (RCCL refuses two ranks on one device): each rank's TrainEngine takes half
    of the rows; after the exchange both hold the same averaged gradients, equal to one engine on the full batch
    (mean-of-shards identity) with the same epsilon rows (world-size-invariant Philox interleave).
"""
import os
import socket

import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D, Z, B = 256, 64, 512


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def nccl_group():
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    yield dist.group.WORLD
    dist.barrier()
    dist.destroy_process_group()


def _same(e_ref, e, what):
    torch.cuda.synchronize()
    assert torch.equal(e_ref.flat_grads, e.flat_grads), what + ": gradients / loss scalars differ"
    assert torch.equal(e_ref.rng_state, e.rng_state), what + ": Philox state differs"
    assert torch.equal(e_ref.ws.acts["eps"], e.ws.acts["eps"]), what + ": epsilon differs"


@pytest.mark.parametrize("mode", ["eager_split", "stream_overlapped", "graph_then_allreduce", "two_graphs", "in_graph", "in_graph_overlap"])
def test_one_rank_rccl_engine_modes_match_the_engine_without_a_group(nccl_group, mode):
    from pseudo_speaker_vae_amd.engine import TrainEngine
    x = torch.randn(B, D, device=DEV)
    ref = TrainEngine(D, Z, B, device=DEV, seed=7, beta=0.9)
    eng = TrainEngine(D, Z, B, device=DEV, seed=7, beta=0.9, process_group=nccl_group)
    assert eng.world == 1 and eng.averager is not None and eng.averager.backend == "nccl"
    if mode == "eager_split":
        step = lambda: eng.forward_backward(x)
    elif mode == "stream_overlapped":  # the decoder bucket's all-reduce on a side stream under the encoder backward
        step = lambda: eng.step_overlapped(x)
    elif mode == "graph_then_allreduce":
        eng.capture(x)
        step = eng.replay
    elif mode == "two_graphs":
        eng.capture_ddp(x)
        step = eng.replay_ddp
    else:
        eng.capture_with_collective(x, overlap=(mode == "in_graph_overlap"))
        step = eng.replay
    # (the captures' warm-up steps leave no trace: the first replay is step 1, same Philox position as `ref`)
    for it in range(3):
        ref.forward_backward(x)
        step()
        _same(ref, eng, f"{mode} step {it}")
    assert float(eng.loss3[0]) == float(ref.loss3[0]) and torch.isfinite(eng.flat_grads).all()


@pytest.mark.parametrize("mode", ["eager", "graph_then_allreduce"])
def test_one_rank_rccl_bf16_engine_matches_the_engine_without_a_group(nccl_group, mode, monkeypatch):
    """the bf16 pass (row chains, 6 launches with the weight packs kept by the optimizer step) under a process group: same gradients, loss scalars and Philox state as
    without one, eager and captured"""
    from pseudo_speaker_vae_amd.engine import TrainEngine
    monkeypatch.delenv("PSVAE_BF16_PACK_IN_STEP", raising=False)
    x = torch.randn(B, D, device=DEV)
    kw = dict(device=DEV, seed=9, beta=0.6, dtype="bf16", conditional=True)
    ref = TrainEngine(D, Z, B, **kw)
    eng = TrainEngine(D, Z, B, process_group=nccl_group, **kw)
    g = torch.Generator().manual_seed(3)
    gender, age = torch.randint(-1, 3, (B,), generator=g), torch.randint(-1, 3, (B,), generator=g)
    ref.set_labels(gender, age)
    eng.set_labels(gender, age)
    if mode == "eager":
        step = lambda: eng.forward_backward(x)
    else:
        eng.capture(x)
        step = eng.replay
    for it in range(3):
        ref.forward_backward(x)
        step()
        torch.cuda.synchronize()
        assert torch.equal(ref.flat_grads, eng.flat_grads), f"{mode} step {it}: gradients / loss scalars differ"
        assert torch.equal(ref.rng_state, eng.rng_state), f"{mode} step {it}: Philox state differs"
        assert torch.equal(ref.ws.eps, eng.ws.eps), f"{mode} step {it}: epsilon differs"
    assert eng.launches_per_step() == 6 and torch.isfinite(eng.flat_grads).all()


def test_one_rank_rccl_optimizer_after_the_exchange(nccl_group):
    """the whole data-parallel iteration: captured step, all-reduce, fused Adam -- same parameters as without a group"""
    from pseudo_speaker_vae_amd.engine import TrainEngine
    x = torch.randn(B, D, device=DEV)
    ref = TrainEngine(D, Z, B, device=DEV, seed=2)
    eng = TrainEngine(D, Z, B, device=DEV, seed=2, process_group=nccl_group)
    eng.capture(x)
    ref.capture(x)
    for _ in range(3):
        ref.replay(); ref.optimizer_step()
        eng.replay(); eng.optimizer_step()
    torch.cuda.synchronize()
    assert torch.equal(ref.flat_params, eng.flat_params) and torch.equal(ref.flat_params_t, eng.flat_params_t)


def _two_rank_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pseudo_speaker_vae_amd.engine import TrainEngine
        torch.cuda.set_device(0)
        g = torch.Generator().manual_seed(11)
        x = torch.randn(B, D, generator=g)
        rows = slice(rank * B // world, (rank + 1) * B // world)
        xs = x[rows].to(DEV)
        eng = TrainEngine(D, Z, B // world, device=DEV, seed=5, beta=0.8, process_group=dist.group.WORLD)
        out = {}
        eng.forward_backward(xs)                       # eager: phase split + two bucketed all-reduces
        torch.cuda.synchronize()
        out["eager_grads"] = eng.flat_grads.cpu().clone()
        out["eager_eps"] = eng.ws.acts["eps"].cpu().clone()
        eng.capture(xs)                                # (warm-up steps are local and leave no trace) + capture
        eng.replay()                                   # captured step, then ONE all-reduce of the flat buffer
        torch.cuda.synchronize()
        out["graph_grads"] = eng.flat_grads.cpu().clone()
        out["graph_eps"] = eng.ws.acts["eps"].cpu().clone()
        eng.optimizer_step()
        torch.cuda.synchronize()
        out["params"] = eng.flat_params.cpu().clone()
        q.put((rank, {k: v.numpy() for k, v in out.items()}))  # by value: the parent may read after this process is gone
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_over_gloo_average_like_one_full_batch_engine():
    import torch.multiprocessing as mp
    from pseudo_speaker_vae_amd.engine import TrainEngine
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res = {r: {k: torch.from_numpy(v) for k, v in d.items()} for r, d in res.items()}
    # both ranks hold the same averaged gradients and, after Adam, the same parameters
    for key in ("eager_grads", "graph_grads", "params"):
        assert torch.equal(res[0][key], res[1][key]), key
    # ... equal to ONE engine on the full batch at the same Philox positions
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, D, generator=g).to(DEV)
    full = TrainEngine(D, Z, B, device=DEV, seed=5, beta=0.8)
    full.forward_backward(x)
    torch.cuda.synchronize()
    eps_full = full.ws.acts["eps"].cpu()
    assert torch.equal(torch.cat([res[0]["eager_eps"], res[1]["eager_eps"]]), eps_full)  # W-invariant epsilon stream
    lay = full.layout
    for k, (o, s) in zip(full.keys, zip(lay.offsets, lay.shapes)):
        n = s.numel()
        assert_close(res[0]["eager_grads"][o:o + n], full.flat_grads[o:o + n].cpu(), 2e-5, "mean of shard grads " + k)
    lo = lay.loss_offset
    assert_close(res[0]["eager_grads"][lo:lo + 3], full.flat_grads[lo:lo + 3].cpu(), 2e-6, "averaged loss scalars")
    full.forward_backward(x)    # the ranks' replayed step = full-batch step 2 (capturing consumed no epsilon)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([res[0]["graph_eps"], res[1]["graph_eps"]]), full.ws.acts["eps"].cpu())
    for k, (o, s) in zip(full.keys, zip(lay.offsets, lay.shapes)):
        n = s.numel()
        assert_close(res[0]["graph_grads"][o:o + n], full.flat_grads[o:o + n].cpu(), 2e-5, "captured: shard mean " + k)


# ---- the direct (IPC) gradient exchange: one-shot reduce-scatter + all-gather, stream-ordered interprocess events --------
def _direct_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pseudo_speaker_vae_amd.ddp import DirectAllReduce
        from pseudo_speaker_vae_amd.engine import TrainEngine
        torch.cuda.set_device(0)
        out = {}
        # (1) the collective alone: ragged payload (not a multiple of world * 4), several steps back to back with NO host
        #     synchronisation in between -- the stream-ordered events must keep the steps apart
        n = 4 * 250_001
        flat = torch.empty(n, device=DEV)
        ar = DirectAllReduce(flat, dist.group.WORLD)
        g = torch.Generator().manual_seed(100 + rank)
        vals = [torch.randn(n, generator=g) for _ in range(4)]
        dvals = [v.to(DEV) for v in vals]
        res = []
        for t in range(4):
            flat.copy_(dvals[t])
            ar.allreduce()
            res.append(flat.clone())          # (stream-ordered: the clone is enqueued behind the exchange)
        torch.cuda.synchronize()
        out["coll"] = torch.stack([r.cpu() for r in res])
        out["mine"] = torch.stack(vals)
        ar.close()
        if B % world == 0:
            # (2) the training engine with the direct exchange in place of the backend all-reduce
            gx = torch.Generator().manual_seed(11)
            x = torch.randn(B, D, generator=gx)
            rows = slice(rank * B // world, (rank + 1) * B // world)
            eng = TrainEngine(D, Z, B // world, device=DEV, seed=5, beta=0.8, process_group=dist.group.WORLD,
                              direct_allreduce=True)
            assert eng.averager.direct is not None
            eng.forward_backward(x[rows].to(DEV))
            torch.cuda.synchronize()
            out["eager_grads"] = eng.flat_grads.cpu().clone()
            eng.capture(x[rows].to(DEV))
            eng.replay()
            eng.optimizer_step()
            torch.cuda.synchronize()
            out["graph_grads"] = eng.flat_grads.cpu().clone()
            out["params"] = eng.flat_params.cpu().clone()
            eng.averager.direct.close()
        q.put((rank, {k: v.numpy() for k, v in out.items()}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_direct_ipc_allreduce_between_processes_sharing_the_gpu(world):
    """csrc/ipc_allreduce.hip with `world` processes on the one GPU: every rank maps every other rank's buffers
    (hipIpcMemHandle), the phases are ordered by interprocess events on the streams (no spin-wait kernel); results equal
    the mean computed on the host, all ranks bit-identical, four exchanges in flight back to back; then the training
    engine over it equals one full-batch engine (as the gloo test above does for the backend all-reduce)."""
    import torch.multiprocessing as mp
    from pseudo_speaker_vae_amd.engine import TrainEngine
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_direct_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res = {r: {k: torch.from_numpy(v) for k, v in d.items()} for r, d in res.items()}
    mean = sum(res[r]["mine"].double() for r in range(world)) / world
    for r in range(world):
        assert torch.equal(res[r]["coll"], res[0]["coll"]), f"rank {r} differs from rank 0"
        for key in ("eager_grads", "graph_grads", "params"):
            if key in res[0]:
                assert torch.equal(res[r][key], res[0][key]), (r, key)
    assert_close(res[0]["coll"], mean.float(), 2e-6, "direct all-reduce vs host mean", elementwise=False)
    if B % world == 0:
        g = torch.Generator().manual_seed(11)
        x = torch.randn(B, D, generator=g).to(DEV)
        full = TrainEngine(D, Z, B, device=DEV, seed=5, beta=0.8)
        full.forward_backward(x)
        torch.cuda.synchronize()
        lay = full.layout
        for k, (o, s) in zip(full.keys, zip(lay.offsets, lay.shapes)):
            n = s.numel()
            assert_close(res[0]["eager_grads"][o:o + n], full.flat_grads[o:o + n].cpu(), 2e-5, "direct: mean of shard grads " + k)
