"""The N > 1 path on CPU: two processes, gloo backend (the GPU build uses the same code with RCCL).

(1) GradAverager: bucketed asynchronous all-reduce gives the mean in every bucket, decoder first.
(2) Data-parallel semantics: the mean over ranks of per-shard gradients (each rank's ELBO is a mean
    over ITS rows) equals the full-batch gradient -- the oracle plays both roles, so this pins the
    collective the engine must issue (mean, not sum) and the row sharding."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import psvae_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from pseudo_speaker_vae_amd.ddp import GradAverager, broadcast_flat
        from pseudo_speaker_vae_amd.engine import FlatLayout
        D, Z, B = 64, 16, 24
        shapes = [torch.Size(s) for s in O.param_shapes(D, Z).values()]
        lay = FlatLayout(shapes)
        # (1) plain averaging of rank-dependent data, bucket by bucket
        flat = torch.full((lay.total,), float(rank + 1))
        av = GradAverager(flat, [0, lay.enc_end, lay.total])
        av.reduce_bucket(1)
        av.reduce_bucket(0)
        av.wait()
        ok1 = bool(torch.allclose(flat, torch.full_like(flat, (1 + world) / 2)))
        # the default path: ONE collective over the whole flat buffer
        flat1 = torch.full((lay.total,), float(10 * (rank + 1)))
        av1 = GradAverager(flat1, [0, lay.enc_end, lay.total])
        av1.reduce_all()
        av1.wait()
        ok1 = ok1 and bool(torch.allclose(flat1, torch.full_like(flat1, 10 * (1 + world) / 2)))
        # parameter broadcast from rank 0
        pflat = torch.full((8,), float(rank))
        broadcast_flat(pflat, 0)
        ok1 = ok1 and bool((pflat == 0).all())
        # (2) shard rows, per-shard oracle step, average through the averager
        p = O.init_params(D, Z, 0)
        g = torch.Generator().manual_seed(7)
        x, eps = torch.randn(B, D, generator=g), torch.randn(B, Z, generator=g)
        rows = slice(rank * B // world, (rank + 1) * B // world)
        losses, grads, _, _ = O.train_step(p, x[rows], eps[rows], 0.7)
        gflat = torch.zeros(lay.total)
        for v, k in zip(lay.views(gflat), O.PARAM_KEYS):
            v.copy_(grads[k])
        gflat[lay.loss_offset:lay.loss_offset + 3] = torch.stack([losses["total"], losses["recon"], losses["kl"]])
        av2 = GradAverager(gflat, [0, lay.enc_end, lay.total])
        av2.reduce_bucket(1)
        av2.reduce_bucket(0)
        av2.wait()
        full_l, full_g, _, _ = O.train_step(p, x, eps, 0.7)
        err = 0.0
        for v, k in zip(lay.views(gflat), O.PARAM_KEYS):
            err = max(err, float((v - full_g[k]).abs().max() / full_g[k].abs().max()))
        lerr = abs(float(gflat[lay.loss_offset]) - float(full_l["total"]))
        q.put((rank, ok1, err, lerr))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gradient_averaging():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok1, err, lerr in res:
        assert ok1, f"rank {rank}: bucket averaging wrong"
        assert err < 1e-5, f"rank {rank}: mean of shard grads != full-batch grad ({err:.2e})"
        assert lerr < 1e-6, f"rank {rank}: averaged loss scalars wrong"
