"""Size-independent properties of the HIP hot path at BASELINE.json's full sizes (the oracle is too slow to be
the checker there): gradient linearity in the loss scale, the data-parallel identity (mean of shard gradients ==
full-batch gradient), loss identities, Philox-stream continuity across batch splits, decode row
independence at 1 M rows."""
import pytest
import torch

from helpers import assert_close
from oracle import psvae_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _step(ops, params, x, eps, beta=1.0, go=None):
    B, D = x.shape
    Z = eps.shape[1]
    ws = ops.Workspace(B, D, Z, DEV)
    grads = [torch.full_like(t, float("nan")) for t in params]
    loss3 = torch.full((3,), float("nan"), device=DEV)
    outs = ws.new_outputs()
    ops.train_step(ws, x, params, grads, outs, loss3, beta=beta, eps=eps,
                   grad_loss=None if go is None else torch.tensor([go], device=DEV))
    torch.cuda.synchronize()
    return loss3.clone(), grads, outs, ws


@pytest.mark.parametrize("D,Z,B", [(256, 64, 1024), (512, 128, 1024), (256, 64, 4096)])
def test_gradients_are_linear_in_the_loss_scale(D, Z, B):
    from pseudo_speaker_vae_amd import ops
    p = O.init_params(D, Z, 2)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    g = torch.Generator().manual_seed(B + D)
    x, eps = torch.randn(B, D, generator=g).to(DEV), torch.randn(B, Z, generator=g).to(DEV)
    l1, g1, _, _ = _step(ops, params, x, eps, 0.7, go=1.0)
    l2, g2, _, _ = _step(ops, params, x, eps, 0.7, go=-2.5)
    assert torch.equal(l1, l2)  # the loss itself does not depend on the upstream scale
    assert abs(float(l1[0]) - (float(l1[1]) + 0.7 * float(l1[2]))) <= 1e-6 * max(1.0, abs(float(l1[0])))
    assert float(l1[2]) >= 0.0  # KL(q || N(0, I)) >= 0
    for k, a, b in zip(O.PARAM_KEYS, g1, g2):
        assert_close(b.cpu(), (-2.5 * a).cpu(), 2e-6, "linearity " + k)


@pytest.mark.parametrize("D,Z,B", [(256, 64, 1024), (512, 128, 2048)])
def test_mean_of_shard_gradients_is_the_full_batch_gradient(D, Z, B):
    """what the DDP all-reduce relies on (training.py:78): with the batch cut in W shards, the mean over shards of
    the per-shard gradients (each normalised by its own B/W) equals the full-batch gradient"""
    from pseudo_speaker_vae_amd import ops
    p = O.init_params(D, Z, 4)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    g = torch.Generator().manual_seed(B)
    x, eps = torch.randn(B, D, generator=g).to(DEV), torch.randn(B, Z, generator=g).to(DEV)
    lf, gf, _, _ = _step(ops, params, x, eps)
    W = 4
    acc = [torch.zeros_like(t) for t in params]
    ls = torch.zeros(3, device=DEV)
    for r in range(W):
        sl = slice(r * B // W, (r + 1) * B // W)
        l, gs, _, _ = _step(ops, params, x[sl].contiguous(), eps[sl].contiguous())
        ls += l / W
        for a, t in zip(acc, gs):
            a += t / W
    assert_close(ls.cpu(), lf.cpu(), 2e-6, "loss")
    for k, a, b in zip(O.PARAM_KEYS, acc, gf):
        assert_close(a.cpu(), b.cpu(), 2e-5, "shard mean " + k)


def test_philox_stream_is_continuous_across_batch_splits():
    """the device-resident Philox state advances by exactly the blocks a batch consumed: two half batches draw
    the same epsilon as one full batch that starts at the same state"""
    from pseudo_speaker_vae_amd import ops
    D, Z, B = 256, 64, 1024
    p = O.init_params(D, Z, 0)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    x = torch.randn(B, D).to(DEV)

    def run(xs, state):
        ws = ops.Workspace(xs.shape[0], D, Z, DEV)
        grads = [torch.empty_like(t) for t in params]
        loss3 = torch.empty(3, device=DEV)
        ops.train_step(ws, xs, params, grads, ws.new_outputs(), loss3, rng_state=state)
        torch.cuda.synchronize()
        return ws.acts["eps"].clone()

    s1 = torch.tensor([77, 1000], dtype=torch.int64, device=DEV)
    full = run(x, s1)
    assert int(s1[1]) == 1000 + B * Z // 4
    s2 = torch.tensor([77, 1000], dtype=torch.int64, device=DEV)
    a = run(x[: B // 2].contiguous(), s2)
    b = run(x[B // 2:].contiguous(), s2)
    assert torch.equal(torch.cat([a, b]), full)
    ref = O.philox_normal(B * Z, 77, 1000).reshape(B, Z)
    assert_close(full.cpu(), ref, 1e-5, "eps vs oracle Philox")


def test_decode_rows_are_independent_at_one_million_rows():
    """cfg5: every output row depends on its own latent row only -- decoding 1,000,000 rows in one call equals
    decoding any slice of them alone (chunk boundaries at 65,536 rows are invisible)"""
    from pseudo_speaker_vae_amd import ops
    D, Z, N = 256, 64, 1_000_000
    p = O.init_params(D, Z, 9)
    dec = [p[k].to(DEV) for k in O.PARAM_KEYS[12:]]
    z = ops.philox_normal(N * Z, 5, 0, DEV).reshape(N, Z)
    full = ops.decode(z, dec, D)
    torch.cuda.synchronize()
    assert torch.isfinite(full).all()
    for lo, hi in ((0, 100), (65_530, 65_545), (499_999, 500_131), (N - 37, N)):
        part = ops.decode(z[lo:hi].contiguous(), dec, D)
        assert_close(part.cpu(), full[lo:hi].cpu(), 2e-6, f"rows {lo}:{hi}")
    ref = O.decode(p, z[:64].cpu())
    assert_close(full[:64].cpu(), ref, 1e-4, "first rows vs oracle")


@pytest.mark.parametrize("graph", [False, True])
def test_data_parallel_epsilon_stream_is_world_size_invariant(graph):
    """ranks share the seed and interleave their Philox counters by global row index, so W ranks of
    B/W rows draw, step after step, exactly the epsilon one rank with B rows draws (eager and captured)"""
    from pseudo_speaker_vae_amd.engine import TrainEngine
    D, Z, B, W = 256, 64, 1024, 4
    x = torch.randn(B, D, device=DEV)
    full = TrainEngine(D, Z, B, device=DEV, seed=11)
    shards = [TrainEngine(D, Z, B // W, device=DEV, seed=11, shard=(r, W)) for r in range(W)]
    xs = [x[r * B // W:(r + 1) * B // W].contiguous() for r in range(W)]
    if graph:
        full.capture(x)
        for e, xr in zip(shards, xs):
            e.capture(xr)
    seen = []
    for step in range(3):
        if graph:
            full.replay()
            [e.replay() for e in shards]
        else:
            full.forward_backward(x)
            [e.forward_backward(xr) for e, xr in zip(shards, xs)]
        torch.cuda.synchronize()
        eps_full = full.ws.acts["eps"].clone()
        eps_sh = torch.cat([e.ws.acts["eps"] for e in shards])
        assert torch.equal(eps_sh, eps_full), f"step {step}"
        seen.append(eps_full)
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])  # fresh noise every step


def test_langevin_rows_are_independent_at_one_million_rows():
    """cfg5, conditional leg (inference.py:72-105) at BASELINE size: 1,000,000 latent rows x 100 Langevin steps in one
    launch.  Every row's chain depends on its own z0 and its own noise blocks only: any slice run alone -- with the
    Philox offsets of its rows -- reproduces the big run bit for bit across workgroup boundaries; and a slice's whole
    history matches the oracle's loop fed the same noise."""
    from pseudo_speaker_vae_amd import ops, sampling
    from pseudo_speaker_vae_amd.latent_classifier import LatentClassifier
    N, Z, T, target, step, nw = 1_000_000, 64, 100, 1, 0.05, 0.7
    torch.manual_seed(0)
    clf = LatentClassifier(Z, 3, num_layers=2, hidden_dim=128).to(DEV)
    with torch.no_grad():
        for q in clf.parameters():
            q.mul_(2.0)
    fus = sampling._fusable_classifier(clf, target)
    z0 = ops.philox_normal(N * Z, 123, 0, DEV).view(N, Z)
    nblk = N * Z // 4
    zT, _ = sampling.langevin_fused(fus, z0, target, step, nw, T, seed=321, offset=0)
    torch.cuda.synchronize()
    assert torch.isfinite(zT).all()
    # (a) slices alone, with injected noise = the big run's Philox blocks of those rows
    for lo, hi in ((0, 64), (255, 321), (499_968, 500_100), (N - 50, N)):
        n = hi - lo
        noises = torch.stack([ops.philox_normal(N * Z, 321, t * nblk, DEV, fast=True).view(N, Z)[lo:hi] for t in range(3)])
        part3, _ = sampling.langevin_fused(fus, z0[lo:hi].contiguous(), target, step, nw, 3, noises)
        big3, _ = sampling.langevin_fused(fus, z0, target, step, nw, 3, seed=321, offset=0)
        assert torch.equal(part3, big3[lo:hi]), f"rows {lo}:{hi} depend on their neighbours"
    # (b) history parity with the oracle on a slice of the full 100-step run
    lo, hi = 777_700, 777_732
    noises = torch.stack([ops.philox_normal(N * Z, 321, t * nblk, DEV, fast=True).view(N, Z)[lo:hi] for t in range(T)])
    zs, hist = sampling.langevin_fused(fus, z0[lo:hi].contiguous(), target, step, nw, T, noises, return_history=True)
    assert torch.equal(zs, zT[lo:hi])
    layers = [(fus[0].cpu(), fus[1].cpu()), (fus[2].cpu(), fus[3].cpu())]
    zr, href = O.langevin(layers, z0[lo:hi].cpu(), noises.cpu(), target, step, nw)
    assert_close(hist.cpu(), torch.stack(href), 1e-4, "100-step history vs oracle")
