"""Parity of the HIP hot path (through the C-ABI) with the CPU oracle and with the golden fixtures
generated from the reference.  Tolerance: 1e-4 relative fp32 (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from helpers import (GOLDEN, MASK_KEYS, REL_TOL, assert_close, check_grads_against_fixture, check_masks,
                     fixture_params, load, step_fixtures)
from oracle import psvae_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from pseudo_speaker_vae_amd import ops as o
    return o


def run_step(ops, p, x, eps, beta, normalize, cos, go=None, seed=None):
    from pseudo_speaker_vae_amd import _ffi
    B, D = x.shape
    Z = p["encoder_mu.4.bias"].shape[0]
    flags = (_ffi.NORMALIZE_DECODER if normalize else 0) | (_ffi.COS_LOSS if cos else 0)
    ws = ops.Workspace(B, D, Z, DEV, normalize)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    xd = x.to(DEV)
    xh, mu, lv = ops.vae_forward(ws, xd, params, eps.to(DEV) if eps is not None else None, flags=flags,
                                 seed=seed or 0)
    loss3 = ops.elbo(ws, xd, xh, mu, lv, beta, flags).clone()
    grads = [torch.full_like(t, float("nan")) for t in params]
    god = torch.tensor([go], device=DEV) if go is not None else None
    ops.vae_backward(ws, xd, params, xh, mu, lv, grads, elbo_grad=True, beta=beta, grad_loss=god, flags=flags)
    torch.cuda.synchronize()
    masks = {k: (ws.acts[k] > 0).cpu() for k in MASK_KEYS}
    return dict(x_hat=xh.cpu(), mu=mu.cpu(), logvar=lv.cpu(), z=ws.acts["z"].cpu(), eps=ws.acts["eps"].cpu(),
                loss3=loss3.cpu(), grads={k: g.cpu() for k, g in zip(O.PARAM_KEYS, grads)}, ws=ws, masks=masks)


@pytest.mark.parametrize("path", step_fixtures(), ids=lambda p: os.path.basename(p)[:-4])
def test_train_step_matches_reference_golden(ops, path):
    """HIP forward + ELBO + backward vs the reference's own outputs (fixtures made by make_golden.py)."""
    fx = load(path)
    p = fixture_params(fx)
    out = run_step(ops, p, torch.from_numpy(fx["x"]), torch.from_numpy(fx["eps"]), float(fx["beta"]),
                   bool(fx["normalize"]), bool(fx["cos"]))
    for k in ("x_hat", "mu", "logvar", "z"):
        assert_close(out[k], fx[k], REL_TOL, k)
    for i, k in enumerate(("total", "recon", "kl")):
        assert abs(float(out["loss3"][i]) - float(fx[k])) <= REL_TOL * max(1.0, abs(float(fx[k]))), k
    check_grads_against_fixture(out["grads"], fx, REL_TOL)


@pytest.mark.parametrize("D,Z,B", [(256, 32, 64), (256, 64, 1024), (512, 128, 1024), (256, 64, 100), (96, 24, 11)])
@pytest.mark.parametrize("variant", ["plain", "norm", "cos", "norm_cos"])
def test_train_step_matches_oracle(ops, D, Z, B, variant):
    """BASELINE configs 1, 2, 4 (per-GPU shape) + ragged shapes, all loss variants, go != 1."""
    if variant != "plain" and B == 1024 and D == 512:
        pytest.skip("variant coverage at the smaller sizes is enough")
    normalize, cos = "norm" in variant, "cos" in variant
    p = O.init_params(D, Z, 2)
    torch.manual_seed(B + D)
    x, eps = torch.randn(B, D), torch.randn(B, Z)
    go = 0.37
    out = run_step(ops, p, x, eps, 0.8, normalize, cos, go=go)
    # gradients are compared at equal ReLU masks (see oracle.backward): the masks themselves are checked
    losses, grads, acts, _ = O.train_step(p, x, eps, 0.8, normalize, cos, go=go, masks=out["masks"])
    check_masks(out["masks"], acts)
    for k in ("x_hat", "mu", "logvar", "z"):
        assert_close(out[k], acts[k], REL_TOL, k)
    for i, k in enumerate(("total", "recon", "kl")):
        assert abs(float(out["loss3"][i]) - float(losses[k])) <= REL_TOL * max(1.0, abs(float(losses[k]))), k
    for k in O.PARAM_KEYS:
        assert_close(out["grads"][k], grads[k], REL_TOL, "grad " + k)


def test_generic_backward_upstream_grads_and_gx(ops):
    """backward with arbitrary upstream gradients of (x_hat, mu, logvar) (the autograd path) and d/dx."""
    D, Z, B = 256, 64, 48
    p = O.init_params(D, Z, 4)
    torch.manual_seed(9)
    x, eps = torch.randn(B, D), torch.randn(B, Z)
    g_xh, g_mu, g_lv = torch.randn(B, D), torch.randn(B, Z), torch.randn(B, Z)
    ws = ops.Workspace(B, D, Z, DEV)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    xd = x.to(DEV)
    xh, mu, lv = ops.vae_forward(ws, xd, params, eps.to(DEV))
    grads = [torch.empty_like(t) for t in params]
    g_x = torch.empty_like(xd)
    ops.vae_backward(ws, xd, params, xh, mu, lv, grads, g_xhat=g_xh.to(DEV), g_mu=g_mu.to(DEV),
                     g_logvar=g_lv.to(DEV), g_x=g_x)
    torch.cuda.synchronize()
    acts = O.forward(p, x, eps)
    ref, extra = O.backward(p, x, acts, g_xh, g_mu, g_lv, need_dx=True)
    for k, g in zip(O.PARAM_KEYS, grads):
        assert_close(g.cpu(), ref[k], REL_TOL, "grad " + k)
    assert_close(g_x.cpu(), extra["g_x"], REL_TOL, "g_x")


def test_backward_phases_equal_single_call(ops):
    from pseudo_speaker_vae_amd import _ffi
    D, Z, B = 256, 64, 64
    p = O.init_params(D, Z, 1)
    torch.manual_seed(1)
    x, eps = torch.randn(B, D).to(DEV), torch.randn(B, Z).to(DEV)
    ws = ops.Workspace(B, D, Z, DEV)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    xh, mu, lv = ops.vae_forward(ws, x, params, eps)
    g1 = [torch.zeros_like(t) for t in params]
    g2 = [torch.zeros_like(t) for t in params]
    ops.vae_backward(ws, x, params, xh, mu, lv, g1, elbo_grad=True)
    ops.vae_backward(ws, x, params, xh, mu, lv, g2, elbo_grad=True, phases=_ffi.PHASE_DECODER)
    ops.vae_backward(ws, x, params, xh, mu, lv, g2, elbo_grad=True, phases=_ffi.PHASE_ENCODER)
    torch.cuda.synchronize()
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)


def test_step_is_bitwise_reproducible(ops):
    """fixed-order reductions, no float atomics: two runs agree bit for bit."""
    D, Z, B = 256, 64, 1024
    p = O.init_params(D, Z, 0)
    torch.manual_seed(0)
    x = torch.randn(B, D)
    a = run_step(ops, p, x, None, 1.0, False, False, seed=5)
    b = run_step(ops, p, x, None, 1.0, False, False, seed=5)
    assert torch.equal(a["eps"], b["eps"]) and torch.equal(a["loss3"], b["loss3"])
    for k in O.PARAM_KEYS:
        assert torch.equal(a["grads"][k], b["grads"][k]), k
    c = run_step(ops, p, x, None, 1.0, False, False, seed=6)
    assert not torch.equal(a["eps"], c["eps"])
    # the internally drawn eps is the oracle's Philox stream
    want = O.philox_normal(B * Z, 5, 0).reshape(B, Z)
    assert np.abs(a["eps"].numpy() - want).max() < 2e-5


@pytest.mark.parametrize("name", ["decode_d256_z64_n33", "decode_d256_z64_n33_norm"])
def test_decode_matches_reference_golden(ops, name):
    from pseudo_speaker_vae_amd import _ffi
    fx = load(os.path.join(GOLDEN, name + ".npz"))
    D, Z = int(fx["D"]), int(fx["Z"])
    p = O.init_params(D, Z, int(fx["pseed"]))
    dec = [p[k].to(DEV) for k in O.PARAM_KEYS[12:]]
    xh = ops.decode(torch.from_numpy(fx["z"]).to(DEV), dec, D,
                    _ffi.NORMALIZE_DECODER if int(fx["normalize"]) else 0)
    torch.cuda.synchronize()
    assert_close(xh.cpu(), fx["x_hat"], REL_TOL, "decode")


def test_decode_chunked_large(ops):
    """more rows than one internal chunk (65536): rows are independent, so any slice must match a direct
    decode of those rows (to rounding: a different batch size may pick a different k-split, i.e. a
    different fp32 summation order)."""
    D, Z, N = 256, 64, 70000
    p = O.init_params(D, Z, 0)
    dec = [p[k].to(DEV) for k in O.PARAM_KEYS[12:]]
    z = ops.philox_normal(N * Z, 3, 0, DEV).view(N, Z)
    full = ops.decode(z, dec, D)
    part = ops.decode(z[65500:65600].contiguous(), dec, D)
    torch.cuda.synchronize()
    assert_close(full[65500:65600].cpu(), part.cpu(), 2e-6, "chunk seam rows")
    ref = O.decode(p, z[-64:].cpu())
    assert_close(full[-64:].cpu(), ref, REL_TOL, "decode tail")


def test_cond_forward_backward_matches_oracle(ops):
    """cfg3 extension in fp32 (parity unpinned by the reference; checked against the oracle's definition)."""
    from pseudo_speaker_vae_amd import _ffi
    D, Z, B = 256, 64, 96
    torch.manual_seed(3)
    p = O.init_params(D, Z, 1)
    tabs = {k: torch.randn(O.COND_DIM, O.HIDDEN) * 0.05 for k in O.COND_KEYS}
    x, eps = torch.randn(B, D), torch.randn(B, Z)
    gender = torch.randint(-1, 3, (B,), dtype=torch.int32)
    age = torch.randint(-1, 3, (B,), dtype=torch.int32)
    acts = O.forward_cond(p, tabs, x, eps, gender, age)
    g_xh, g_mu, g_lv = O.elbo_grads(acts["x_hat"], x, acts["mu"], acts["logvar"], 1.0)
    ref, tref, _ = O.backward_cond(p, tabs, x, acts, g_xh, g_mu, g_lv, gender, age)

    ws = ops.Workspace(B, D, Z, DEV)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    tabd = [tabs[k].to(DEV) for k in O.COND_KEYS]
    gtab = [torch.empty_like(t) for t in tabd]
    gd, ad = gender.to(DEV), age.to(DEV)  # keep alive: the struct holds raw pointers
    cond = _ffi.make_cond(tabd, gtab, gd, ad)
    xd = x.to(DEV)
    xh, mu, lv = ops.vae_forward(ws, xd, params, eps.to(DEV), cond=cond)
    grads = [torch.empty_like(t) for t in params]
    ops.vae_backward(ws, xd, params, xh, mu, lv, grads, elbo_grad=True, beta=1.0, cond=cond)
    torch.cuda.synchronize()
    assert_close(xh.cpu(), acts["x_hat"], REL_TOL, "x_hat")
    assert_close(mu.cpu(), acts["mu"], REL_TOL, "mu")
    for k, g in zip(O.PARAM_KEYS, grads):
        assert_close(g.cpu(), ref[k], REL_TOL, "grad " + k)
    for k, g in zip(O.COND_KEYS, gtab):
        assert_close(g.cpu(), tref[k], REL_TOL, "grad " + k)


@pytest.mark.parametrize("D,Z,B", [(256, 64, 1024), (256, 32, 64), (512, 128, 1024), (256, 64, 37), (784, 20, 32)])
def test_fused_train_step_matches_unfused_path_and_oracle(ops, D, Z, B):
    """psvae_train_step_f32 (elementwise work folded into GEMM epilogues) == forward + elbo + backward."""
    p = O.init_params(D, Z, 5)
    torch.manual_seed(D + B)
    x, eps = torch.randn(B, D), torch.randn(B, Z)
    go = 1.7
    ref = run_step(ops, p, x, eps, 0.9, False, False, go=go)
    ws = ops.Workspace(B, D, Z, DEV)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    grads = [torch.full_like(t, float("nan")) for t in params]
    outs = ws.new_outputs()
    loss3 = torch.full((3,), float("nan"), device=DEV)
    ops.train_step(ws, x.to(DEV), params, grads, outs, loss3, beta=0.9, eps=eps.to(DEV),
                   grad_loss=torch.tensor([go], device=DEV))
    torch.cuda.synchronize()
    # (the fused step sums the heads' products and decoder.0's dgrad as 8 partial products in fixed order: the same
    # numbers to fp32 rounding, not bit for bit)
    assert_close(outs[0].cpu(), ref["x_hat"], 5e-6, "x_hat")
    assert_close(loss3.cpu(), ref["loss3"], 5e-6, "loss3")
    masks = {k: (ws.acts[k] > 0).cpu() for k in MASK_KEYS}
    # A ReLU-kink flip between the two paths (an activation within rounding of 0) moves the gradient of the flipped
    # layer and of everything upstream of it by O(1/B): compare every parameter whose gradient no flipped mask feeds
    # into, bound the flips as check_masks does, and make a partial comparison VISIBLE instead of skipping.
    flips = {k: int((masks[k] != ref["masks"][k]).sum()) for k in MASK_KEYS}
    for k, n in flips.items():
        assert n <= max(2, 2e-5 * masks[k].numel()), f"{k}: {n} ReLU masks differ between the fused and the unfused path"
    feeds = {"decoder.4": (), "decoder.2": ("d2",), "decoder.0": ("d2", "d1"),
             "encoder_mu.4": ("d2", "d1"), "encoder_mu.2": ("d2", "d1", "h2m"), "encoder_mu.0": ("d2", "d1", "h2m", "h1m"),
             "encoder_sigma.4": ("d2", "d1"), "encoder_sigma.2": ("d2", "d1", "h2s"),
             "encoder_sigma.0": ("d2", "d1", "h2s", "h1s")}
    compared = 0
    for k, g in zip(O.PARAM_KEYS, grads):
        if not any(flips[m] for m in feeds[k.rsplit(".", 1)[0]]):
            assert_close(g.cpu(), ref["grads"][k], 2e-5, "fused vs unfused grad " + k)
            compared += 1
    assert compared >= 2  # decoder.4's weight and bias gradients depend on no mask
    if compared < len(O.PARAM_KEYS):
        import warnings
        warnings.warn(f"fused vs unfused (D={D}, Z={Z}, B={B}): ReLU-kink flips {flips}; {compared} of "
                      f"{len(O.PARAM_KEYS)} gradient tensors compared at 2e-5, the rest against the oracle only")
    losses, ograds, oacts, _ = O.train_step(p, x, eps, 0.9, go=go, masks=masks)
    check_masks(masks, oacts)
    for i, k in enumerate(("total", "recon", "kl")):
        assert abs(float(loss3[i]) - float(losses[k])) <= REL_TOL * max(1.0, abs(float(losses[k]))), k
    for k, g in zip(O.PARAM_KEYS, grads):
        assert_close(g.cpu(), ograds[k], REL_TOL, "fused vs oracle grad " + k)


@pytest.mark.parametrize("D,Z,B", [(256, 64, 1000), (96, 24, 37), (200, 48, 130)])
def test_wgrad_cross_workgroup_split_matches_oracle(ops, D, Z, B):
    """Weight gradients with the batch cut over S workgroups per tile (partials in slabs, summed in fixed
    order) agree with the unsplit kernel and the oracle for every S, on ragged shapes too."""
    from pseudo_speaker_vae_amd import _ffi
    p = O.init_params(D, Z, 11)
    torch.manual_seed(B)
    x, eps = torch.randn(B, D), torch.randn(B, Z)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    results = {}
    try:
        for S in (1, 2, 3, 4):
            assert _ffi.lib().psvae_debug_force_wgrad_split(S) == 0
            ws = ops.Workspace(B, D, Z, DEV)
            grads = [torch.full_like(t, float("nan")) for t in params]
            loss3 = torch.full((3,), float("nan"), device=DEV)
            ops.train_step(ws, x.to(DEV), params, grads, ws.new_outputs(), loss3, beta=1.3, eps=eps.to(DEV))
            torch.cuda.synchronize()
            results[S] = ([g.cpu() for g in grads], {k: (ws.acts[k] > 0).cpu() for k in MASK_KEYS})
    finally:
        _ffi.lib().psvae_debug_force_wgrad_split(0)
    assert _ffi.lib().psvae_debug_force_wgrad_split(5) == -4
    _, ograds, oacts, _ = O.train_step(p, x, eps, 1.3, masks=results[1][1])
    check_masks(results[1][1], oacts)
    for S, (grads, _) in results.items():
        for k, g, g1 in zip(O.PARAM_KEYS, grads, results[1][0]):
            assert_close(g, g1, 5e-6, f"S={S} vs S=1 grad {k}")
            assert_close(g, ograds[k], REL_TOL, f"S={S} vs oracle grad {k}")


def test_autotuned_launch_plan_keeps_results(ops):
    """psvae_tune_train_step_f32 only changes HOW the step is launched: after tuning the step still matches the
    oracle, and the plan has one entry per GEMM launch."""
    D, Z, B = 128, 32, 512
    p = O.init_params(D, Z, 3)
    torch.manual_seed(7)
    x, eps = torch.randn(B, D), torch.randn(B, Z)
    ws = ops.Workspace(B, D, Z, DEV)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    grads = [torch.full_like(t, float("nan")) for t in params]
    outs = ws.new_outputs()
    loss3 = torch.full((3,), float("nan"), device=DEV)
    assert ops.describe_plan(ws) == "untuned"
    plan = ops.tune_train_step(ws, x.to(DEV), params, grads, outs, loss3, iters=3)
    assert plan == ops.describe_plan(ws)
    sites = plan.split()
    assert [s.split(":")[0] for s in sites] == ["F1", "F2", "F3", "F5", "F6", "F7", "B1", "B2", "B3", "B4", "B5", "B6"]
    assert all(len(s.split(":")[1].split("/")) == 5 for s in sites)
    for t in grads:
        t.fill_(float("nan"))
    ops.train_step(ws, x.to(DEV), params, grads, outs, loss3, beta=0.7, eps=eps.to(DEV))
    torch.cuda.synchronize()
    masks = {k: (ws.acts[k] > 0).cpu() for k in MASK_KEYS}
    losses, ograds, oacts, _ = O.train_step(p, x, eps, 0.7, masks=masks)
    check_masks(masks, oacts)
    for i, k in enumerate(("total", "recon", "kl")):
        assert abs(float(loss3[i]) - float(losses[k])) <= REL_TOL * max(1.0, abs(float(losses[k]))), k
    for k, g in zip(O.PARAM_KEYS, grads):
        assert_close(g.cpu(), ograds[k], REL_TOL, "tuned step vs oracle grad " + k)


def test_plan_cache_file_roundtrip(ops, tmp_path):
    """tune_or_load_plan writes the tuned plan to a JSON cache and a second workspace of the same shape loads it
    instead of tuning again"""
    import json
    D, Z, B = 64, 16, 256
    p = O.init_params(D, Z, 3)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    x = torch.randn(B, D, device=DEV)
    cache = str(tmp_path / "plans.json")
    ws = ops.Workspace(B, D, Z, DEV)
    grads = [torch.empty_like(t) for t in params]
    loss3 = torch.empty(3, device=DEV)
    plan = ops.tune_or_load_plan(ws, x, params, grads, ws.new_outputs(), loss3, iters=2, cache_path=cache)
    stored = json.load(open(cache))
    assert list(stored.values()) == [plan] and f"{B}x{D}x{Z}" in list(stored.keys())[0]
    ops.set_plan(ws, "untuned")
    assert ops.describe_plan(ws) == "untuned"
    again = ops.tune_or_load_plan(ws, x, params, grads, ws.new_outputs(), loss3, iters=2, cache_path=cache)
    assert again == plan == ops.describe_plan(ws)
    ops.set_plan(ws, "untuned")


@pytest.mark.parametrize("split", [1, 4])
def test_fused_train_step_phases_equal_single_call(ops, split):
    """psvae_train_step_f32 in two calls (PHASE_DECODER then PHASE_ENCODER: what the overlapped data-parallel step
    does around its first all-reduce) gives bitwise the same outputs, loss and gradients as one PHASE_ALL call --
    also when weight gradients are split over workgroups (the decoder-only call must finish its own slabs)."""
    from pseudo_speaker_vae_amd import _ffi
    D, Z, B = 256, 64, 1024
    p = O.init_params(D, Z, 8)
    torch.manual_seed(3)
    x, eps = torch.randn(B, D).to(DEV), torch.randn(B, Z).to(DEV)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    assert _ffi.lib().psvae_debug_force_wgrad_split(split) == 0
    try:
        res = []
        for phases in ((_ffi.PHASE_ALL,), (_ffi.PHASE_DECODER, _ffi.PHASE_ENCODER)):
            ws = ops.Workspace(B, D, Z, DEV)
            grads = [torch.full_like(t, float("nan")) for t in params]
            outs = ws.new_outputs()
            loss3 = torch.full((3,), float("nan"), device=DEV)
            for i, ph in enumerate(phases):
                ops.train_step(ws, x, params, grads, outs, loss3, beta=0.9, eps=eps, phases=ph)
                if len(phases) == 2 and i == 0:  # after the decoder phase: loss and decoder gradients are complete
                    torch.cuda.synchronize()
                    assert torch.isfinite(loss3).all() and all(torch.isfinite(g).all() for g in grads[12:])
            torch.cuda.synchronize()
            res.append((loss3.clone(), [g.clone() for g in grads], [o.clone() for o in outs]))
    finally:
        _ffi.lib().psvae_debug_force_wgrad_split(0)
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        assert torch.equal(a, b)
