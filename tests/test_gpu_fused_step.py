"""The fused training step (psvae_train_step_f32, ABI 2): launch count, the second-stage fusion (mu / logvar heads and
decoder.0's dgrad as partial products on their neighbours' launches), optimizer-maintained k-major weights, the loss
variants and the one-hot conditioning inside the fused step, the device-state Adam and the cosine schedule -- all
against the CPU oracle at 1e-4 (north_star), gradients at equal ReLU masks with the masks themselves checked."""
import math

import pytest
import torch

from helpers import MASK_KEYS, REL_TOL, assert_close, check_masks, masks_of
from oracle import psvae_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _kmajor(params):
    """k-major copies of an 18-slot parameter list (weights transposed; bias slots just alias)"""
    return [t.t().contiguous() if t.dim() == 2 else t for t in params]


def _fused(ops, p, x, eps, beta, flags=0, kmajor=True, go=None, phases=None, cond=None):
    from pseudo_speaker_vae_amd import _ffi
    B, D = x.shape
    Z = p["encoder_mu.4.bias"].shape[0]
    ws = ops.Workspace(B, D, Z, DEV, bool(flags & _ffi.NORMALIZE_DECODER))
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    pk = _kmajor(params) if kmajor else None
    grads = [torch.full_like(t, float("nan")) for t in params]
    outs = ws.new_outputs()
    loss3 = torch.full((3,), float("nan"), device=DEV)
    god = torch.tensor([go], device=DEV) if go is not None else None
    n = 0
    for ph in (phases or (_ffi.PHASE_ALL,)):
        ops.train_step(ws, x.to(DEV), params, grads, outs, loss3, beta=beta, eps=eps.to(DEV), grad_loss=god, phases=ph,
                       params_kmajor=pk, flags=flags, cond=cond)
        n += ops.last_step_launches()
    torch.cuda.synchronize()
    return dict(ws=ws, grads=grads, outs=outs, loss3=loss3.cpu(), launches=n, masks=masks_of(ws.acts))


@pytest.fixture(scope="module")
def ops():
    from pseudo_speaker_vae_amd import ops as o
    return o


@pytest.mark.parametrize("D,Z,B,launches", [(256, 64, 1024, 10), (512, 128, 1024, 12), (256, 64, 37, 10),
                                            (256, 64, 4096, 12), (256, 32, 64, 12), (96, 24, 11, 13)])
def test_launch_count_and_parity_with_optimizer_maintained_kmajor_weights(ops, D, Z, B, launches):
    """cfg2 / cfg4 / cfg1 shapes: TEN launches at Z = 64, B <= 2048 (where the second-stage folds were measured to win),
    twelve otherwise (separate heads and decoder.0-dgrad launches), thirteen when Z is not a multiple of 16 (no
    bottleneck kernel): same numbers as the oracle everywhere"""
    p = O.init_params(D, Z, 5)
    torch.manual_seed(D + B)
    x, eps = torch.randn(B, D), torch.randn(B, Z)
    r = _fused(ops, p, x, eps, 0.9, go=1.7)
    assert r["launches"] == launches
    losses, ograds, oacts, _ = O.train_step(p, x, eps, 0.9, go=1.7, masks=r["masks"])
    check_masks(r["masks"], oacts)
    for k, t in zip(("x_hat", "mu", "logvar"), r["outs"]):
        assert_close(t, oacts[k], REL_TOL, k)
    assert_close(r["ws"].acts["z"], oacts["z"], REL_TOL, "z")
    for i, k in enumerate(("total", "recon", "kl")):
        assert abs(float(r["loss3"][i]) - float(losses[k])) <= REL_TOL * max(1.0, abs(float(losses[k]))), k
    for k, g in zip(O.PARAM_KEYS, r["grads"]):
        assert_close(g, ograds[k], REL_TOL, "grad " + k)


@pytest.mark.parametrize("D,Z,B,mask,launches", [(512, 128, 1024, "3", 10), (256, 64, 4096, "3", 10), (256, 64, 1024, "0", 12),
                                                 (256, 64, 1024, "1", 11), (256, 64, 1024, "2", 11)])
def test_second_stage_folds_forced_on_and_off_keep_parity(ops, monkeypatch, D, Z, B, mask, launches):
    """PSVAE_STAGE2 forces the folds where the default policy leaves them off (Z = 128: two column groups per second
    stage; B = 4096: 1024 tiles per launch, not the persistent kernel) and off where it has them on: every
    combination computes the oracle's numbers"""
    monkeypatch.setenv("PSVAE_STAGE2", mask)
    p = O.init_params(D, Z, 6)
    torch.manual_seed(D + B + int(mask))
    x, eps = torch.randn(B, D), torch.randn(B, Z)
    r = _fused(ops, p, x, eps, 1.1)
    assert r["launches"] == launches
    losses, ograds, oacts, _ = O.train_step(p, x, eps, 1.1, masks=r["masks"])
    check_masks(r["masks"], oacts)
    assert_close(r["outs"][1], oacts["mu"], REL_TOL, "mu")
    assert abs(float(r["loss3"][0]) - float(losses["total"])) <= REL_TOL * max(1.0, abs(float(losses["total"])))
    for k, g in zip(O.PARAM_KEYS, r["grads"]):
        assert_close(g, ograds[k], REL_TOL, "grad " + k)


def test_supplied_kmajor_copies_equal_self_made_ones_bitwise(ops):
    """without params_kmajor the step transposes the weights itself (one more launch) and computes the same bits"""
    D, Z, B = 256, 64, 512
    p = O.init_params(D, Z, 1)
    torch.manual_seed(0)
    x, eps = torch.randn(B, D), torch.randn(B, Z)
    a = _fused(ops, p, x, eps, 1.0, kmajor=True)
    b = _fused(ops, p, x, eps, 1.0, kmajor=False)
    assert b["launches"] == a["launches"] + 1
    assert torch.equal(a["loss3"], b["loss3"])
    for u, v in zip(a["grads"] + list(a["outs"]), b["grads"] + list(b["outs"])):
        assert torch.equal(u, v)


@pytest.mark.parametrize("Z", [64, 32])
def test_two_phase_calls_equal_one_call_bitwise(ops, Z):
    """PHASE_DECODER then PHASE_ENCODER (the overlapped data-parallel step) == PHASE_ALL, with and without the fusion;
    after the decoder phase the decoder gradients and the loss are complete"""
    from pseudo_speaker_vae_amd import _ffi
    D, B = 256, 256
    p = O.init_params(D, Z, 2)
    torch.manual_seed(1)
    x, eps = torch.randn(B, D), torch.randn(B, Z)
    a = _fused(ops, p, x, eps, 0.7)
    b = _fused(ops, p, x, eps, 0.7, phases=(_ffi.PHASE_DECODER, _ffi.PHASE_ENCODER))
    assert torch.equal(a["loss3"], b["loss3"])
    for u, v in zip(a["grads"], b["grads"]):
        assert torch.equal(u, v)
    c = _fused(ops, p, x, eps, 0.7, phases=(_ffi.PHASE_DECODER,))
    assert torch.equal(c["loss3"], a["loss3"])
    for u, v in zip(a["grads"][12:], c["grads"][12:]):
        assert torch.equal(u, v)


@pytest.mark.parametrize("variant", ["cos", "norm", "norm_cos"])
@pytest.mark.parametrize("D,Z,B", [(256, 64, 200), (256, 32, 64)])
def test_loss_variants_inside_the_fused_step(ops, variant, D, Z, B):
    """cosine loss (lightning.py:110-111) and normalize_decoder (model.py:60-61) no longer take the 19-launch separate
    path: one row kernel after the last decoder GEMM; 11 launches (13 without the second-stage fusion)"""
    from pseudo_speaker_vae_amd import _ffi
    normalize, cos = "norm" in variant, "cos" in variant
    flags = (_ffi.NORMALIZE_DECODER if normalize else 0) | (_ffi.COS_LOSS if cos else 0)
    p = O.init_params(D, Z, 3)
    torch.manual_seed(B)
    x, eps = torch.randn(B, D), torch.randn(B, Z)
    r = _fused(ops, p, x, eps, 0.8, flags=flags, go=0.37)
    assert r["launches"] == (11 if Z == 64 else 13)
    losses, ograds, oacts, _ = O.train_step(p, x, eps, 0.8, normalize, cos, go=0.37, masks=r["masks"])
    check_masks(r["masks"], oacts)
    assert_close(r["outs"][0], oacts["x_hat"], REL_TOL, "x_hat")
    for i, k in enumerate(("total", "recon", "kl")):
        assert abs(float(r["loss3"][i]) - float(losses[k])) <= REL_TOL * max(1.0, abs(float(losses[k]))), k
    for k, g in zip(O.PARAM_KEYS, r["grads"]):
        assert_close(g, ograds[k], REL_TOL, "grad " + k)


def test_fused_step_conditional_matches_oracle(ops):
    """SURVEY A9 (build-defined, parity unpinned by the reference): the one-hot gather tables inside the fused fp32
    step -- first-layer epilogues, the bottleneck kernel's decoder.0, the three table gradients -- vs the oracle's
    closed-form definition at equal masks"""
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
    tabd = [tabs[k].to(DEV) for k in O.COND_KEYS]
    gtab = [torch.full_like(t, float("nan")) for t in tabd]
    gd, ad = gender.to(DEV), age.to(DEV)  # keep alive: the struct holds raw pointers
    cond = _ffi.make_cond(tabd, gtab, gd, ad)
    r = _fused(ops, p, x, eps, 1.0, cond=cond)
    assert r["launches"] == 12  # 10 + the two table-gradient launches
    for k in MASK_KEYS:  # backward_cond uses the oracle's own masks: require them equal here (seeded, deterministic)
        assert torch.equal(r["masks"][k], acts[k] > 0), k
    assert_close(r["outs"][0], acts["x_hat"], REL_TOL, "x_hat")
    for k, g in zip(O.PARAM_KEYS, r["grads"]):
        assert_close(g, ref[k], REL_TOL, "grad " + k)
    for k, g in zip(O.COND_KEYS, gtab):
        assert_close(g, tref[k], REL_TOL, "table grad " + k)


def test_engine_conditional_fp32_and_variants_run_the_fused_step():
    """TrainEngine(conditional=True) in fp32 (round-1 gap) and with the cosine loss: the engine's flat buffers,
    k-major copies and label buffers drive the same fused step"""
    from pseudo_speaker_vae_amd.engine import TrainEngine
    D, Z, B = 256, 64, 128
    eng = TrainEngine(D, Z, B, device=DEV, seed=4, conditional=True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, D, generator=g)
    gender, age = torch.randint(-1, 3, (B,), generator=g), torch.randint(-1, 3, (B,), generator=g)
    eng.set_labels(gender, age)
    eng.forward_backward(x.to(DEV))
    torch.cuda.synchronize()
    assert eng.launches_per_step() == 12
    p = {k: v.cpu() for k, v in eng.state_dict().items()}
    tabs = {k: t.cpu() for k, t in zip(O.COND_KEYS, eng.cond_tabs)}
    eps = eng.ws.acts["eps"].cpu()
    acts = O.forward_cond(p, tabs, x, eps, gender.int(), age.int())
    g_xh, g_mu, g_lv = O.elbo_grads(acts["x_hat"], x, acts["mu"], acts["logvar"], 1.0)
    ref, tref, _ = O.backward_cond(p, tabs, x, acts, g_xh, g_mu, g_lv, gender.int(), age.int())
    if all(torch.equal((eng.ws.acts[k] > 0).cpu(), acts[k] > 0) for k in MASK_KEYS):
        for k, gd in zip(O.PARAM_KEYS, eng.grads):
            assert_close(gd, ref[k], REL_TOL, "grad " + k)
        for k, gd in zip(O.COND_KEYS, eng.cond_gtabs):
            assert_close(gd, tref[k], REL_TOL, "table grad " + k)
    assert_close(eng.outs[0], acts["x_hat"], REL_TOL, "x_hat")
    # cosine loss + normalised decoder through the engine
    e2 = TrainEngine(D, Z, B, device=DEV, seed=4, use_cos_loss=True, normalize_decoder=True, beta=0.5)
    e2.forward_backward(x.to(DEV))
    torch.cuda.synchronize()
    assert e2.launches_per_step() == 11
    masks = masks_of(e2.ws.acts)
    p2 = {k: v.cpu() for k, v in e2.state_dict().items()}
    losses, ograds, oacts, _ = O.train_step(p2, x, e2.ws.acts["eps"].cpu(), 0.5, True, True, masks=masks)
    check_masks(masks, oacts)
    assert abs(float(e2.loss3[0]) - float(losses["total"])) <= REL_TOL
    for k, gd in zip(O.PARAM_KEYS, e2.grads):
        assert_close(gd, ograds[k], REL_TOL, "cos+norm grad " + k)


def test_device_state_adam_keeps_both_layouts_and_counts_steps(ops):
    """psvae_adam_step_dev_f32 == torch.optim.Adam semantics (oracle.adam_step) for 3 steps with weight decay and a
    changing learning rate; the k-major copies are the exact transposes of the updated weights; the step counter
    lives on the device"""
    from pseudo_speaker_vae_amd.engine import TrainEngine
    D, Z, B = 256, 64, 64
    eng = TrainEngine(D, Z, B, device=DEV, seed=1, lr=1e-3, weight_decay=0.01)
    g = torch.Generator().manual_seed(2)
    cur = {k: v.cpu().clone() for k, v in eng.state_dict().items()}
    m = {k: torch.zeros_like(v) for k, v in cur.items()}
    v = {k: torch.zeros_like(t) for k, t in cur.items()}
    for step in range(1, 4):
        lr = 1e-3 / step
        grads = [torch.randn(t.shape, generator=g) * 0.1 for t in eng.params]
        for dst, src in zip(eng.grads, grads):
            dst.copy_(src)
        eng.optimizer_step(lr=lr)
        torch.cuda.synchronize()
        for k, gr in zip(O.PARAM_KEYS, grads):
            cur[k], m[k], v[k] = O.adam_step(cur[k], gr, m[k], v[k], step, lr, weight_decay=0.01)
        for k, q, qt in zip(O.PARAM_KEYS, eng.params, eng.params_t):
            # 2e-6: where g ~ -wd * p the kernel's fused multiply-add (one rounding of g + wd p) and the oracle's mul + add
            # differ by an ulp of a number that nearly cancels; Adam's g / (|g| + eps) then amplifies it (measured 1.3e-6)
            assert_close(q, cur[k], 2e-6, f"adam step {step} {k}")
            if q.dim() == 2:
                assert torch.equal(qt, q.t()), f"k-major copy of {k} after step {step}"
        assert eng.adam_steps_done() == step


def test_captured_step_with_adam_equals_eager_training_bitwise():
    """forward + ELBO + backward + Adam as ONE HIP graph (device-resident step counter and Philox state): N replays
    == N eager iterations, bit for bit -- bias correction, fresh noise and the k-major copies included"""
    from pseudo_speaker_vae_amd.engine import TrainEngine
    D, Z, B = 256, 64, 256
    x = torch.randn(B, D, device=DEV)
    e1 = TrainEngine(D, Z, B, device=DEV, seed=3, lr=2e-3)
    e2 = TrainEngine(D, Z, B, device=DEV, seed=3, lr=2e-3)
    e2.capture(x, with_adam=True)           # capturing must not train: the warm-up runs are undone
    torch.cuda.synchronize()
    assert torch.equal(e1.flat_params, e2.flat_params) and torch.equal(e1.flat_params_t, e2.flat_params_t)
    assert torch.equal(e1.exp_avg, e2.exp_avg) and torch.equal(e1.exp_avg_sq, e2.exp_avg_sq)
    assert torch.equal(e1.rng_state, e2.rng_state) and e2.adam_steps_done() == 0
    for it in range(4):
        e1.forward_backward(x)
        e1.optimizer_step()
        e2.replay()
        torch.cuda.synchronize()
        assert torch.equal(e1.flat_params, e2.flat_params), f"iteration {it}"
        assert torch.equal(e1.flat_params_t, e2.flat_params_t)
        assert torch.equal(e1.loss3, e2.loss3)
    assert e1.adam_steps_done() == e2.adam_steps_done() == 4
    assert float(e2.loss3[0]) < 10.0 and torch.isfinite(e2.flat_params).all()


def test_scheduler_step_is_cosine_annealing():
    """TrainEngine.scheduler_step == torch CosineAnnealingLR stepped per epoch (lightning.py:206-213)"""
    from pseudo_speaker_vae_amd.engine import TrainEngine
    eng = TrainEngine(256, 32, 16, device=DEV, lr=0.01, scheduler=dict(T_max=10, eta_min=1e-4))
    w = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([w], lr=0.01)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10, eta_min=1e-4)
    for ep in range(1, 36):  # T_max < max_epochs is a legal configuration: torch's recursion, every epoch
        opt.step()
        sch.step()
        lr = eng.scheduler_step()
        assert abs(lr - opt.param_groups[0]["lr"]) < 1e-12, ep
        if ep <= 10:  # (up to T_max the recursion is the closed form)
            assert abs(lr - O.cosine_lr(0.01, ep, 10, 1e-4)) < 1e-12
    assert abs(eng.scheduler_step(epoch=4) - O.cosine_lr(0.01, 4, 10, 1e-4)) < 1e-15  # explicit epoch: closed form
    import struct
    dev_lr = struct.unpack("<d", eng.adam_state[:8].cpu().numpy().tobytes())[0]
    assert dev_lr == eng.lr  # the captured optimizer reads the rate from the device


def test_label_out_of_range_is_reported_by_the_fused_heads():
    """ADVICE (round 1): y outside [0, C) -- e.g. the -1 "unknown" of the label maps -- must not train silently: the
    reference's F.cross_entropy raises (on a GPU: an asynchronous device-side assert).  The fused kernel turns such a row's
    loss into NaN; the watch (latent_classifier.LabelWatch) checks the first batch before its launch, folds every later
    step's loss into a sticky device flag WITHOUT a host sync, reads the flag every `every` steps and raises; `sync`
    mode checks every batch up front (round 2's behaviour)."""
    from pseudo_speaker_vae_amd import _ffi, latent_classifier as LC
    from pseudo_speaker_vae_amd.latent_classifier import LatentClassifier
    torch.manual_seed(0)
    clf = LatentClassifier(64, 3, num_layers=2, hidden_dim=32).to(DEV)
    z = torch.randn(40, 64, device=DEV)
    good = torch.randint(0, 3, (40,), device=DEV)
    bad = good.clone()
    bad[7] = -1
    saved = LC.label_watch
    try:
        LC.label_watch = w = LC.LabelWatch()
        w.mode, w.every = "1", 4
        with pytest.raises(_ffi.PsvaeError, match="label out of range"):      # first batch of a run: checked at once
            clf.cross_entropy_step(z, bad)
        LC.label_watch = w = LC.LabelWatch()
        w.mode, w.every = "1", 4
        loss, hits = clf.cross_entropy_step(z, good)                            # call 1
        assert torch.isfinite(loss)
        loss_bad, _ = clf.cross_entropy_step(z, bad)                            # call 2: no sync, no raise -- but poisoned
        assert torch.isnan(loss_bad)
        clf.cross_entropy_step(z, good)                                         # call 3
        with pytest.raises(_ffi.PsvaeError, match="label outside"):           # call 4: the poll finds the sticky flag
            clf.cross_entropy_step(z, good)
        for _ in range(4):                                                      # the flag was cleared: clean steps pass
            clf.cross_entropy_step(z, good)
        bad[7] = 3
        clf.cross_entropy_step(z, good); clf.cross_entropy_step(z, good); clf.cross_entropy_step(z, good)
        with pytest.raises(_ffi.PsvaeError, match=r"label out of range \[0, 3\): \[3\]"):  # polled on the bad batch itself
            clf.cross_entropy_step(z, bad)
        LC.label_watch = w = LC.LabelWatch()
        w.mode = "sync"
        clf.cross_entropy_step(z, good)
        with pytest.raises(_ffi.PsvaeError, match="label out of range"):
            clf.cross_entropy_step(z, bad)
    finally:
        LC.label_watch = saved


@pytest.mark.parametrize("cos", [False, True])
def test_elbo_terms_on_hip_match_the_reference_formulas_and_autograd(cos):
    """the general training_step branch (multi-label heads, frozen VAE; lightning.py:110-117) takes recon and kl -- and
    their gradients -- from psvae_elbo_f32 / psvae_elbo_grad_f32: same values and derivatives as the reference's
    F.mse_loss / F.cosine_embedding_loss / KL expression under torch autograd (CPU, fp32)"""
    import torch.nn.functional as F
    from pseudo_speaker_vae_amd.model import VAEModel
    B, D, Z = 200, 256, 64
    torch.manual_seed(5)
    m = VAEModel(D, Z).to(DEV)
    x = torch.randn(B, D)
    xh, mu, lv = (torch.randn(B, D) * 0.7, torch.randn(B, Z) * 0.5, torch.randn(B, Z) * 0.3)
    leaves = [t.clone().to(DEV).requires_grad_() for t in (xh, mu, lv)]
    recon, kl = m.elbo_terms(*leaves, x.to(DEV), cos)
    (1.7 * recon - 0.4 * kl).backward()
    ref = [t.clone().requires_grad_() for t in (xh, mu, lv)]
    r_ref = F.cosine_embedding_loss(ref[0], x, torch.ones(B)) if cos else F.mse_loss(ref[0], x, reduction="mean") / 10
    k_ref = -0.5 * torch.mean(torch.sum(1 + ref[2] - ref[1].pow(2) - ref[2].exp(), dim=-1))
    (1.7 * r_ref - 0.4 * k_ref).backward()
    assert abs(float(recon) - float(r_ref)) <= REL_TOL * max(1.0, abs(float(r_ref)))
    assert abs(float(kl) - float(k_ref)) <= REL_TOL * max(1.0, abs(float(k_ref)))
    for name, a, b in zip(("g_xhat", "g_mu", "g_logvar"), leaves, ref):
        assert_close(a.grad, b.grad, REL_TOL, name)


def test_training_step_general_branch_uses_hip_elbo():
    """multi-label classifier heads take the general branch of training_step: its ELBO terms come from the HIP kernels,
    and the whole thing still backpropagates into every parameter"""
    from ps_vae.lightning import PseudoSpeakerVAE
    D, Z, B = 128, 32, 40
    torch.manual_seed(0)
    mod = PseudoSpeakerVAE(model=dict(input_dim=D, latent_dim=Z),
                           classifier=dict(input_dim=Z, num_classes={"a": 3, "b": 2}, num_layers=2, hidden_dim=32,
                                           activation="relu"), classifier_loss_weight=0.5).to(DEV)
    x = torch.randn(B, D, device=DEV)
    y = {"a": torch.randint(0, 3, (B,), device=DEV), "b": torch.randint(0, 2, (B,), device=DEV)}
    out = mod.training_step((x, y), 0)
    assert type(out["loss"].grad_fn).__name__ != "NoneType"
    out["loss"].backward()
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in mod.parameters())
    r, k = float(mod.logged["train_recon_loss"]), float(mod.logged["train_kl_loss"])
    # (lightning.py:90-95: the per-head log carries the running sum; the mean over the two heads enters the total)
    cum = float(mod.logged["train_classifier_loss_b"])
    assert r > 0 and k > 0 and abs(float(out["loss"]) - (r + k + 0.5 * cum / 2)) < 1e-4
