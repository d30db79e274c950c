"""bf16 training pass (BASELINE config 3) against the oracle's bf16 emulation (same rounding points).
The reference has no bf16 or conditional code: parity for this path is unpinned by the reference."""
import pytest
import torch

from helpers import assert_close as _assert_close, rel_err
from oracle import psvae_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# bf16 keeps 8 significant bits (ulp 2^-8 = 3.9e-3 relative).  The HIP path and the emulation round at the
# same points, but fp32 sums that differ in the last bit can round to different bf16 neighbours.
BF16_TOL = 6e-3


def assert_close(a, b, tol, what):
    """bf16 emulation checks are bounded against the tensor scale (helpers.assert_close explains why)"""
    _assert_close(a, b, tol, what, elementwise=False)


def assert_close_bf16(a, b, what):
    """gradients: a ReLU-mask or bf16-neighbour flip moves one row of dW by O(1/sqrt(B)), so the bound is on
    the relative L2 error of the whole tensor, with a loose cap on the worst element"""
    a, b = a.double(), b.double()
    l2 = float((a - b).norm() / b.norm().clamp_min(1e-30))
    assert l2 <= 1e-2, f"{what}: relative L2 error {l2:.3e}"
    assert rel_err(a, b) <= 6e-2, f"{what}: worst element {rel_err(a, b):.3e}"


@pytest.fixture(params=["default-tiles", "lds-staged-128x128", "split-wgrad-3", "no-split"])
def tile_mode(request, monkeypatch):
    """the 128x128 LDS-staged kernel normally starts at >= 2048 64x64 tiles; force it for forward / dgrad-only
    launches so the parity cases below cover both main loops (PSVAE_BF16_BIG_MIN is read per launch)"""
    monkeypatch.setenv("PSVAE_BF16_NO_CHAIN", "1")  # these four modes are the launch-per-layer path's (the row chains, the
    if request.param == "lds-staged-128x128":         # default from B = 64 up, have their own cases below)
        monkeypatch.setenv("PSVAE_BF16_BIG_MIN", "1")
    if request.param == "split-wgrad-3":  # the large-batch backward (weight gradients cut over 3 workgroups per tile,
        monkeypatch.setenv("PSVAE_BF16_WGRAD_S", "3")  # some slices empty at small B) forced at every size
    if request.param == "no-split":       # ... and switched off at every size (B = 4096 uses it by default)
        monkeypatch.setenv("PSVAE_BF16_WGRAD_S", "0")
    return request.param


@pytest.mark.parametrize("D,Z,B,cond", [(256, 64, 4096, True), (256, 64, 1024, False), (256, 64, 100, True),
                                        (512, 128, 64, False), (256, 64, 2100, True)])
def test_bf16_train_step_matches_emulation(D, Z, B, cond, tile_mode):
    _check_train_step(D, Z, B, cond)


@pytest.mark.parametrize("D,Z,B,cond,how", [
    (256, 64, 4096, True, "chain-64"),     # 64-row tiles forced (default at this size: 32)
    (256, 64, 4096, True, "per-layer"),    # the launch-per-layer path stays covered at the size the chains took over
    (512, 128, 2048, False, "chain"),      # D = 512: two output staging passes; Z = 128
    (128, 64, 2500, True, "chain"),        # ragged last tile, D = 128
    (512, 64, 2049, True, "chain-64"),     # one row in the last 64-row tile
    (256, 64, 1000, True, "wgrad-loader-wave"),  # the weight-gradient launch's alternative form (ragged last tile included)
    (256, 64, 1000, True, "chain-16"),     # 16-row tiles for every chain (by default only launches of at most 256 such tiles)
    (512, 128, 520, False, "chain-16"),    # ... the 512-wide output staged across the image and the staging buffer
    (256, 128, 3000, False, "chain"),
    (256, 64, 300, False, "chain"),        # 10 tiles, the last one ragged
    (256, 64, 100, True, "chain"),
    (256, 64, 1024, False, "chain"),
    (256, 64, 40, False, "chain-forced"),  # below the chains' batch threshold (64): two tiles, 8 valid rows in the second
])
def test_bf16_row_chains_match_emulation(D, Z, B, cond, how, monkeypatch):
    """B >= 2048 runs every 3-layer MLP of the pass as one chain launch per direction (csrc/chain_bf16.hip) and all weight
    gradients as one launch; same emulation, same bounds as the per-layer path"""
    if how == "chain-64":
        monkeypatch.setenv("PSVAE_CHAIN_RG", "4")
    if how == "chain-16":
        monkeypatch.setenv("PSVAE_CHAIN_RG", "1")
    if how == "wgrad-loader-wave":
        monkeypatch.setenv("PSVAE_WGRAD_LW", "1")
    if how == "per-layer":
        monkeypatch.setenv("PSVAE_BF16_NO_CHAIN", "1")
    if how == "chain-forced":
        monkeypatch.setenv("PSVAE_BF16_CHAIN_MIN", "1")
    _check_train_step(D, Z, B, cond)


@pytest.mark.parametrize("D,Z,B,cond,normalize,cos", [(256, 64, 4096, True, True, True), (256, 64, 300, False, True, False),
                                                      (512, 128, 2048, False, False, True), (128, 64, 1000, True, True, True)])
def test_bf16_loss_variants_match_emulation(D, Z, B, cond, normalize, cos):
    """normalize_decoder_output (model.py:60-61) and the cosine loss (lightning.py:110-111) on the bf16 pass: the decoder
    chain leaves the raw output, the row kernel of the fp32 fused step does the rest and emits the gradient as bf16"""
    from pseudo_speaker_vae_amd import _ffi
    flags = (_ffi.NORMALIZE_DECODER if normalize else 0) | (_ffi.COS_LOSS if cos else 0)
    _check_train_step(D, Z, B, cond, flags)


def test_bf16_loss_variants_need_the_chain_shapes():
    from pseudo_speaker_vae_amd import _ffi, ops
    D, Z, B = 64, 32, 128  # no row chains at these dims
    p = O.init_params(D, Z, 0)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    ws = ops.Bf16Workspace(B, D, Z, DEV)
    with pytest.raises(_ffi.PsvaeError):
        ops.train_step_bf16(ws, torch.randn(B, D, device=DEV), params, [torch.empty_like(t) for t in params],
                            torch.empty(3, device=DEV), flags=_ffi.COS_LOSS)


def _check_train_step(D, Z, B, cond, flags=0):
    from pseudo_speaker_vae_amd import _ffi, ops
    p = O.init_params(D, Z, 1)
    torch.manual_seed(B)
    x, eps = torch.randn(B, D), torch.randn(B, Z)
    tabs = gender = age = None
    c = None
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    keep = []
    if cond:
        tabs = {k: torch.randn(O.COND_DIM, O.HIDDEN) * 0.05 for k in O.COND_KEYS}
        gender = torch.randint(-1, 3, (B,), dtype=torch.int32)
        age = torch.randint(-1, 3, (B,), dtype=torch.int32)
        tabd = [tabs[k].to(DEV) for k in O.COND_KEYS]
        gtab = [torch.full_like(t, float("nan")) for t in tabd]
        gd, ad = gender.to(DEV), age.to(DEV)
        keep = [tabd, gtab, gd, ad]
        c = _ffi.make_cond(tabd, gtab, gd, ad)
    ws = ops.Bf16Workspace(B, D, Z, DEV)
    grads = [torch.full_like(t, float("nan")) for t in params]
    loss3 = torch.full((3,), float("nan"), device=DEV)
    go = 0.9
    ops.train_step_bf16(ws, x.to(DEV), params, grads, loss3, beta=0.7, eps=eps.to(DEV),
                        grad_loss=torch.tensor([go], device=DEV), cond=c, flags=flags)
    torch.cuda.synchronize()
    losses, ref, outs, tg = O.train_step_bf16(p, x, eps, 0.7, go, tabs, gender, age,
                                              normalize_decoder=bool(flags & _ffi.NORMALIZE_DECODER),
                                              use_cos_loss=bool(flags & _ffi.COS_LOSS))
    import os
    chains = (B >= int(os.environ.get("PSVAE_BF16_CHAIN_MIN", 64)) and not os.environ.get("PSVAE_BF16_NO_CHAIN")
              and D in (128, 256, 512) and Z in (64, 128))
    if chains:  # pack, encoders, decoder (+ the variants' row kernel), two gradient chains, weight gradients, slab sums
        assert ops.last_step_launches_bf16() in ((7, 6) if not flags else (8, 7)), ops.last_step_launches_bf16()
    assert_close(ws.x_hat.cpu(), outs["x_hat"], BF16_TOL, "x_hat")
    assert_close(ws.mu.cpu(), outs["mu"], BF16_TOL, "mu")
    assert_close(ws.logvar.cpu(), outs["logvar"], BF16_TOL, "logvar")
    for i, k in enumerate(("total", "recon", "kl")):
        assert abs(float(loss3[i]) - float(losses[k])) <= BF16_TOL * max(1.0, abs(float(losses[k]))), k
    for k, g in zip(O.PARAM_KEYS, grads):
        assert torch.isfinite(g).all(), k
        assert_close_bf16(g.cpu(), ref[k], "grad " + k)
    if cond:
        for k, g in zip(O.COND_KEYS, keep[1]):
            assert_close_bf16(g.cpu(), tg[k], "grad " + k)
    # and the bf16 path stays close to the fp32 oracle (sanity of the emulation itself)
    if not cond and flags == 0:
        f32_losses, f32_grads, _, _ = O.train_step(p, x, eps, 0.7, go=go)
        assert abs(float(loss3[0]) - float(f32_losses["total"])) < 2e-2
        assert rel_err(grads[16].cpu(), f32_grads["decoder.4.weight"]) < 5e-2


def test_bf16_rejects_unsupported_dims():
    from pseudo_speaker_vae_amd import _ffi, ops
    with pytest.raises(_ffi.PsvaeError):
        D, Z, B = 784, 20, 8
        p = O.init_params(D, Z, 0)
        params = [p[k].to(DEV) for k in O.PARAM_KEYS]
        ws = ops.Bf16Workspace(B, D, Z, DEV)
        ops.train_step_bf16(ws, torch.randn(B, D, device=DEV), params, [torch.empty_like(t) for t in params],
                            torch.empty(3, device=DEV))


@pytest.mark.parametrize("D,Z,N,normalize", [(256, 64, 1000, False), (256, 64, 70000, False), (512, 128, 300, True)])
def test_bf16_decode_matches_emulation(D, Z, N, normalize):
    """psvae_decode_bf16 against the oracle's bf16 emulation of VAEModel.decode (same rounding points), and within
    bf16 accuracy of the fp32 decode"""
    from pseudo_speaker_vae_amd import _ffi, ops
    p = O.init_params(D, Z, 2)
    torch.manual_seed(N)
    z = torch.randn(N, Z)
    dec = [p[k].to(DEV) for k in O.PARAM_KEYS[12:]]
    flags = _ffi.NORMALIZE_DECODER if normalize else 0
    got = ops.decode_bf16(z.to(DEV), dec, D, flags=flags)
    torch.cuda.synchronize()
    ref = O.decode_bf16(p, z, normalize)
    assert_close(got.cpu(), ref, BF16_TOL, "x_hat vs bf16 emulation")
    full = ops.decode(z.to(DEV), dec, D, flags=flags)
    assert rel_err(got.cpu(), full.cpu()) <= 3e-2, "bf16 decode vs fp32 decode"


@pytest.mark.parametrize("B,cond", [(4096, True), (200, False)])
def test_bf16_forward_plus_backward_equals_train_step(B, cond):
    """psvae_forward_bf16 followed by psvae_backward_bf16 on the same workspace gives bitwise what
    psvae_train_step_bf16 gives (outputs, loss, every gradient), and the forward alone leaves no gradient touched"""
    from pseudo_speaker_vae_amd import _ffi, ops
    D, Z = 256, 64
    p = O.init_params(D, Z, 4)
    torch.manual_seed(B)
    x, eps = torch.randn(B, D).to(DEV), torch.randn(B, Z).to(DEV)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    keep, c1, c2 = [], None, None
    if cond:
        tabd = [(torch.randn(O.COND_DIM, O.HIDDEN) * 0.05).to(DEV) for _ in O.COND_KEYS]
        gd, ad = torch.randint(-1, 3, (B,), dtype=torch.int32).to(DEV), torch.randint(-1, 3, (B,), dtype=torch.int32).to(DEV)
        gt1 = [torch.full_like(t, float("nan")) for t in tabd]
        gt2 = [torch.full_like(t, float("nan")) for t in tabd]
        c1, c2 = _ffi.make_cond(tabd, gt1, gd, ad), _ffi.make_cond(tabd, gt2, gd, ad)
        keep = [tabd, gd, ad, gt1, gt2]
    go = torch.tensor([0.8], device=DEV)
    ws1, ws2 = ops.Bf16Workspace(B, D, Z, DEV), ops.Bf16Workspace(B, D, Z, DEV)
    g1 = [torch.full_like(t, float("nan")) for t in params]
    g2 = [torch.full_like(t, float("nan")) for t in params]
    l1, l2 = torch.full((3,), float("nan"), device=DEV), torch.full((3,), float("nan"), device=DEV)
    ops.train_step_bf16(ws1, x, params, g1, l1, beta=0.6, eps=eps, grad_loss=go, cond=c1)
    ops.forward_bf16(ws2, x, params, eps=eps, grad_loss=go, cond=c2)
    torch.cuda.synchronize()
    assert torch.equal(ws1.x_hat, ws2.x_hat) and torch.equal(ws1.mu, ws2.mu) and torch.equal(ws1.logvar, ws2.logvar)
    assert all(torch.isnan(g).all() for g in g2) and torch.isnan(l2).all()
    ops.backward_bf16(ws2, x, params, g2, l2, beta=0.6, grad_loss=go, cond=c2)
    torch.cuda.synchronize()
    assert torch.equal(l1, l2)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
    if cond:
        for a, b in zip(keep[3], keep[4]):
            assert torch.equal(a, b)


def test_bf16_engine_runs_the_loss_variants():
    """TrainEngine(dtype="bf16", normalize_decoder=True, use_cos_loss=True) (round-1 gap: the bf16 engine rejected every
    variant): its step stays within bf16 accuracy of the fp32 engine on the same parameters and noise"""
    from pseudo_speaker_vae_amd.engine import TrainEngine
    D, Z, B = 256, 64, 512
    kw = dict(device=DEV, seed=3, normalize_decoder=True, use_cos_loss=True, beta=0.3)
    e32 = TrainEngine(D, Z, B, **kw)
    e16 = TrainEngine(D, Z, B, dtype="bf16", **kw)
    for a, b in zip(e16.params, e32.params):  # (same seed, same init; copied anyway)
        a.copy_(b)
    x = torch.randn(B, D, generator=torch.Generator().manual_seed(1)).to(DEV)
    eps = torch.randn(B, Z, generator=torch.Generator().manual_seed(2)).to(DEV)
    e32.forward_backward(x, eps=eps)
    e16.forward_backward(x, eps=eps)
    torch.cuda.synchronize()
    l32, l16 = e32.loss3.cpu(), e16.loss3.cpu()
    assert torch.isfinite(l16).all()
    assert float((l32 - l16).abs().max()) < 2e-2 * max(1.0, float(l32.abs().max()))
    assert_close(e16.outs[0].norm(dim=1).cpu(), torch.ones(B), 1e-5, "unit rows")
    for k, a, b in zip(O.PARAM_KEYS, e16.grads, e32.grads):  # bf16 against TRUE fp32 (not its emulation): a few per cent
        l2 = float((a.double() - b.double()).norm() / b.double().norm())
        assert l2 < 0.1, f"grad {k}: relative L2 distance {l2:.3e} from the fp32 engine"
    with pytest.raises(ValueError):
        TrainEngine(64, 32, 128, dtype="bf16", device=DEV, use_cos_loss=True)


@pytest.mark.parametrize("B", [4096, 300])
def test_bf16_pass_draws_the_same_philox_epsilon_as_the_fp32_path(B):
    """the decoder chain's input stage draws epsilon itself: same Philox block per element as reparam_kl_kernel / the fp32
    step (block = flat element index / 4 from the device-resident {seed, offset}), and the backward advances the state by
    the blocks the batch consumed"""
    from pseudo_speaker_vae_amd import ops
    D, Z = 256, 64
    p = O.init_params(D, Z, 5)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    x = torch.randn(B, D, generator=torch.Generator().manual_seed(B)).to(DEV)
    ws = ops.Bf16Workspace(B, D, Z, DEV)
    state = torch.tensor([1234, 777], dtype=torch.int64, device=DEV)
    grads = [torch.empty_like(t) for t in params]
    loss3 = torch.empty(3, device=DEV)
    ops.train_step_bf16(ws, x, params, grads, loss3, rng_state=state)
    torch.cuda.synchronize()
    assert ops.last_step_launches_bf16() in (6, 7)  # (no slab sums where the batch is one slice)
    ref = ops.philox_normal(B * Z, 1234, 777, DEV, fast=True).view(B, Z)  # (the in-kernel form of the stream: psvae.h)
    assert torch.equal(ws.eps, ref)
    assert_close(ws.eps.cpu(), torch.from_numpy(O.philox_normal(B * Z, 1234, 777).reshape(B, Z)), 1e-5, "epsilon vs the oracle's stream")
    assert int(state[1]) == 777 + B * Z // 4 and int(state[0]) == 1234
    z = ws.mu + torch.exp(0.5 * ws.logvar) * ws.eps
    assert_close(ws.z.cpu(), z.cpu(), 1e-6, "z = mu + exp(logvar / 2) eps")
    assert torch.isfinite(loss3).all()


def test_bf16_row_chains_are_deterministic_and_row_local():
    """size-independent properties at cfg3's size: a second run is bit-identical (fixed-order slab sums, no atomics), and every
    row's forward depends on that row alone -- the first 2,048 rows run as their own batch reproduce the big run's rows bit
    for bit (tile boundaries are invisible), the loss partials add up, and the weight gradients are the row-shards' mean"""
    from pseudo_speaker_vae_amd import ops
    D, Z, B = 256, 64, 4096
    p = O.init_params(D, Z, 8)
    params = [p[k].to(DEV) for k in O.PARAM_KEYS]
    g = torch.Generator().manual_seed(4)
    x, eps = torch.randn(B, D, generator=g).to(DEV), torch.randn(B, Z, generator=g).to(DEV)

    def run(xs, es):
        ws = ops.Bf16Workspace(xs.shape[0], D, Z, DEV)
        grads = [torch.full_like(t, float("nan")) for t in params]
        loss3 = torch.full((3,), float("nan"), device=DEV)
        ops.train_step_bf16(ws, xs, params, grads, loss3, beta=0.5, eps=es)
        torch.cuda.synchronize()
        return ws, grads, loss3

    w1, g1, l1 = run(x, eps)
    w2, g2, l2 = run(x, eps)
    assert torch.equal(l1, l2) and all(torch.equal(a, b) for a, b in zip(g1, g2)) and torch.equal(w1.x_hat, w2.x_hat)
    halves = [run(x[h * 2048:(h + 1) * 2048].contiguous(), eps[h * 2048:(h + 1) * 2048].contiguous()) for h in range(2)]
    for h, (wh, gh, lh) in enumerate(halves):
        sl = slice(h * 2048, (h + 1) * 2048)
        assert torch.equal(wh.mu, w1.mu[sl]) and torch.equal(wh.logvar, w1.logvar[sl]) and torch.equal(wh.x_hat, w1.x_hat[sl])
    lm = (halves[0][2] + halves[1][2]) / 2
    assert_close(lm.cpu(), l1.cpu(), 2e-6, "loss = mean of the halves' losses")
    for k, a, b0, b1 in zip(O.PARAM_KEYS, g1, halves[0][1], halves[1][1]):
        # (each half normalises by its own 2,048 rows and rounds its MSE / KL gradients to bf16 at that scale: the mean of
        #  the shards' gradients equals the full-batch gradient up to those roundings)
        assert_close_bf16(((b0 + b1) / 2).cpu(), a.cpu(), "shard mean " + k)


def test_bf16_packs_kept_by_the_optimizer_step_equal_packs_rebuilt_in_the_pass(monkeypatch):
    """PSVAE_BF16_PACKS_CURRENT (include/psvae.h): a pass that is told the bf16 weight packs in its scratch are current
    skips the pack launch and gives the SAME bits as a pass that rebuilds them; the engine keeps them current across its
    fused Adam and notices parameters written through torch ops (their version counter)"""
    from pseudo_speaker_vae_amd import _ffi, ops
    from pseudo_speaker_vae_amd.engine import TrainEngine
    monkeypatch.delenv("PSVAE_BF16_PACK_IN_STEP", raising=False)
    D, Z, B = 256, 64, 640
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, D, generator=g).to(DEV)
    eps = torch.randn(B, Z, generator=g).to(DEV)
    gender, age = torch.randint(-1, 3, (B,), generator=g), torch.randint(-1, 3, (B,), generator=g)
    eng = TrainEngine(D, Z, B, device=DEV, seed=4, dtype="bf16", conditional=True, lr=1e-2)
    eng.set_labels(gender, age)

    def raw_pass():  # the C entry without the flag, on its own workspace: packs rebuilt inside the pass
        ws = ops.Bf16Workspace(B, D, Z, DEV)
        grads = [torch.full_like(t, float("nan")) for t in eng.params]
        gtab = [torch.full_like(t, float("nan")) for t in eng.cond_tabs]
        c = _ffi.make_cond(list(eng.cond_tabs), gtab, eng.gender, eng.age)
        loss3 = torch.full((3,), float("nan"), device=DEV)
        ops.train_step_bf16(ws, x, eng.params, grads, loss3, beta=eng.beta, eps=eps, cond=c)
        n = ops.last_step_launches_bf16()
        torch.cuda.synchronize()
        return grads, gtab, loss3, n

    for it in range(3):
        eng.forward_backward(x, eps=eps)
        assert eng.launches_per_step() == 6
        torch.cuda.synchronize()
        grads, gtab, loss3, n = raw_pass()
        assert n == 7
        for k, a, b in zip(O.PARAM_KEYS, eng.grads, grads):
            assert torch.equal(a, b), f"iteration {it}: grad {k} differs from the pass that packs for itself"
        assert torch.equal(eng.loss3, loss3)
        if it == 0:
            eng.optimizer_step()               # Adam changes the master weights and repacks
        elif it == 1:
            with torch.no_grad():              # a write through torch: noticed by the version counter
                eng.params[2].mul_(1.25)
                eng.params[16].add_(0.01)
    # a stale pack is what the flag would give without the repack: the raw entry with the flag on a scratch whose packs
    # were made from OTHER weights must differ (the flag is honoured, not ignored)
    ws = ops.Bf16Workspace(B, D, Z, DEV)
    other = [t.clone() for t in eng.params]
    other[2].mul_(0.5)
    ops.pack_params_bf16(ws, other)
    grads = [torch.zeros_like(t) for t in eng.params]
    c = _ffi.make_cond(list(eng.cond_tabs), [torch.zeros_like(t) for t in eng.cond_tabs], eng.gender, eng.age)
    ops.train_step_bf16(ws, x, eng.params, grads, torch.zeros(3, device=DEV), beta=eng.beta, eps=eps, cond=c,
                        flags=_ffi.BF16_PACKS_CURRENT)
    torch.cuda.synchronize()
    assert ops.last_step_launches_bf16() == 6
    assert not torch.equal(grads[4], eng.grads[4])
