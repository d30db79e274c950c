#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

What it does
  * imports the reference's own, unmodified `ps_vae/model.py`, `ps_vae/lightning.py`,
    `ps_vae/latent_classifier.py` and `ps_vae/inference.py` from /root/reference (read-only);
  * `pytorch_lightning` and `torchmetrics` are not installed in this image, so two in-memory
    stand-in modules (the build's own ~40 lines below; base-class plumbing only, no math) are put
    in `sys.modules` before the import -- SURVEY.md Appendix C;
  * runs forward / training_step + loss.backward() / decode / conditional_synthesis under fixed
    seeds and stores inputs + outputs as small .npz files (data only, no reference source text).

The epsilon used inside `model.py:57` is recovered by replaying `torch.manual_seed` (verified exact
below).  Parameters are NOT stored (5 MB per case): they are default nn.Linear inits in the
reference's creation order under a seed, which `oracle.psvae_oracle.init_params` reproduces
bit-exactly (asserted here); fixtures keep an fp64 checksum per tensor.  Weight gradients are
stored as a fixed random sample of 2048 entries + fp64 sum / sum-of-squares; bias grads fully.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

# ----------------------------------------------------------------------------- stand-ins
class _HParams(dict):
    __getattr__ = dict.__getitem__


pl = types.ModuleType("pytorch_lightning")


class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self._logged = {}

    def save_hyperparameters(self):
        frame = sys._getframe(1)
        loc = frame.f_locals
        hp = dict(loc.get("hparams", {}))
        for k, v in loc.items():
            if k not in ("self", "hparams", "__class__") and not k.startswith("_"):
                hp.setdefault(k, v)
        self._hparams = _HParams(hp)

    @property
    def hparams(self):
        return self._hparams

    @property
    def device(self):
        return next(self.parameters()).device

    def log(self, name, value, **kw):
        self._logged[name] = float(value)


class Callback:
    pass


pl.LightningModule = LightningModule
pl.Callback = Callback
tm = types.ModuleType("torchmetrics")


class Accuracy(nn.Module):
    def __init__(self, task=None, num_classes=None):
        super().__init__()

    def forward(self, y_hat, y):
        return (y_hat.argmax(-1) == y).float().mean()


tm.Accuracy = Accuracy
sys.modules["pytorch_lightning"] = pl
sys.modules["torchmetrics"] = tm
sys.path.insert(0, REPO)
sys.path.insert(0, REF)  # the reference first: the repo has its own ps_vae/ compat package

from ps_vae.model import VAEModel  # noqa: E402  (the reference)
from ps_vae.lightning import PseudoSpeakerVAE  # noqa: E402
from ps_vae import inference as ref_inference  # noqa: E402
assert os.path.realpath(sys.modules["ps_vae.model"].__file__).startswith(REF), "must import the REFERENCE"

from oracle import psvae_oracle as O  # noqa: E402

N_SAMPLE = 2048


def checksum(t):
    a = t.detach().double().numpy().ravel()
    return np.array([a.sum(), (a * a).sum()], dtype=np.float64)


def sample_idx(numel, seed):
    rng = np.random.default_rng(seed)
    return np.sort(rng.choice(numel, size=min(N_SAMPLE, numel), replace=False)).astype(np.int64)


def make_step_case(name, D, Z, B, pseed, dseed, beta=1.0, normalize=False, cos=False, x_scale=1.0):
    torch.manual_seed(pseed)
    hp = dict(model=dict(input_dim=D, latent_dim=Z, normalize_decoder=normalize),
              kl_loss_weight=beta, use_cos_loss=cos)
    mod = PseudoSpeakerVAE(**hp)
    sd = {k[len("model."):]: v for k, v in mod.state_dict().items() if k.startswith("model.")}
    # the oracle's init must reproduce the reference's parameters bit-for-bit
    op = O.init_params(D, Z, pseed)
    assert list(sd.keys()) == list(O.PARAM_KEYS)
    for k in O.PARAM_KEYS:
        assert torch.equal(sd[k], op[k]), k

    g = torch.Generator().manual_seed(dseed)
    x = torch.randn(B, D, generator=g) * x_scale
    y = torch.zeros(B, dtype=torch.long)
    torch.manual_seed(1000 + dseed)
    out = mod.training_step((x, y), 0)
    torch.manual_seed(1000 + dseed)
    eps = torch.randn(B, Z)
    loss = out["loss"]
    loss.backward()
    # recover forward outputs with the same eps draw
    torch.manual_seed(1000 + dseed)
    with torch.no_grad():
        xh, mu, lv = mod(x)
        z = mu + torch.exp(0.5 * lv) * eps
        xh2 = mod.decode(z)
    assert torch.equal(xh, xh2), "eps replay must be exact"

    fx = dict(D=D, Z=Z, B=B, pseed=pseed, beta=beta, normalize=int(normalize), cos=int(cos),
              x=x.numpy(), eps=eps.numpy(), x_hat=xh.numpy(), mu=mu.numpy(), logvar=lv.numpy(), z=z.numpy(),
              total=np.float64(loss.item()), recon=np.float64(mod._logged["train_recon_loss"]),
              kl=np.float64(mod._logged["train_kl_loss"]))
    for i, k in enumerate(O.PARAM_KEYS):
        p = dict(mod.model.named_parameters())[k]
        fx[f"pck/{k}"] = checksum(p)
        gr = p.grad
        fx[f"gck/{k}"] = checksum(gr)
        if gr.dim() == 1:
            fx[f"g/{k}"] = gr.numpy().copy()
        else:
            idx = sample_idx(gr.numel(), 7 + i)
            fx[f"gi/{k}"] = idx
            fx[f"g/{k}"] = gr.reshape(-1)[idx].numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    print(f"{name}: total={loss.item():.6f} recon={fx['recon']:.6f} kl={fx['kl']:.6f}")


def make_classifier_step_case(name, D, Z, B, num_layers, hidden, cw, beta=1.0):
    """the reference's training_step WITH a LatentClassifier head (lightning.py:73-98): total loss, logged classifier
    loss and accuracy, and every gradient (VAE and classifier) after loss.backward()"""
    torch.manual_seed(5)
    hp = dict(model=dict(input_dim=D, latent_dim=Z, normalize_decoder=False), kl_loss_weight=beta,
              classifier=dict(input_dim=Z, num_classes=3, num_layers=num_layers, hidden_dim=hidden, activation="relu"),
              classifier_loss_weight=cw)
    mod = PseudoSpeakerVAE(**hp)
    with torch.no_grad():
        for q in mod.classifier.parameters():
            q.mul_(3.0)  # default init gives near-uniform logits
    op = O.init_params(D, Z, 5)  # the VAE is created first: its parameters are the seed-5 init
    for k in O.PARAM_KEYS:
        assert torch.equal(dict(mod.model.named_parameters())[k].detach(), op[k]), k
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, D, generator=g)
    y = torch.randint(0, 3, (B,), generator=g)
    torch.manual_seed(2000)
    out = mod.training_step((x, y), 0)
    torch.manual_seed(2000)
    eps = torch.randn(B, Z)
    out["loss"].backward()
    fx = dict(D=D, Z=Z, B=B, pseed=5, beta=beta, cw=cw, x=x.numpy(), y=y.numpy(), eps=eps.numpy(),
              total=np.float64(out["loss"].item()), recon=np.float64(mod._logged["train_recon_loss"]),
              kl=np.float64(mod._logged["train_kl_loss"]), ce=np.float64(mod._logged["train_classifier_loss"]),
              acc=np.float64(mod._logged["train_classifier_acc"]))
    lins = [l for l in mod.classifier.layers if isinstance(l, nn.Linear)]
    fx["n_cls_layers"] = len(lins)
    for i, l in enumerate(lins):
        fx[f"cls_w{i}"] = l.weight.detach().numpy()
        fx[f"cls_b{i}"] = l.bias.detach().numpy()
        fx[f"cls_gw{i}"] = l.weight.grad.numpy().copy()
        fx[f"cls_gb{i}"] = l.bias.grad.numpy().copy()
    for i, k in enumerate(O.PARAM_KEYS):
        gr = dict(mod.model.named_parameters())[k].grad
        fx[f"gck/{k}"] = checksum(gr)
        if gr.dim() == 1:
            fx[f"g/{k}"] = gr.numpy().copy()
        else:
            idx = sample_idx(gr.numel(), 7 + i)
            fx[f"gi/{k}"] = idx
            fx[f"g/{k}"] = gr.reshape(-1)[idx].numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    print(f"{name}: total={fx['total']:.6f} ce={fx['ce']:.6f} acc={fx['acc']:.4f}")


def make_consistency_step_case(name, D, Z, B, hidden, cw, sw, beta=1.0):
    """the reference's training_step with a LatentClassifier head AND a frozen EmbeddingClassifier consistency head
    (lightning.py:44-52,100-108).  The reference builds the head with load_from_checkpoint (a Lightning feature the
    stand-in base does not have), so it is constructed here from the reference's class and attached, frozen and in
    eval mode, exactly as lightning.py:48-50 leaves it.  (self.accuracy only exists with a classifier configured,
    lightning.py:33 vs 103 -- hence both heads.)"""
    from ps_vae.embedding_classifier.embedding_classifier import EmbeddingClassifier
    assert os.path.realpath(sys.modules[EmbeddingClassifier.__module__].__file__).startswith(REF)
    torch.manual_seed(5)
    hp = dict(model=dict(input_dim=D, latent_dim=Z, normalize_decoder=False), kl_loss_weight=beta,
              classifier=dict(input_dim=Z, num_classes=3, num_layers=2, hidden_dim=128, activation="relu"),
              classifier_loss_weight=cw, consistency_loss_weight=sw)
    mod = PseudoSpeakerVAE(**hp)
    torch.manual_seed(9)
    cons = EmbeddingClassifier(input_dim=D, num_classes=3, hidden_dim=hidden)
    with torch.no_grad():
        for q in list(mod.classifier.parameters()) + list(cons.parameters()):
            q.mul_(3.0)  # default init gives near-uniform logits
    for q in cons.parameters():
        q.requires_grad = False
    cons.eval()
    mod.consistency_classifier = cons
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, D, generator=g)
    y = torch.randint(0, 3, (B,), generator=g)
    torch.manual_seed(2100)
    out = mod.training_step((x, y), 0)
    torch.manual_seed(2100)
    eps = torch.randn(B, Z)
    out["loss"].backward()
    fx = dict(D=D, Z=Z, B=B, pseed=5, beta=beta, cw=cw, sw=sw, hidden=hidden, x=x.numpy(), y=y.numpy(), eps=eps.numpy(),
              total=np.float64(out["loss"].item()), recon=np.float64(mod._logged["train_recon_loss"]),
              kl=np.float64(mod._logged["train_kl_loss"]), ce=np.float64(mod._logged["train_classifier_loss"]),
              acc=np.float64(mod._logged["train_classifier_acc"]),
              cons=np.float64(mod._logged["train_consistency_loss"]),
              cons_acc=np.float64(mod._logged["train_consistency"]))
    lins = [l for l in mod.classifier.layers if isinstance(l, nn.Linear)]
    for i, l in enumerate(lins):
        fx[f"cls_w{i}"] = l.weight.detach().numpy()
        fx[f"cls_b{i}"] = l.bias.detach().numpy()
        fx[f"cls_gw{i}"] = l.weight.grad.numpy().copy()
        fx[f"cls_gb{i}"] = l.bias.grad.numpy().copy()
    for i, l in enumerate((cons.fc1, cons.fc2, cons.fc3)):
        fx[f"cons_w{i}"] = l.weight.detach().numpy()
        fx[f"cons_b{i}"] = l.bias.detach().numpy()
    # the head alone on a given input: logits + d CE / d input and its own parameter gradients (training the head)
    cons2 = EmbeddingClassifier(input_dim=D, num_classes=3, hidden_dim=hidden)
    cons2.load_state_dict(cons.state_dict())
    xin = (x * 0.5).clone().requires_grad_()
    logits = cons2(xin)
    l2 = cons2.loss_fn(logits, y)
    l2.backward()
    fx.update(head_x=xin.detach().numpy(), head_logits=logits.detach().numpy(), head_loss=np.float64(l2.item()),
              head_gx=xin.grad.numpy().copy())
    for i, l in enumerate((cons2.fc1, cons2.fc2, cons2.fc3)):
        fx[f"head_gw{i}"] = l.weight.grad.numpy().copy()
        fx[f"head_gb{i}"] = l.bias.grad.numpy().copy()
    for i, k in enumerate(O.PARAM_KEYS):
        gr = dict(mod.model.named_parameters())[k].grad
        fx[f"gck/{k}"] = checksum(gr)
        if gr.dim() == 1:
            fx[f"g/{k}"] = gr.numpy().copy()
        else:
            idx = sample_idx(gr.numel(), 7 + i)
            fx[f"gi/{k}"] = idx
            fx[f"g/{k}"] = gr.reshape(-1)[idx].numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    print(f"{name}: total={fx['total']:.6f} ce={fx['ce']:.6f} cons={fx['cons']:.6f} cons_acc={fx['cons_acc']:.4f}")


def make_decode_case(name, D, Z, N, pseed, normalize):
    torch.manual_seed(pseed)
    m = VAEModel(D, Z, normalize)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(N, Z, generator=g)
    with torch.no_grad():
        xh = m.decode(z)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), D=D, Z=Z, N=N, pseed=pseed,
                        normalize=int(normalize), z=z.numpy(), x_hat=xh.numpy())
    print(name, "ok")


def make_langevin_case(name, D, Z, N, steps, target, step_size, noise_weight, num_layers):
    torch.manual_seed(3)
    hp = dict(model=dict(input_dim=D, latent_dim=Z, normalize_decoder=False),
              classifier=dict(input_dim=Z, num_classes=3, num_layers=num_layers, hidden_dim=128,
                              activation="relu"))
    mod = PseudoSpeakerVAE(**hp)
    # make the classifier non-trivial (default init gives tiny logits)
    with torch.no_grad():
        for p in mod.classifier.parameters():
            p.mul_(4.0)
    torch.manual_seed(11)
    xh, hist = ref_inference.conditional_synthesis(
        mod, num_samples=N, classifier_target=target, step_size=step_size, num_steps=steps,
        noise_weight=noise_weight, return_history=True, device="cpu")
    torch.manual_seed(11)
    z0 = torch.randn(N, Z)
    noises = torch.stack([torch.randn(N, Z) for _ in range(steps)])
    fx = dict(D=D, Z=Z, N=N, steps=steps, target=target, step_size=step_size, noise_weight=noise_weight,
              pseed=3, z0=z0.numpy(), noises=noises.numpy(), hist=np.stack(hist), x_hat=xh.numpy())
    lins = [l for l in mod.classifier.layers if isinstance(l, nn.Linear)]
    fx["n_cls_layers"] = len(lins)
    for i, l in enumerate(lins):
        fx[f"cls_w{i}"] = l.weight.detach().numpy()
        fx[f"cls_b{i}"] = l.bias.detach().numpy()
    # decoder params: needed for the final decode; they are NOT seed-reproducible by init_params here
    # because the classifier is created after the VAE (same RNG stream, VAE first) -> they are.
    op = O.init_params(D, Z, 3)
    for k in O.PARAM_KEYS:
        assert torch.equal(dict(mod.model.named_parameters())[k].detach(), op[k]), k
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    print(name, "ok")


def _ref_general_classifier(Z, num_classes, num_layers, hidden, activation, scale, seed):
    """the REFERENCE's LatentClassifier (latent_classifier.py:7-70) in a configuration its Lightning module cannot build as
    written (lightning.py:39 reads classifier_hparams['label_classes'], a key LatentClassifier does not accept): constructed
    from the reference's own class, unmodified, and attached to the module afterwards"""
    from ps_vae.latent_classifier import LatentClassifier as RefLC
    assert os.path.realpath(sys.modules[RefLC.__module__].__file__).startswith(REF)
    torch.manual_seed(seed)
    clf = RefLC(Z, num_classes, num_layers=num_layers, hidden_dim=hidden, activation=activation)
    with torch.no_grad():
        for p in clf.parameters():
            p.mul_(scale)
    return clf


def _save_classifier(fx, clf, names):
    lins = [l for l in clf.layers if isinstance(l, nn.Linear)]
    fx["n_trunk"] = len(lins) if names is not None else len(lins) - 1
    trunk = lins if names is not None else lins[:-1]
    for i, l in enumerate(trunk):
        fx[f"trunk_w{i}"] = l.weight.detach().numpy()
        fx[f"trunk_b{i}"] = l.bias.detach().numpy()
    heads = [clf.output_layers[k] for k in names] if names is not None else [lins[-1]]
    for i, l in enumerate(heads):
        fx[f"head_w{i}"] = l.weight.detach().numpy()
        fx[f"head_b{i}"] = l.bias.detach().numpy()
    return trunk, heads


def make_langevin_general_case(name, D, Z, N, steps, num_classes, target, num_layers, hidden, activation, step_size,
                               noise_weight):
    """the reference's conditional_synthesis (inference.py:29-110, unmodified) driving the reference's LatentClassifier in
    the configurations the general fused kernel covers: several labels (dict target), the one-logit head, non-ReLU trunks"""
    torch.manual_seed(3)
    mod = PseudoSpeakerVAE(model=dict(input_dim=D, latent_dim=Z, normalize_decoder=False))
    mod.classifier = _ref_general_classifier(Z, num_classes, num_layers, hidden, activation, 4.0, 17)
    torch.manual_seed(11)
    xh, hist = ref_inference.conditional_synthesis(
        mod, num_samples=N, classifier_target=target, step_size=step_size, num_steps=steps,
        noise_weight=noise_weight, return_history=True, device="cpu")
    torch.manual_seed(11)
    z0 = torch.randn(N, Z)
    noises = torch.stack([torch.randn(N, Z) for _ in range(steps)])
    multi = isinstance(num_classes, dict)
    names = list(num_classes.keys()) if multi else None
    fx = dict(D=D, Z=Z, N=N, steps=steps, step_size=step_size, noise_weight=noise_weight, pseed=3, z0=z0.numpy(),
              noises=noises.numpy(), hist=np.stack(hist), x_hat=xh.numpy(), num_layers=num_layers, hidden=hidden,
              activation=np.array(activation), multi=int(multi),
              names=np.array(names if multi else ["-"]), sizes=np.array(list(num_classes.values()) if multi else [num_classes]),
              target_names=np.array(list(target.keys()) if multi else ["-"]),
              target_vals=np.array(list(target.values()) if multi else [target]))
    _save_classifier(fx, mod.classifier, names)
    op = O.init_params(D, Z, 3)
    for k in O.PARAM_KEYS:
        assert torch.equal(dict(mod.model.named_parameters())[k].detach(), op[k]), k
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    print(name, "ok")


def make_multilabel_step_case(name, D, Z, B, num_classes, num_layers, hidden, activation, cw, beta=1.0):
    """the reference's training_step (lightning.py:67-131, unmodified) on its multi-label branch (85-95).  The module
    cannot be BUILT multi-label as written (lightning.py:39, 95: `label_classes`), so the reference's classifier is
    attached after construction with the attributes those lines read: multilabel, accuracy[label], classifier.label_classes."""
    torch.manual_seed(5)
    mod = PseudoSpeakerVAE(model=dict(input_dim=D, latent_dim=Z, normalize_decoder=False), kl_loss_weight=beta,
                           classifier_loss_weight=cw)
    clf = _ref_general_classifier(Z, num_classes, num_layers, hidden, activation, 3.0, 19)
    clf.label_classes = dict(num_classes)
    mod.classifier = clf
    mod.multilabel = True
    mod.accuracy = nn.ModuleDict({k: Accuracy() for k in num_classes})
    g = torch.Generator().manual_seed(29)
    x = torch.randn(B, D, generator=g)
    y = {k: torch.randint(0, n, (B,), generator=g) for k, n in num_classes.items()}
    torch.manual_seed(2200)
    out = mod.training_step((x, y), 0)
    torch.manual_seed(2200)
    eps = torch.randn(B, Z)
    out["loss"].backward()
    names = list(num_classes.keys())
    fx = dict(D=D, Z=Z, B=B, pseed=5, beta=beta, cw=cw, hidden=hidden, num_layers=num_layers, activation=np.array(activation),
              names=np.array(names), sizes=np.array(list(num_classes.values())), x=x.numpy(), eps=eps.numpy(),
              total=np.float64(out["loss"].item()), recon=np.float64(mod._logged["train_recon_loss"]),
              kl=np.float64(mod._logged["train_kl_loss"]))
    for k in names:
        fx[f"y_{k}"] = y[k].numpy()
        fx[f"logged_loss_{k}"] = np.float64(mod._logged[f"train_classifier_loss_{k}"])  # the RUNNING sum (lightning.py:93)
        fx[f"logged_acc_{k}"] = np.float64(mod._logged[f"train_classifier_acc_{k}"])
    trunk, heads = _save_classifier(fx, clf, names)
    for i, l in enumerate(trunk):
        fx[f"trunk_gw{i}"] = l.weight.grad.numpy().copy()
        fx[f"trunk_gb{i}"] = l.bias.grad.numpy().copy()
    for i, l in enumerate(heads):
        fx[f"head_gw{i}"] = l.weight.grad.numpy().copy()
        fx[f"head_gb{i}"] = l.bias.grad.numpy().copy()
    for i, k in enumerate(O.PARAM_KEYS):
        gr = dict(mod.model.named_parameters())[k].grad
        fx[f"gck/{k}"] = checksum(gr)
        if gr.dim() == 1:
            fx[f"g/{k}"] = gr.numpy().copy()
        else:
            idx = sample_idx(gr.numel(), 7 + i)
            fx[f"gi/{k}"] = idx
            fx[f"g/{k}"] = gr.reshape(-1)[idx].numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    print(f"{name}: total={fx['total']:.6f}")


def make_label_maps(name):
    # tables of utils.py:82-136, read by executing those three pure functions out of the file text
    # (ps_vae/utils.py itself cannot import here: it needs pytorch_lightning + torchvision)
    src = open(os.path.join(REF, "ps_vae/utils.py")).read()
    start = src.index("def map_cv_age_to_label")
    end = src.index("def sample_from_cv_metadata")
    ns = {}
    exec(compile(src[start:end], "utils_label_maps", "exec"), ns)  # noqa: S102 (reference functions, run in-container only)
    ages = ["teens", "twenties", "thirties", "fourties", "fifties", "sixties", "seventies", "eighties",
            "nineties", "", "forties", None, "unknown"]
    genders = ["male", "female", "other", "", None, "M", "F", "Male"]
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        ages=np.array([str(a) for a in ages]), age_is_none=np.array([a is None for a in ages]),
        age_labels=np.array([ns["map_cv_age_to_label"](a) for a in ages]),
        genders=np.array([str(g) for g in genders]), gender_is_none=np.array([g is None for g in genders]),
        cv_gender_labels=np.array([ns["map_cv_gender_to_label"](g) for g in genders]),
        vctk_gender_labels=np.array([ns["map_vctk_gender_to_label"](g) for g in genders]))
    print(name, "ok")


if __name__ == "__main__":
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "consistency":  # only the newest family (the rest is unchanged)
        make_consistency_step_case("consstep_d256_z64_b21_h128", 256, 64, 21, 128, 0.3, 0.7)
        make_consistency_step_case("consstep_d512_z128_b9_h64", 512, 128, 9, 64, 1.0, 1.0, beta=0.5)
        sys.exit(0)
    # cfg1 shape (D=256,Z=32), cfg2 shape (256,64), cfg4 shape (512,128); small batches incl. ragged ones
    make_step_case("step_d256_z32_b8_s0", 256, 32, 8, 0, 0)
    make_step_case("step_d256_z64_b16_s0", 256, 64, 16, 0, 0)
    make_step_case("step_d256_z64_b16_s1_beta05", 256, 64, 16, 1, 1, beta=0.5)
    make_step_case("step_d512_z128_b8_s0", 512, 128, 8, 0, 0)
    make_step_case("step_d256_z64_b37_s1", 256, 64, 37, 1, 2)            # ragged batch
    make_step_case("step_d256_z64_b1_s0", 256, 64, 1, 0, 3)              # single row
    make_step_case("step_d784_z20_b32_s0", 784, 20, 32, 0, 4)            # model.py:72 smoke dims
    make_step_case("step_d256_z64_b16_norm", 256, 64, 16, 0, 5, normalize=True)
    make_step_case("step_d256_z64_b16_cos", 256, 64, 16, 0, 6, cos=True)
    make_step_case("step_d256_z64_b16_norm_cos", 256, 64, 16, 1, 7, normalize=True, cos=True)
    make_step_case("step_d256_z64_b16_bignorm", 256, 64, 16, 0, 8, x_scale=18.0)  # ||x|| ~ 290 like plots/norms.png
    make_decode_case("decode_d256_z64_n33", 256, 64, 33, 0, False)
    make_decode_case("decode_d256_z64_n33_norm", 256, 64, 33, 0, True)
    make_langevin_case("langevin_d256_z64_n8_t5", 256, 64, 8, 5, 1, 0.05, 1.0, 2)
    make_langevin_case("langevin_d256_z64_n8_t5_lin", 256, 64, 8, 5, 2, 0.1, 0.5, 1)
    make_langevin_general_case("langevin_ml_tanh_d256_z64_n8_t5", 256, 64, 8, 5, {"gender": 3, "age": 3},
                               {"gender": 1, "age": 2}, 2, 128, "tanh", 0.05, 1.0)
    make_langevin_general_case("langevin_ml_lin_d256_z64_n8_t5", 256, 64, 8, 5, {"gender": 3, "accent": 5, "age": 4},
                               {"age": 3, "gender": 0}, 1, 128, "relu", 0.1, 0.5)   # (a label the target does not name)
    make_langevin_general_case("langevin_onelogit_d256_z64_n8_t5", 256, 64, 8, 5, 1, 1, 2, 128, "relu", 0.05, 1.0)
    make_langevin_general_case("langevin_c20_leaky_d256_z32_n8_t5", 256, 32, 8, 5, 20, 13, 2, 64, "leaky_relu", 0.08, 0.7)
    make_langevin_general_case("langevin_sigmoid_d256_z64_n8_t5", 256, 64, 8, 5, 3, 2, 2, 128, "sigmoid", 0.05, 1.0)
    make_multilabel_step_case("mlstep_d256_z64_b24_tanh", 256, 64, 24, {"gender": 3, "age": 3}, 2, 128, "tanh", 0.3)
    make_multilabel_step_case("mlstep_d256_z64_b19_lin", 256, 64, 19, {"gender": 3, "age": 4, "accent": 5}, 1, 128, "relu",
                              1.0, beta=0.5)
    make_label_maps("label_maps")
    make_classifier_step_case("clfstep_d256_z64_b24_h128", 256, 64, 24, 2, 128, 0.3)
    make_classifier_step_case("clfstep_d256_z64_b19_lin", 256, 64, 19, 1, 128, 1.0, beta=0.5)
    make_consistency_step_case("consstep_d256_z64_b21_h128", 256, 64, 21, 128, 0.3, 0.7)
    make_consistency_step_case("consstep_d512_z128_b9_h64", 512, 128, 9, 64, 1.0, 1.0, beta=0.5)
