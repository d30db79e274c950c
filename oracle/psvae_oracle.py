"""CPU ORACLE for the pseudo-speaker VAE hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain torch-CPU / numpy, closed-form forward AND backward,
no autograd) of the reference's algorithm for the hot path named in BASELINE.json:

    VAEModel.forward / decode          /root/reference/ps_vae/model.py:38-69
    ELBO in training_step              /root/reference/ps_vae/lightning.py:110-124
    (backward = autograd of the above; closed forms of SURVEY.md Appendix A)
    Adam + CosineAnnealingLR           /root/reference/ps_vae/lightning.py:204-214
    Langevin conditional sampler       /root/reference/ps_vae/inference.py:29-110
    label -> index maps                /root/reference/ps_vae/utils.py:82-136

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it,
and only as the checker / the timed CPU baseline -- never as part of the product path.
The product (pseudo_speaker_vae_amd) must not import anything from `oracle/`.

Parity pinning: the reference ships no tests or golden vectors, so this oracle
is pinned against outputs of the reference itself, imported in the build container by
`tests/golden/make_golden.py` (forward, losses, all 18 parameter gradients, decode, Langevin).
The committed fixtures live in `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks the
oracle against every one of them.  The one-hot conditional extension (cfg3) has NO reference
semantics: for it only the label->index maps are pinned; the concat itself
is "parity unpinned" and defined by `forward_cond_concat` below.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

HIDDEN = 512  # hard-coded hidden width, model.py:15-35

# state_dict key order == creation order in model.py:14-36 (row A1 of SURVEY.md section 8a)
PARAM_KEYS = tuple(
    f"{blk}.{i}.{wb}"
    for blk in ("encoder_mu", "encoder_sigma", "decoder")
    for i in (0, 2, 4)
    for wb in ("weight", "bias")
)


def param_shapes(D: int, Z: int, H: int = HIDDEN) -> Dict[str, Tuple[int, ...]]:
    """Shapes of the 18 tensors (model.py:14-36): nn.Linear weight is [out, in]."""
    enc = [(H, D), (H,), (H, H), (H,), (Z, H), (Z,)]
    dec = [(H, Z), (H,), (H, H), (H,), (D, H), (D,)]
    return dict(zip(PARAM_KEYS, enc + enc + dec))


def init_params(D: int, Z: int, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Default nn.Linear init in the reference's creation order (model.py:14-36) under `seed`.

    Built from torch.nn.Linear directly (not from the reference) so it can run on the GPU box.
    """
    torch.manual_seed(seed)
    dims = [(D, HIDDEN), (HIDDEN, HIDDEN), (HIDDEN, Z)] * 2 + [(Z, HIDDEN), (HIDDEN, HIDDEN), (HIDDEN, D)]
    out: Dict[str, torch.Tensor] = {}
    it = iter(PARAM_KEYS)
    for fan_in, fan_out in dims:
        lin = torch.nn.Linear(fan_in, fan_out)
        out[next(it)] = lin.weight.detach().clone()
        out[next(it)] = lin.bias.detach().clone()
    return out


# --------------------------------------------------------------------------------------
# forward  (model.py:54-61)
# --------------------------------------------------------------------------------------
def _mlp3(x, p, blk):
    """Sequential(Linear, ReLU, Linear, ReLU, Linear) -- model.py:14-36."""
    h1 = torch.relu(torch.addmm(p[f"{blk}.0.bias"], x, p[f"{blk}.0.weight"].t()))
    h2 = torch.relu(torch.addmm(p[f"{blk}.2.bias"], h1, p[f"{blk}.2.weight"].t()))
    y = torch.addmm(p[f"{blk}.4.bias"], h2, p[f"{blk}.4.weight"].t())
    return h1, h2, y


def normalize_rows(y: torch.Tensor, eps: float = 1e-12):
    """F.normalize(p=2, dim=1) -- model.py:60-61: y / max(||y||_2, eps)."""
    n = y.pow(2).sum(dim=1, keepdim=True).sqrt().clamp_min(eps)
    return y / n, n


def forward(p: Dict[str, torch.Tensor], x: torch.Tensor, eps: torch.Tensor,
            normalize_decoder: bool = False) -> Dict[str, torch.Tensor]:
    """VAEModel.forward (model.py:54-63) with the N(0,1) draw `eps` injected.

    Returns every intermediate; (x_hat, mu, logvar) is the reference's return tuple
    (the third element is a log-VARIANCE, model.py:55-56).
    """
    h1m, h2m, mu = _mlp3(x, p, "encoder_mu")           # model.py:54
    h1s, h2s, lv = _mlp3(x, p, "encoder_sigma")        # model.py:55
    sigma = torch.exp(0.5 * lv)                         # model.py:56
    z = mu + sigma * eps                                # model.py:57
    d1, d2, y = _mlp3(z, p, "decoder")                  # model.py:58
    out = dict(h1m=h1m, h2m=h2m, mu=mu, h1s=h1s, h2s=h2s, logvar=lv, sigma=sigma, eps=eps,
               z=z, d1=d1, d2=d2, y=y)
    if normalize_decoder:                               # model.py:60-61
        out["x_hat"], out["ynorm"] = normalize_rows(y)
    else:
        out["x_hat"] = y
    return out


def decode(p: Dict[str, torch.Tensor], z: torch.Tensor, normalize_decoder: bool = False) -> torch.Tensor:
    """VAEModel.decode (model.py:65-69)."""
    _, _, y = _mlp3(z, p, "decoder")
    return normalize_rows(y)[0] if normalize_decoder else y


# --------------------------------------------------------------------------------------
# ELBO  (lightning.py:110-124)
# --------------------------------------------------------------------------------------
COS_EPS = 1e-12  # ATen cosine_embedding_loss EPSILON


def elbo(x_hat, x, mu, lv, beta: float = 1.0, use_cos_loss: bool = False):
    """recon (lightning.py:110-113), kl (115-117), total (119-124, no classifier terms)."""
    B, D = x.shape
    if use_cos_loss:  # F.cosine_embedding_loss(x_hat, x, ones): mean_b(1 - cos)
        dot = (x_hat * x).sum(1)
        den = torch.sqrt((x_hat.pow(2).sum(1) + COS_EPS) * (x.pow(2).sum(1) + COS_EPS))
        recon = (1.0 - dot / den).mean()
    else:
        recon = (x_hat - x).pow(2).sum() / (B * D) / 10.0
    kl = -0.5 * torch.sum(1 + lv - mu.pow(2) - lv.exp()) / B
    return recon, kl, recon + beta * kl


def elbo_grads(x_hat, x, mu, lv, beta: float = 1.0, use_cos_loss: bool = False, go: float = 1.0):
    """d total / d(x_hat, mu, lv) -- the direct (non-decoder) terms; SURVEY.md Appendix A."""
    B, D = x.shape
    if use_cos_loss:
        a = x_hat.pow(2).sum(1, keepdim=True) + COS_EPS
        b = x.pow(2).sum(1, keepdim=True) + COS_EPS
        den = torch.sqrt(a * b)
        cos = (x_hat * x).sum(1, keepdim=True) / den
        g_xh = -(x / den - cos * x_hat / a) / B
    else:
        g_xh = 2.0 * (x_hat - x) / (10.0 * B * D)
    g_mu = beta * mu / B
    g_lv = beta * (lv.exp() - 1.0) / (2.0 * B)
    return go * g_xh, go * g_mu, go * g_lv


# --------------------------------------------------------------------------------------
# backward (autograd of model.py:54-61; closed forms of SURVEY.md Appendix A)
# --------------------------------------------------------------------------------------
def _mlp3_backward(g_y, x_in, h1, h2, p, blk, grads, need_dx: bool, m1=None, m2=None):
    """m1/m2: optional ReLU masks overriding (h1 > 0)/(h2 > 0) -- see `backward(masks=...)`."""
    W2, W3 = p[f"{blk}.2.weight"], p[f"{blk}.4.weight"]
    grads[f"{blk}.4.weight"] = g_y.t() @ h2
    grads[f"{blk}.4.bias"] = g_y.sum(0)
    g_h2 = (g_y @ W3) * ((h2 > 0) if m2 is None else m2)
    grads[f"{blk}.2.weight"] = g_h2.t() @ h1
    grads[f"{blk}.2.bias"] = g_h2.sum(0)
    g_h1 = (g_h2 @ W2) * ((h1 > 0) if m1 is None else m1)
    grads[f"{blk}.0.weight"] = g_h1.t() @ x_in
    grads[f"{blk}.0.bias"] = g_h1.sum(0)
    return (g_h1 @ p[f"{blk}.0.weight"]) if need_dx else None


def backward(p, x, acts, g_xhat, g_mu_direct=None, g_lv_direct=None,
             normalize_decoder: bool = False, need_dx: bool = False, masks=None):
    """Gradients of the 18 parameters given d/d(x_hat), plus direct d/d(mu), d/d(logvar) terms.

    `masks` (optional dict h1m,h2m,h1s,h2s,d1,d2 -> bool tensors) replaces the ReLU derivative
    masks (h > 0).  The ReLU derivative is discontinuous at 0: two fp32 implementations whose
    pre-activations differ in the last bit can legitimately disagree on the mask of an element that
    sits within rounding of 0, which moves a whole row of dW by O(1/sqrt(B)).  Parity tests therefore
    compare gradients at EQUAL masks (taken from the implementation under test) and separately
    check that the masks differ from the oracle's only where |activation| is within rounding of 0."""
    grads: Dict[str, torch.Tensor] = {}
    mk = (lambda k: None) if masks is None else (lambda k: masks[k])
    g_y = g_xhat
    if normalize_decoder:  # backward of y / max(||y||, eps)
        xh, n = acts["x_hat"], acts["ynorm"]
        clamped = n <= 1e-12
        g_y = torch.where(clamped, g_xhat / n, (g_xhat - xh * (g_xhat * xh).sum(1, keepdim=True)) / n)
    g_z = _mlp3_backward(g_y, acts["z"], acts["d1"], acts["d2"], p, "decoder", grads, True, mk("d1"), mk("d2"))
    g_mu = g_z if g_mu_direct is None else g_z + g_mu_direct
    g_lv = g_z * acts["eps"] * 0.5 * acts["sigma"]
    if g_lv_direct is not None:
        g_lv = g_lv + g_lv_direct
    gx_m = _mlp3_backward(g_mu, x, acts["h1m"], acts["h2m"], p, "encoder_mu", grads, need_dx, mk("h1m"), mk("h2m"))
    gx_s = _mlp3_backward(g_lv, x, acts["h1s"], acts["h2s"], p, "encoder_sigma", grads, need_dx, mk("h1s"), mk("h2s"))
    extra = dict(g_z=g_z, g_mu=g_mu, g_lv=g_lv)
    if need_dx:
        extra["g_x"] = gx_m + gx_s
    return {k: grads[k] for k in PARAM_KEYS}, extra


def train_step(p, x, eps, beta: float = 1.0, normalize_decoder: bool = False,
               use_cos_loss: bool = False, go: float = 1.0, masks=None):
    """forward + ELBO + backward: what one Lightning training_step + loss.backward() computes
    (lightning.py:67-131 without classifier / consistency branches)."""
    acts = forward(p, x, eps, normalize_decoder)
    recon, kl, total = elbo(acts["x_hat"], x, acts["mu"], acts["logvar"], beta, use_cos_loss)
    g_xh, g_mu_d, g_lv_d = elbo_grads(acts["x_hat"], x, acts["mu"], acts["logvar"], beta, use_cos_loss, go)
    grads, extra = backward(p, x, acts, g_xh, g_mu_d, g_lv_d, normalize_decoder, masks=masks)
    return dict(recon=recon, kl=kl, total=total), grads, acts, extra


# --------------------------------------------------------------------------------------
# label -> index maps (utils.py:82-136) and the build-defined one-hot conditional VAE (row A9)
# --------------------------------------------------------------------------------------
CV_AGE_TO_LABEL = {  # utils.py:92-102
    "teens": 0, "twenties": 0,
    "thirties": 1, "fourties": 1, "fifties": 1,
    "sixties": 2, "seventies": 2, "eighties": 2, "nineties": 2,
}
CV_GENDER_TO_LABEL = {"male": 0, "female": 1, "other": 2}  # utils.py:115-119
VCTK_GENDER_TO_LABEL = {"M": 0, "F": 1}  # utils.py:132-135


def map_cv_age_to_label(age) -> int:  # utils.py:82-103
    return CV_AGE_TO_LABEL.get(age, -1)


def map_cv_gender_to_label(gender) -> int:  # utils.py:105-120
    return CV_GENDER_TO_LABEL.get(gender, -1)


def map_vctk_gender_to_label(gender) -> int:  # utils.py:122-136
    return VCTK_GENDER_TO_LABEL.get(gender, -1)


N_GENDER, N_AGE = 3, 3
COND_DIM = N_GENDER + N_AGE


def onehot_cond(gender: torch.Tensor, age: torch.Tensor) -> torch.Tensor:
    """c = [onehot3(gender), onehot3(age)] in {0,1}^6; an unknown label (-1) gives an all-zero block."""
    B = gender.shape[0]
    c = torch.zeros(B, COND_DIM)
    for b in range(B):
        if 0 <= int(gender[b]) < N_GENDER:
            c[b, int(gender[b])] = 1.0
        if 0 <= int(age[b]) < N_AGE:
            c[b, N_GENDER + int(age[b])] = 1.0
    return c


def forward_cond_concat(p_cat, x, eps, gender, age):
    """DEFINITION of the cfg3 conditional VAE (no reference code exists for it -- parity unpinned):
    encoder input [x || c], decoder input [z || c]; `p_cat` has first-layer weights of shape
    [512, D+6] (encoders) and [512, Z+6] (decoder); everything else as model.py:54-58."""
    c = onehot_cond(gender, age)
    xc = torch.cat([x, c], dim=1)
    h1m, h2m, mu = _mlp3(xc, p_cat, "encoder_mu")
    h1s, h2s, lv = _mlp3(xc, p_cat, "encoder_sigma")
    sigma = torch.exp(0.5 * lv)
    z = mu + sigma * eps
    zc = torch.cat([z, c], dim=1)
    d1, d2, y = _mlp3(zc, p_cat, "decoder")
    return dict(h1m=h1m, h2m=h2m, mu=mu, h1s=h1s, h2s=h2s, logvar=lv, sigma=sigma, eps=eps, z=z,
                d1=d1, d2=d2, y=y, x_hat=y, c=c)


COND_KEYS = ("cond.encoder_mu", "cond.encoder_sigma", "cond.decoder")


def split_cond_params(p_cat, D: int, Z: int):
    """concat parameterisation -> (plain 18 tensors, three [6, 512] gather tables).
    x_c W^T = x W[:, :D]^T + sum_j c_j W[:, D+j]  => table[j] = W[:, D+j]."""
    p = dict(p_cat)
    tabs = {}
    for blk, n_in in (("encoder_mu", D), ("encoder_sigma", D), ("decoder", Z)):
        W = p_cat[f"{blk}.0.weight"]
        p[f"{blk}.0.weight"] = W[:, :n_in].contiguous()
        tabs[f"cond.{blk}"] = W[:, n_in:].t().contiguous()  # [6, 512]
    return p, tabs


def forward_cond(p, tabs, x, eps, gender, age):
    """Same function as forward_cond_concat in the gather-add form the HIP path uses:
    the one-hot block never materialises; first-layer bias gets table rows added per sample."""
    def rows(tab):
        out = torch.zeros(x.shape[0], tab.shape[1])
        for lab, base, n in ((gender, 0, N_GENDER), (age, N_GENDER, N_AGE)):
            ok = (lab >= 0) & (lab < n)
            out[ok] += tab[base + lab[ok].long()]
        return out

    def mlp(inp, blk):
        h1 = torch.relu(inp @ p[f"{blk}.0.weight"].t() + p[f"{blk}.0.bias"] + rows(tabs[f"cond.{blk}"]))
        h2 = torch.relu(torch.addmm(p[f"{blk}.2.bias"], h1, p[f"{blk}.2.weight"].t()))
        return h1, h2, torch.addmm(p[f"{blk}.4.bias"], h2, p[f"{blk}.4.weight"].t())

    h1m, h2m, mu = mlp(x, "encoder_mu")
    h1s, h2s, lv = mlp(x, "encoder_sigma")
    sigma = torch.exp(0.5 * lv)
    z = mu + sigma * eps
    d1, d2, y = mlp(z, "decoder")
    return dict(h1m=h1m, h2m=h2m, mu=mu, h1s=h1s, h2s=h2s, logvar=lv, sigma=sigma, eps=eps, z=z,
                d1=d1, d2=d2, y=y, x_hat=y)


def backward_cond(p, tabs, x, acts, g_xhat, g_mu_direct, g_lv_direct, gender, age):
    """backward of forward_cond: the 18 plain grads + d/d(table[j]) = sum_{b: label_b = j} dY1[b]."""
    grads, extra = backward(p, x, acts, g_xhat, g_mu_direct, g_lv_direct)
    # recompute first-layer pre-activation grads
    def g_first(g_y, h1, h2, blk):
        g_h2 = (g_y @ p[f"{blk}.4.weight"]) * (h2 > 0)
        return (g_h2 @ p[f"{blk}.2.weight"]) * (h1 > 0)

    g1 = {
        "decoder": g_first(g_xhat, acts["d1"], acts["d2"], "decoder"),
        "encoder_mu": g_first(extra["g_mu"], acts["h1m"], acts["h2m"], "encoder_mu"),
        "encoder_sigma": g_first(extra["g_lv"], acts["h1s"], acts["h2s"], "encoder_sigma"),
    }
    tgrads = {}
    for blk, g in g1.items():
        t = torch.zeros(COND_DIM, g.shape[1])
        for lab, base, n in ((gender, 0, N_GENDER), (age, N_GENDER, N_AGE)):
            for j in range(n):
                t[base + j] = g[lab == j].sum(0)
        tgrads[f"cond.{blk}"] = t
    return grads, tgrads, extra


# --------------------------------------------------------------------------------------
# Philox4x32-10 + Box-Muller: the epsilon stream of the HIP path (build-defined; the reference
# draws from torch's global generator, model.py:57, which cannot be matched -- parity tests inject eps)
# --------------------------------------------------------------------------------------
_PH_M0, _PH_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PH_W0, _PH_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(ctr: np.ndarray, key: np.ndarray) -> np.ndarray:
    """ctr [n,4] uint32, key [2] uint32 -> [n,4] uint32 (Salmon et al. 2011, 10 rounds)."""
    c = ctr.astype(np.uint32).copy()
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _PH_M0 * c[:, 0].astype(np.uint64)
            p1 = _PH_M1 * c[:, 2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & mask).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & mask).astype(np.uint32)
            c = np.stack([hi1 ^ c[:, 1] ^ k0, lo1, hi0 ^ c[:, 3] ^ k1, lo0], axis=1)
            k0 = np.uint32(k0 + _PH_W0)
            k1 = np.uint32(k1 + _PH_W1)
    return c


def philox_normal(n: int, seed: int, offset: int) -> np.ndarray:
    """n standard normals, fp32.  Element e lives in Philox block (offset + e//4), lane e%4:
    counter = (blk_lo, blk_hi, 0, 0), key = (seed_lo, seed_hi); u = ((r >> 8) + 0.5) * 2^-24;
    Box-Muller on (r0, r1) -> lanes 0,1 and (r2, r3) -> lanes 2,3."""
    nblk = (n + 3) // 4
    blk = np.arange(nblk, dtype=np.uint64) + np.uint64(offset)
    ctr = np.zeros((nblk, 4), dtype=np.uint32)
    ctr[:, 0] = (blk & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (blk >> np.uint64(32)).astype(np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    r = philox4x32_10(ctr, key)
    u = ((r >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    out = np.empty((nblk, 4), dtype=np.float32)
    for a, b in ((0, 1), (2, 3)):
        rad = np.sqrt(np.float32(-2.0) * np.log(u[:, a]))
        ang = np.float32(2.0 * math.pi) * u[:, b]
        out[:, a] = rad * np.cos(ang)
        out[:, b] = rad * np.sin(ang)
    return out.reshape(-1)[:n]


# --------------------------------------------------------------------------------------
# The general LatentClassifier head (latent_classifier.py:7-70): a trunk of 0 or 1 hidden Linear + activation, one output
# Linear per label (single-label heads are the one-label case).  Checker for csrc/heads_general.hip.
# --------------------------------------------------------------------------------------
_HEAD_ACTS = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid,
              "leaky_relu": lambda t: torch.nn.functional.leaky_relu(t, 0.01)}


def heads_forward(trunk, heads, z, activation: str = "relu"):
    """trunk: [(W, b)] (0 or 1 entries), heads: [(W, b)] one per label -> list of logits (latent_classifier.py:58-68)"""
    h = z
    for W, b in trunk:
        h = _HEAD_ACTS[activation](torch.addmm(b, h, W.t()))
    return [torch.addmm(b, h, W.t()) for W, b in heads]


def langevin_heads(trunk, heads, targets, activation, z0, noises, step_size: float, noise_weight: float,
                   prior_scale: float = 1.0):
    """inference.py:72-103 with log p(y|z) = sum over the targeted labels of log_softmax(logits_label)[target_label]
    (inference.py:80-84); gradient by torch autograd on the CPU.  `heads` / `targets`: the labels the target names, in
    the target dict's order.  prior_scale: N for the reference's single one-logit head (its [N,1] + [N] broadcast sums the
    prior N times, inference.py:62-64,90-92), else 1; a one-logit head contributes no gradient (log_softmax == 0)."""
    z = z0.clone()
    hist = []
    for t in range(noises.shape[0]):
        zz = z.clone().requires_grad_()
        lpy = 0.0
        for logits, tgt in zip(heads_forward(trunk, heads, zz, activation), targets):
            lp = torch.log_softmax(logits, dim=-1)
            lpy = lpy + (lp[:, tgt] if lp.shape[1] > 1 else lp[:, 0] * (1.0 if tgt == 1 else -1.0))
        total = (lpy - prior_scale * 0.5 * (zz ** 2).sum(dim=1)).sum()
        grad = torch.autograd.grad(total, zz)[0]
        z = z + 0.5 * step_size ** 2 * grad + step_size * noise_weight * noises[t]
        hist.append(z.clone())
    return z, hist


def classifier_heads_ce(trunk, heads, z, ys, activation: str = "relu"):
    """the training step's classifier term on its multi-label branch (lightning.py:85-95): the MEAN over the labels of the
    mean-over-rows cross entropy; torch autograd on the CPU.  Returns loss, per-label losses, per-label hits, d loss / d z,
    trunk gradients [(dW, db)], head gradients [(dW, db)]"""
    zz = z.clone().requires_grad_()
    tr = [(W.clone().requires_grad_(), b.clone().requires_grad_()) for W, b in trunk]
    hd = [(W.clone().requires_grad_(), b.clone().requires_grad_()) for W, b in heads]
    logits = heads_forward(tr, hd, zz, activation)
    per = [torch.nn.functional.cross_entropy(l, y.long()) for l, y in zip(logits, ys)]
    loss = sum(per) / len(per)
    loss.backward()
    hits = [int((l.argmax(1) == y).sum()) for l, y in zip(logits, ys)]
    return (loss.detach(), [q.detach() for q in per], hits, zz.grad, [(W.grad, b.grad) for W, b in tr],
            [(W.grad, b.grad) for W, b in hd])


def train_step_with_label_heads(p, trunk, heads, x, ys, eps, beta: float = 1.0, cw: float = 1.0, activation: str = "relu",
                                masks=None):
    """lightning.py:67-131 with a (multi-label) LatentClassifier on mu: total = recon + beta kl + cw * mean_label CE.
    Returns dict(total, recon, kl, ce, per_label, hits), the 18 VAE gradients, trunk and head gradients (times cw)."""
    acts = forward(p, x, eps, False)
    recon, kl, total = elbo(acts["x_hat"], x, acts["mu"], acts["logvar"], beta, False)
    ce, per, hits, g_mu_cls, tg, hg = classifier_heads_ce(trunk, heads, acts["mu"], ys, activation)
    g_xh, g_mu_d, g_lv_d = elbo_grads(acts["x_hat"], x, acts["mu"], acts["logvar"], beta, False, 1.0)
    grads, _ = backward(p, x, acts, g_xh, g_mu_d + cw * g_mu_cls, g_lv_d, False, masks=masks)
    return (dict(total=total + cw * ce, recon=recon, kl=kl, ce=ce, per_label=per, hits=hits, acts=acts), grads,
            [(cw * gW, cw * gb) for gW, gb in tg], [(cw * gW, cw * gb) for gW, gb in hg])


# --------------------------------------------------------------------------------------
# Adam + CosineAnnealingLR  (lightning.py:204-214; torch.optim.Adam defaults)
# --------------------------------------------------------------------------------------
def adam_step(param, grad, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8,
              weight_decay=0.0):
    """One torch.optim.Adam update (non-amsgrad, L2 weight decay); `step` is 1-based."""
    if weight_decay:
        grad = grad + weight_decay * param
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return param - (lr / bc1) * m / denom, m, v


def cosine_lr(base_lr: float, epoch: int, T_max: int, eta_min: float = 0.0) -> float:
    """Closed form of torch CosineAnnealingLR (lightning.py:206), stepped per epoch (210-211)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / T_max)) / 2


# --------------------------------------------------------------------------------------
# LatentClassifier forward (latent_classifier.py:58-70, single-label ReLU MLP) and the
# Langevin conditional sampler (inference.py:72-105) with injected noise
# --------------------------------------------------------------------------------------
def classifier_forward(layers, z):
    """layers: list of (W[out,in], b[out]); ReLU between Linear layers (latent_classifier.py:30-40)."""
    h = z
    for i, (W, b) in enumerate(layers):
        h = torch.addmm(b, h, W.t())
        if i + 1 < len(layers):
            h = torch.relu(h)
    return h


def classifier_ce(layers, z, y):
    """mean cross entropy of classifier(z) against labels y (lightning.py:73-79) with its gradients by torch autograd
    on the CPU: returns loss, number of argmax hits, d loss/d z and [(dW, db), ...]"""
    zz = z.clone().requires_grad_()
    ps = [(W.clone().requires_grad_(), b.clone().requires_grad_()) for W, b in layers]
    logits = classifier_forward(ps, zz)
    loss = torch.nn.functional.cross_entropy(logits, y.long())
    loss.backward()
    hits = int((logits.argmax(1) == y).sum())
    return loss.detach(), hits, zz.grad, [(W.grad, b.grad) for W, b in ps]


def train_step_with_classifier(p, layers, x, y, eps, beta: float = 1.0, cw: float = 1.0, masks=None):
    """lightning.py:67-131 with a single-label LatentClassifier head on mu: total = recon + beta*kl + cw*CE.
    Returns dict(total, recon, kl, ce, hits), the 18 VAE gradients and [(dW, db), ...] of the classifier."""
    acts = forward(p, x, eps, False)
    recon, kl, total = elbo(acts["x_hat"], x, acts["mu"], acts["logvar"], beta, False)
    ce, hits, g_mu_cls, cgrads = classifier_ce(layers, acts["mu"], y)
    g_xh, g_mu_d, g_lv_d = elbo_grads(acts["x_hat"], x, acts["mu"], acts["logvar"], beta, False, 1.0)
    grads, _ = backward(p, x, acts, g_xh, g_mu_d + cw * g_mu_cls, g_lv_d, False, masks=masks)
    return (dict(total=total + cw * ce, recon=recon, kl=kl, ce=ce, hits=hits, acts=acts), grads,
            [(cw * gW, cw * gb) for gW, gb in cgrads])


def train_step_with_heads(p, layers, cons_layers, x, y, eps, beta: float = 1.0, cw: float = 1.0, sw: float = 1.0):
    """lightning.py:67-131 with BOTH heads: a LatentClassifier on mu (lightning.py:73-98) and a frozen
    EmbeddingClassifier (embedding_classifier.py:28-31,49-62: Linear-ReLU-Linear-ReLU-Linear) on x_hat
    (lightning.py:100-108): total = recon + beta*kl + cw*CE(classifier(mu), y) + sw*CE(consistency(x_hat), y).
    `layers` may be None (no latent head).  Returns dict(total, recon, kl, ce, hits, cons, cons_hits), the 18 VAE
    gradients and [(dW, db), ...] of the latent classifier (the consistency head gets no gradient: frozen)."""
    acts = forward(p, x, eps, False)
    recon, kl, total = elbo(acts["x_hat"], x, acts["mu"], acts["logvar"], beta, False)
    g_xh, g_mu, g_lv = elbo_grads(acts["x_hat"], x, acts["mu"], acts["logvar"], beta, False, 1.0)
    out = dict(recon=recon, kl=kl)
    cgrads = []
    if layers is not None:
        ce, hits, g_mu_cls, cg = classifier_ce(layers, acts["mu"], y)
        g_mu = g_mu + cw * g_mu_cls
        total = total + cw * ce
        cgrads = [(cw * gW, cw * gb) for gW, gb in cg]
        out.update(ce=ce, hits=hits)
    cons, cons_hits, g_xh_cons, _ = classifier_ce(cons_layers, acts["x_hat"], y)
    g_xh = g_xh + sw * g_xh_cons
    total = total + sw * cons
    out.update(total=total, cons=cons, cons_hits=cons_hits)
    grads, _ = backward(p, x, acts, g_xh, g_mu, g_lv, False)
    return out, grads, cgrads


def langevin(layers, z0, noises, target: int, step_size: float, noise_weight: float):
    """inference.py:77-101 for a multiclass single-label classifier, closed-form gradient:
    grad = d/dz [log_softmax(logits)[target] - 0.5 |z|^2];  z += 0.5 s^2 grad + s w noise."""
    z = z0.clone()
    hist = []
    for t in range(noises.shape[0]):
        hs = [z]
        h = z
        for i, (W, b) in enumerate(layers):
            h = torch.addmm(b, h, W.t())
            if i + 1 < len(layers):
                h = torch.relu(h)
            hs.append(h)
        logits = hs[-1]
        g = -torch.softmax(logits, dim=-1)
        g[:, target] += 1.0                       # d log_softmax[target] / d logits
        for i in range(len(layers) - 1, -1, -1):
            W = layers[i][0]
            g = g @ W
            if i > 0:
                g = g * (hs[i] > 0)
        grad = g - z                              # + d(-0.5|z|^2)/dz   (inference.py:90)
        z = z + 0.5 * step_size ** 2 * grad + step_size * noise_weight * noises[t]  # inference.py:97
        hist.append(z.clone())
    return z, hist


# --------------------------------------------------------------------------------------
# bf16 path (BASELINE config 3): bf16 MFMA operands, fp32 accumulation, fp32 master weights.
# The reference has no bf16 code: this emulation DEFINES the rounding points of the HIP bf16 path
# (every GEMM operand is rounded to bf16 once, where the producing kernel stores it; accumulators,
# biases, mu / logvar / x_hat, the loss and all parameter gradients stay fp32).  Parity unpinned by the
# reference; pinned to the fp32 oracle only through the bf16 tolerance in tests/test_gpu_bf16.py.
# --------------------------------------------------------------------------------------
def bf(t: torch.Tensor) -> torch.Tensor:
    return t.bfloat16().float()


def decode_bf16(p, z, normalize_decoder: bool = False):
    """bf16 emulation of VAEModel.decode (model.py:65-69): bf16 operands (z, weights, hidden activations), fp32
    accumulation, biases and output -- the rounding points of psvae_decode_bf16"""
    W = {k: bf(v) for k, v in p.items() if k.startswith("decoder") and k.endswith("weight")}
    d1 = bf(torch.relu(bf(z) @ W["decoder.0.weight"].t() + p["decoder.0.bias"]))
    d2 = bf(torch.relu(d1 @ W["decoder.2.weight"].t() + p["decoder.2.bias"]))
    y = d2 @ W["decoder.4.weight"].t() + p["decoder.4.bias"]
    return normalize_rows(y)[0] if normalize_decoder else y


def train_step_bf16(p, x, eps, beta: float = 1.0, go: float = 1.0, tabs=None, gender=None, age=None,
                    normalize_decoder: bool = False, use_cos_loss: bool = False):
    """bf16 emulation of one training pass; with the loss variants (model.py:60-61 F.normalize on the decoder output,
    lightning.py:110-111 cosine loss) the decoder's raw fp32 output goes through them in fp32 and the gradient w.r.t. the raw
    output is rounded to bf16 where the MSE gradient is in the default pass"""
    B, D = x.shape
    W = {k: bf(v) for k, v in p.items() if k.endswith("weight")}
    xb = bf(x)

    def rows(blk, n_out):
        out = torch.zeros(B, n_out)
        if tabs is not None:
            tab = tabs[f"cond.{blk}"]
            for lab, base, n in ((gender, 0, N_GENDER), (age, N_GENDER, N_AGE)):
                ok = (lab >= 0) & (lab < n)
                out[ok] += tab[base + lab[ok].long()]
        return out

    def enc(blk):
        h1 = bf(torch.relu(xb @ W[f"{blk}.0.weight"].t() + p[f"{blk}.0.bias"] + rows(blk, HIDDEN)))
        h2 = bf(torch.relu(h1 @ W[f"{blk}.2.weight"].t() + p[f"{blk}.2.bias"]))
        return h1, h2, h2 @ W[f"{blk}.4.weight"].t() + p[f"{blk}.4.bias"]

    h1m, h2m, mu = enc("encoder_mu")
    h1s, h2s, lv = enc("encoder_sigma")
    z = mu + torch.exp(0.5 * lv) * eps
    zb = bf(z)
    d1 = bf(torch.relu(zb @ W["decoder.0.weight"].t() + p["decoder.0.bias"] + rows("decoder", HIDDEN)))
    d2 = bf(torch.relu(d1 @ W["decoder.2.weight"].t() + p["decoder.2.bias"]))
    xh = d2 @ W["decoder.4.weight"].t() + p["decoder.4.bias"]
    if normalize_decoder or use_cos_loss:
        y = xh.clone().requires_grad_(True)
        xh = torch.nn.functional.normalize(y, p=2, dim=1) if normalize_decoder else y
        recon, kl, total = elbo(xh, x, mu, lv, beta, use_cos_loss)
        (go * recon).backward()
        gy = bf(y.grad)
        xh, recon, total = xh.detach(), recon.detach(), total.detach()
    else:
        recon, kl, total = elbo(xh, x, mu, lv, beta)
        gy = bf(go * 2.0 * (xh - x) / (10.0 * B * D))
    g = {}

    def back(gyb, inp, a1, a2, blk, need_dx):
        g[f"{blk}.4.weight"] = gyb.t() @ a2
        g[f"{blk}.4.bias"] = gyb.sum(0)
        da2 = bf((gyb @ W[f"{blk}.4.weight"]) * (a2 > 0))
        g[f"{blk}.2.weight"] = da2.t() @ a1
        g[f"{blk}.2.bias"] = da2.sum(0)
        da1 = bf((da2 @ W[f"{blk}.2.weight"]) * (a1 > 0))
        g[f"{blk}.0.weight"] = da1.t() @ inp
        g[f"{blk}.0.bias"] = da1.sum(0)
        return da1, ((da1 @ W[f"{blk}.0.weight"]) if need_dx else None)

    dd1, gz = back(gy, zb, d1, d2, "decoder", True)
    gmu = bf(gz + go * beta * mu / B)
    glv = bf(gz * eps * 0.5 * torch.exp(0.5 * lv) + go * beta * (lv.exp() - 1.0) / (2.0 * B))
    dh1m, _ = back(gmu, xb, h1m, h2m, "encoder_mu", False)
    dh1s, _ = back(glv, xb, h1s, h2s, "encoder_sigma", False)
    tg = None
    if tabs is not None:
        tg = {}
        for blk, dy in (("encoder_mu", dh1m), ("encoder_sigma", dh1s), ("decoder", dd1)):
            t = torch.zeros(COND_DIM, HIDDEN)
            for lab, base, n in ((gender, 0, N_GENDER), (age, N_GENDER, N_AGE)):
                for j in range(n):
                    t[base + j] = dy[lab == j].sum(0)
            tg[f"cond.{blk}"] = t
    return dict(recon=recon, kl=kl, total=total), {k: g[k] for k in PARAM_KEYS}, dict(x_hat=xh, mu=mu, logvar=lv, z=z), tg
