"""Samplers of the reference's inference script (/root/reference/ps_vae/inference.py:10-110).

unconditional_synthesis: z ~ N(0, I) drawn ON THE DEVICE from the Philox stream, decoded by the HIP
decode path.  conditional_synthesis: the reference's unadjusted Langevin loop on
log p(y|z) + log p(z) with the same update (inference.py:97), kept on the device with no per-step
host synchronisation (the reference copies z to the host and calls .item() twice per step)."""
from __future__ import annotations

from typing import Optional, Union

import torch
import torch.nn.functional as F

import torch.nn as nn

from . import _ffi, ops


def _latent_dim(vae_model) -> int:
    return vae_model.hparams.model["latent_dim"]


# The samplers draw from their OWN Philox stream: key = torch's seed XOR a constant (so the draws never coincide with
# the epsilon stream of VAEModel._next_rng, which uses the plain seed), and a starting block taken FROM TORCH'S DEFAULT
# GENERATOR (one 62-bit draw per call).  So, as with the reference's torch.randn (inference.py:23,73,95): every call
# advances torch's generator and returns fresh samples, and torch.manual_seed() makes the sequence of calls repeatable.
_SAMPLER_KEY_XOR = 0x5A17D00DC0FFEE11


def _take_blocks(nblk: int, seed: Optional[int] = None):
    """(key, first block) of this call's draw.  An explicit `seed` gives a private, repeatable stream that starts at
    block 0 and leaves torch's generator alone."""
    if seed is not None:
        return (int(seed) ^ _SAMPLER_KEY_XOR) & ((1 << 64) - 1), 0
    key = (torch.initial_seed() ^ _SAMPLER_KEY_XOR) & ((1 << 64) - 1)
    first = int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64).item())  # 2^62 blocks apart w.h.p.: no overlap
    return key, first


def unconditional_synthesis(vae_model, num_samples: int, device: str, seed: Optional[int] = None) -> torch.Tensor:
    """inference.py:10-27."""
    dim = _latent_dim(vae_model)
    key, off = _take_blocks((num_samples * dim + 3) // 4, seed)
    model = getattr(vae_model, "model", vae_model)
    if torch.device(device).type == "cuda" and hasattr(model, "ordered_params"):
        # draw + decode as ONE call: on sampling-sized batches z is drawn inside the fused decode kernel (same Philox
        # stream, same values as philox_normal followed by decode; no draw launch, no round trip of z through memory)
        dec = [p.detach() for p in model.ordered_params()[12:]]
        if dec[0].is_cuda:
            xh, _ = ops.sample_decode(num_samples, dim, dec, model.input_dim, key, off, model.flags)
            return xh.cpu()
    z = ops.philox_normal(num_samples * dim, key, off, device).view(num_samples, dim)
    with torch.no_grad():
        return vae_model.decode(z).detach().cpu()


def _log_p_y_given_z(logits, target):
    log_probs = F.log_softmax(logits, dim=-1)
    if log_probs.shape[1] == 1:  # inference.py:60-66 (degenerate branch kept as written)
        if target == 1:
            return log_probs
        if target == 0:
            return 1 - log_probs
        raise ValueError("classifier_target must be 0 or 1 for binary classifier")
    return log_probs[:, target]


def _fusable_classifier(clf, target):
    """(W1, b1, W2, b2, hidden, C) if the classifier is what the fused kernel covers: single label, ReLU,
    one or two Linear layers (latent_classifier.py:30-40), hidden <= 128, C <= 8; else None."""
    if clf is None or not getattr(clf, "single_label_mode", False) or not isinstance(target, int):
        return None
    mods = list(clf.layers)
    lins = [m for m in mods if isinstance(m, nn.Linear)]
    acts = [m for m in mods if not isinstance(m, nn.Linear)]
    if len(lins) not in (1, 2) or any(not isinstance(a, nn.ReLU) for a in acts):
        return None
    C = lins[-1].out_features
    hidden = lins[0].out_features if len(lins) == 2 else 0
    if C > 8 or hidden > 128 or lins[0].in_features not in (16, 32, 64, 128) or not (0 <= target < C):
        return None
    if C == 1:  # the one-logit head has the reference's N-fold prior (see _general_classifier): the general kernel's case
        return None
    w = [p.detach().float().contiguous() for l in lins for p in (l.weight, l.bias)]
    return (w[0], w[1], w[2] if hidden else None, w[3] if hidden else None, hidden, C)


def langevin_fused(clf_params, z: torch.Tensor, target: int, step_size: float, noise_weight: float, num_steps: int,
                   noises=None, seed: int = 0, offset: int = 0, return_history: bool = False):
    """psvae_langevin_f32: all `num_steps` updates of z[N,Z] in one kernel, no host round trips"""
    W1, b1, W2, b2, hidden, C = clf_params
    z = z.detach().float().contiguous().clone()
    N, Z = z.shape
    hist = torch.empty(num_steps, N, Z, dtype=torch.float32, device=z.device) if return_history else None
    nz = None if noises is None else noises.to(device=z.device, dtype=torch.float32).contiguous()
    _ffi.check(_ffi.lib().psvae_langevin_f32(N, Z, hidden, C, _ffi.ptr(W1), _ffi.ptr(b1), _ffi.ptr(W2), _ffi.ptr(b2),
                                             _ffi.ptr(z), int(target), float(step_size), float(noise_weight),
                                             int(num_steps), _ffi.ptr(nz), seed, offset, _ffi.ptr(hist),
                                             _ffi.stream_ptr()), "langevin")
    return z, hist


def _general_classifier(clf, target, num_samples: int = 1):
    """(hidden, activation, sizes, targets, [W1, b1, W2, b2], prior_scale) for the general fused Langevin kernel (csrc/heads_general.hip):
    several labels (dict target: the labels named in it, inference.py:80-84), any of the four activations, the one-logit
    head, up to 32 classes; None when the head has more than one hidden layer (the torch-op loop below serves those)."""
    spec = clf.general_spec() if clf is not None and hasattr(clf, "general_spec") else None
    if spec is None:
        return None
    hidden, activation, names, sizes, trunk, heads = spec
    if names == [None]:
        if not isinstance(target, int):
            return None
        pick, targets = [0], [target]
    else:
        if not isinstance(target, dict) or any(k not in names for k in target):
            return None
        pick = [names.index(k) for k in target]
        targets = [int(target[k]) for k in target]
    sizes = [sizes[i] for i in pick]
    prior_scale = 1.0
    for t, c in zip(targets, sizes):
        if c == 1:
            # The reference's binary branch (inference.py:60-66): log-softmax over ONE logit is identically 0, so the head
            # has no gradient; its [N, 1] result is then added to the [N] prior term (inference.py:92), which broadcasts
            # to [N, N] and sums the prior N times -- kept as written (quirks are not fixed silently).
            # Inside a dict target the reference's in-place add raises on those shapes: the torch-op loop is left to do so.
            if names != [None]:
                return None
            if t not in (0, 1):
                raise ValueError("classifier_target must be 0 or 1 for binary classifier")  # inference.py:66
            prior_scale = float(num_samples)
        elif not 0 <= t < c:
            return None
    targets = [0 if c == 1 else t for t, c in zip(targets, sizes)]
    Wc = torch.cat([heads[i][0].detach().float() for i in pick]).contiguous()
    bc = torch.cat([heads[i][1].detach().float() for i in pick]).contiguous()
    if hidden:
        w = [trunk[0].detach().float().contiguous(), trunk[1].detach().float().contiguous(), Wc, bc]
    else:
        w = [Wc, bc, None, None]
    return hidden, activation, sizes, targets, w, prior_scale


def langevin_general(spec, z: torch.Tensor, step_size: float, noise_weight: float, num_steps: int, noises=None, seed: int = 0,
                     offset: int = 0, return_history: bool = False):
    """psvae_langevin_heads_f32: the whole chain in one kernel for any head _general_classifier accepts"""
    import ctypes as C
    hidden, activation, sizes, targets, (W1, b1, W2, b2), prior_scale = spec
    z = z.detach().float().contiguous().clone()
    N, Z = z.shape
    hist = torch.empty(num_steps, N, Z, dtype=torch.float32, device=z.device) if return_history else None
    nz = None if noises is None else noises.to(device=z.device, dtype=torch.float32).contiguous()
    heads = _ffi.make_heads(sizes, targets, activation, prior_scale)
    _ffi.check(_ffi.lib().psvae_langevin_heads_f32(N, Z, hidden, C.byref(heads), _ffi.ptr(W1), _ffi.ptr(b1), _ffi.ptr(W2),
                                                   _ffi.ptr(b2), _ffi.ptr(z), float(step_size), float(noise_weight),
                                                   int(num_steps), _ffi.ptr(nz), seed, offset, _ffi.ptr(hist),
                                                   _ffi.stream_ptr()), "langevin_heads")
    return z, hist


def conditional_synthesis(vae_model, num_samples: int, classifier_target: Union[int, dict], step_size: float = 0.01,
                          num_steps: int = 100, noise_weight: float = 1.0, return_history: bool = False,
                          device: str = "cpu", z0: Optional[torch.Tensor] = None, noises=None):
    """inference.py:29-110, same signature and defaults (`device="cpu"`, inference.py:37).  There is no CPU path:
    a non-GPU `device` raises, naming what to pass.  `z0` / `noises` (optional) inject the initial point and the
    per-step noise (parity tests); by default both come from torch's generator on `device`."""
    if torch.device(device).type != "cuda":
        raise RuntimeError(f"conditional_synthesis: GPU only (MI355X HIP path, no CPU fallback): got device={device!r}; "
                           "pass device='cuda' (inference.py --device cuda) with the model moved there")
    dim = _latent_dim(vae_model)
    fus = _fusable_classifier(getattr(vae_model, "classifier", None), classifier_target)
    gen = None if fus is not None else _general_classifier(getattr(vae_model, "classifier", None), classifier_target, num_samples)
    if gen is not None and dim in (16, 32, 64, 128):
        # general fused path: multi-label / non-ReLU / one-logit / up to 32 classes, still ONE kernel for the whole chain
        nblk = (num_samples * dim + 3) // 4
        key, off = _take_blocks((1 + num_steps) * nblk)
        z_init = ops.philox_normal(num_samples * dim, key, off, device).view(num_samples, dim) if z0 is None else z0.to(device)
        z, hist = langevin_general(gen, z_init, step_size, noise_weight, num_steps, noises, seed=key, offset=off + nblk,
                                   return_history=return_history)
        with torch.no_grad():
            x_hat = vae_model.decode(z).detach().cpu()
        if return_history:
            return x_hat, list(hist.cpu().numpy())
        return x_hat
    if fus is not None and torch.device(device).type == "cuda":
        # fused HIP path: z0 and the per-step noise from the samplers' device Philox stream (follows torch.manual_seed;
        # every call consumes (1 + num_steps) * nblk fresh blocks)
        nblk = (num_samples * dim + 3) // 4
        key, off = _take_blocks((1 + num_steps) * nblk)
        z_init = ops.philox_normal(num_samples * dim, key, off, device).view(num_samples, dim) if z0 is None else z0.to(device)
        z, hist = langevin_fused(fus, z_init, classifier_target, step_size, noise_weight, num_steps, noises,
                                 seed=key, offset=off + nblk, return_history=return_history)
        with torch.no_grad():
            x_hat = vae_model.decode(z).detach().cpu()
        if return_history:
            return x_hat, list(hist.cpu().numpy())
        return x_hat
    # what is left for torch ops: heads with more than one hidden layer (or > 32 classes, hidden > 128): the reference's loop
    # on the device, still without per-step host synchronisation
    z = (torch.randn((num_samples, dim), device=device) if z0 is None else z0.to(device)).requires_grad_()
    history = []
    for t in range(num_steps):
        logits = vae_model.classifier(z)
        if isinstance(logits, dict):
            assert isinstance(classifier_target, dict), "classifier_target must be a dict for multi-label classifier"
            lpy = 0
            for label, target in classifier_target.items():
                lpy = lpy + _log_p_y_given_z(logits[label], target)
        else:
            lpy = _log_p_y_given_z(logits, classifier_target)
        log_p_z = -0.5 * (z ** 2).sum(dim=1)                       # inference.py:90
        grad = torch.autograd.grad((lpy + log_p_z).sum(), z)[0]    # inference.py:93
        noise = torch.randn_like(z) if noises is None else noises[t].to(device)
        z = (z + 0.5 * (step_size ** 2) * grad + step_size * noise_weight * noise).detach()  # inference.py:97
        if return_history:
            history.append(z.clone())                              # stays on the device
        z.requires_grad_()
    with torch.no_grad():
        x_hat = vae_model.decode(z.detach()).detach().cpu()
    if return_history:
        return x_hat, [h.cpu().numpy() for h in history]
    return x_hat
