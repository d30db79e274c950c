"""Samplers of the reference's inference script (/root/reference/ps_vae/inference.py:10-110).

unconditional_synthesis: z ~ N(0, I) drawn ON THE DEVICE from the Philox stream, decoded by the HIP
decode path.  conditional_synthesis: the reference's unadjusted Langevin loop on
log p(y|z) + log p(z) with the same update (inference.py:97), kept on the device with no per-step
host synchronisation (the reference copies z to the host and calls .item() twice per step)."""
from __future__ import annotations

from typing import Optional, Union

import torch
import torch.nn.functional as F

from . import ops


def _latent_dim(vae_model) -> int:
    return vae_model.hparams.model["latent_dim"]


def unconditional_synthesis(vae_model, num_samples: int, device: str, seed: Optional[int] = None) -> torch.Tensor:
    """inference.py:10-27."""
    dim = _latent_dim(vae_model)
    seed = torch.initial_seed() if seed is None else seed
    z = ops.philox_normal(num_samples * dim, seed & ((1 << 64) - 1), 0, device).view(num_samples, dim)
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


def conditional_synthesis(vae_model, num_samples: int, classifier_target: Union[int, dict], step_size: float = 0.01,
                          num_steps: int = 100, noise_weight: float = 1.0, return_history: bool = False,
                          device: str = "cuda", z0: Optional[torch.Tensor] = None, noises=None):
    """inference.py:29-110.  `z0` / `noises` (optional) inject the initial point and the per-step
    noise (parity tests); by default both come from torch's generator on `device`."""
    dim = _latent_dim(vae_model)
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
