"""ConditionalVAEModel: the one-hot conditional VAE of BASELINE config 3 (gender + age one-hot concat).

The reference has NO such model: this is a build-defined extension
whose only reference-pinned part is the label -> index maps (labels.py <- ps_vae/utils.py:82-136).
Definition: c = [onehot3(gender) || onehot3(age)], encoder input [x || c], decoder input [z || c]
(an unknown label, -1, gives an all-zero block).  A one-hot block times a weight matrix is a row gather,
so the concat is never materialised: each first layer gets a [6, 512] table whose rows are added to the
bias per sample inside the GEMM epilogue, and the tables' gradients are label-segmented column sums.
`to_concat_state_dict` / `from_concat_state_dict` convert to the [512, D+6] / [512, Z+6] weights of the
literal concat formulation (checked against it in the tests).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _ffi
from .model import ELBOStepFunction, HIDDEN, VAEModel

COND_DIM = _ffi.COND_DIM
_BLOCKS = ("encoder_mu", "encoder_sigma", "decoder")


class ConditionalVAEModel(VAEModel):
    def __init__(self, input_dim: int = 256, latent_dim: int = 64, normalize_decoder: bool = False):
        if normalize_decoder:
            raise ValueError("the conditional extension does not combine with normalize_decoder")
        super().__init__(input_dim, latent_dim, False)
        bound = 1.0 / (input_dim + COND_DIM) ** 0.5  # nn.Linear's U(+-1/sqrt(fan_in)) for the extra columns
        self.cond = nn.ParameterDict({
            blk: nn.Parameter(torch.empty(COND_DIM, HIDDEN).uniform_(-bound, bound)) for blk in _BLOCKS})

    def cond_tables(self):
        return [self.cond[b] for b in _BLOCKS]

    def elbo_step(self, x, gender, age, beta: float = 1.0, eps: Optional[torch.Tensor] = None):
        """fused forward + ELBO; returns (total, recon, kl, x_hat, mu, logvar); labels are int tensors [B]"""
        gender = gender.to(device=x.device, dtype=torch.int32).contiguous()
        age = age.to(device=x.device, dtype=torch.int32).contiguous()
        seed, off = self._next_rng(x.shape[0] * self.latent_dim)
        meta = (seed, off, 0, self._scratch_cache, float(beta), (None, gender, age))
        return ELBOStepFunction.apply(x, eps, meta, *self.ordered_params(), *self.cond_tables())

    def forward(self, x, gender=None, age=None, eps=None):
        if gender is None or age is None:
            raise _ffi.PsvaeError("ConditionalVAEModel.forward needs gender and age label tensors")
        total, recon, kl, xh, mu, lv = self.elbo_step(x, gender, age, eps=eps)
        return xh, mu, lv

    def decode(self, z, gender=None, age=None):
        if gender is None or age is None:
            raise _ffi.PsvaeError("ConditionalVAEModel.decode needs gender and age label tensors")
        from . import ops
        dec = [p.detach() for p in self.ordered_params()[12:]]
        return ops.decode(z.detach().float().contiguous(), dec, self.input_dim, 0,
                          cond_tab=self.cond["decoder"].detach(),
                          gender=gender.to(device=z.device, dtype=torch.int32).contiguous(),
                          age=age.to(device=z.device, dtype=torch.int32).contiguous())

    # ---- literal-concat parameterisation -----------------------------------------------------------
    def to_concat_state_dict(self) -> Dict[str, torch.Tensor]:
        sd = {k: v.detach().clone() for k, v in self.state_dict().items() if not k.startswith("cond.")}
        for blk in _BLOCKS:
            sd[f"{blk}.0.weight"] = torch.cat([sd[f"{blk}.0.weight"], self.cond[blk].detach().t()], dim=1)
        return sd

    def from_concat_state_dict(self, sd: Dict[str, torch.Tensor]):
        own = {}
        for k, v in sd.items():
            own[k] = v
        for blk, n_in in zip(_BLOCKS, (self.input_dim, self.input_dim, self.latent_dim)):
            W = sd[f"{blk}.0.weight"]
            own[f"{blk}.0.weight"] = W[:, :n_in].contiguous()
            own[f"cond.{blk}"] = W[:, n_in:].t().contiguous()
        self.load_state_dict(own)
