"""EmbeddingClassifier: the reference's 3-layer MLP on speaker embeddings
(/root/reference/ps_vae/embedding_classifier/embedding_classifier.py:6-99), which PseudoSpeakerVAE loads frozen as
its "consistency" head and applies to the reconstruction x_hat (/root/reference/ps_vae/lightning.py:44-52,100-108).

Same constructor arguments, layer names (`fc1`, `fc2`, `fc3`) and therefore the same state_dict keys, the same
training_step / validation_step / configure_optimizers.  SURVEY.md section 8f ranks it "next" (N4).  `forward` is the
module's plain Linear chain (logits for callers that want them); the training-step use -- cross entropy of the head
on a batch plus the gradient flowing back into x_hat -- is ONE fused autograd node on the GPU,
`cross_entropy_step` (psvae_consistency_ce_f32: fc1 on the grouped fp32 MFMA kernel, fc2 + fc3 + softmax cross
entropy and the way back per row in one kernel, d x_hat and the optional weight gradients as one grouped GEMM launch).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _ffi
from .lightning import _Base, _accuracy


def consistency_ce_raw(x_hat, y, params, grad_scale: float = 1.0, want_param_grads: bool = False):
    """psvae_consistency_ce_f32 on detached tensors: loss2 = {mean CE, hits}, g_xhat and (optionally) the six parameter
    gradients, all times grad_scale"""
    x_hat = x_hat.detach().float().contiguous()
    if not x_hat.is_cuda:
        raise _ffi.PsvaeError("the consistency step runs on the GPU (cuda:N); there is no CPU path")
    y = y.to(device=x_hat.device, dtype=torch.int64).contiguous()
    B, D = x_hat.shape
    ps = [p.detach().float().contiguous() for p in params]
    hidden, C = ps[0].shape[0], ps[4].shape[0]
    from .latent_classifier import label_watch
    label_watch.before(y, C, "EmbeddingClassifier (consistency) cross entropy")
    lib = _ffi.lib()
    scratch = torch.empty(int(lib.psvae_consistency_scratch_bytes(B, hidden, C)), dtype=torch.uint8, device=x_hat.device)
    loss2 = torch.empty(2, dtype=torch.float32, device=x_hat.device)
    g_x = torch.empty_like(x_hat)
    grads, pg = None, None
    if want_param_grads:
        grads = [torch.empty_like(p) for p in ps]
        pg = _ffi.Mlp3Grads()
        for i, g in enumerate(grads):
            pg.g[i] = _ffi.ptr(g)
    _ffi.check(lib.psvae_consistency_ce_f32(
        B, D, hidden, C, _ffi.ptr(x_hat), _ffi.ptr(y), *[_ffi.ptr(p) for p in ps], float(grad_scale), _ffi.ptr(loss2),
        _ffi.ptr(g_x), pg, _ffi.ptr(scratch), _ffi.stream_ptr()), "consistency_ce")
    label_watch.after(loss2[0], y, C, "EmbeddingClassifier (consistency) cross entropy")
    return loss2, g_x, grads


class _ConsistencyCEFunction(torch.autograd.Function):
    """(mean CE, hits) = CE(head(x_hat), y): forward runs the fused step with grad_scale = 1 and keeps the gradients;
    backward hands them out scaled by the upstream d(loss) on the device (no host sync)."""

    @staticmethod
    def forward(ctx, x_hat, y, *params):
        want = any(p.requires_grad for p in params)
        loss2, g_x, grads = consistency_ce_raw(x_hat, y, params, 1.0, want)
        ctx.g_x, ctx.grads = g_x, grads
        loss, hits = loss2[0], loss2[1]
        ctx.mark_non_differentiable(hits)
        return loss, hits

    @staticmethod
    def backward(ctx, g_loss, _g_hits):
        s = g_loss.float()
        g_x = ctx.g_x.mul_(s) if ctx.needs_input_grad[0] else None
        if ctx.grads is None:
            grads = [None] * 6
        else:
            grads = [g.mul_(s) if need else None for g, need in zip(ctx.grads, ctx.needs_input_grad[2:])]
        ctx.g_x = ctx.grads = None
        return (g_x, None, *grads)


class EmbeddingClassifier(_Base):
    def __init__(self, input_dim: int, num_classes: int, hidden_dim: int = 128, optimizer_cfg: dict = {}) -> None:
        super().__init__()
        self.save_hyperparameters()                                        # embedding_classifier.py:21
        self.input_dim = input_dim
        self.num_classes = num_classes
        self.hidden_dim = hidden_dim
        self.optimizer_cfg = optimizer_cfg
        self.fc1 = nn.Linear(self.input_dim, self.hidden_dim)              # embedding_classifier.py:29-31
        self.fc2 = nn.Linear(self.hidden_dim, self.hidden_dim)
        self.fc3 = nn.Linear(self.hidden_dim, self.num_classes)
        self.relu = nn.ReLU()
        self.softmax = nn.Softmax(dim=1)
        self.loss_fn = nn.CrossEntropyLoss()
        self.accuracy = _accuracy(task="multiclass", num_classes=self.num_classes)

    def configure_optimizers(self):                                        # embedding_classifier.py:39-47
        import os
        if os.environ.get("PSVAE_FUSED_ADAM", "1") != "0":                  # a torch.optim.Adam either way (optim.py)
            from .optim import FusedAdam
            return FusedAdam(self.parameters(), **self.optimizer_cfg)
        return torch.optim.Adam(self.parameters(), **self.optimizer_cfg)

    def forward(self, x: torch.Tensor) -> torch.Tensor:                    # embedding_classifier.py:49-62
        from . import ops
        if ops.mlp_forward_ok(x, [self.fc1, self.fc2, self.fc3]):           # inference on the GPU: three MFMA GEMM launches
            return ops.mlp_forward(x, [(self.fc1.weight, self.fc1.bias, True), (self.fc2.weight, self.fc2.bias, True),
                                       (self.fc3.weight, self.fc3.bias, False)])
        x = self.relu(self.fc1(x))
        x = self.relu(self.fc2(x))
        return self.fc3(x)

    def fused_ok(self) -> bool:
        """whether psvae_consistency_ce_f32 covers this head: hidden in {16, 32, 64, 128}, C <= 8, input dim % 4 == 0"""
        return self.hidden_dim in (16, 32, 64, 128) and self.num_classes <= 8 and self.input_dim % 4 == 0

    def fused_params(self):
        return [self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self.fc3.weight, self.fc3.bias]

    def cross_entropy_step(self, x: torch.Tensor, y: torch.Tensor):
        """(mean cross entropy, number of correct rows) of this head on embeddings x[B, D] against labels y[B] in
        [0, C) as ONE fused autograd node on the GPU (lightning.py:101-102 / embedding_classifier.py:75-77)."""
        if not self.fused_ok() or not x.is_cuda:
            raise _ffi.PsvaeError("cross_entropy_step needs hidden_dim in {16,32,64,128}, num_classes <= 8, "
                                  "input_dim % 4 == 0 and GPU tensors; use forward() + loss_fn for the others")
        return _ConsistencyCEFunction.apply(x, y, *self.fused_params())

    def _step(self, batch, prefix: str):
        x, y = batch
        if self.fused_ok() and x.is_cuda:
            loss, hits = self.cross_entropy_step(x, y)
            acc = hits / x.size(0)
        else:
            y_hat = self(x)
            loss = self.loss_fn(y_hat, y)
            acc = self.accuracy(y_hat, y)
        self.log(f"{prefix}_acc", acc, sync_dist=True)
        self.log(f"{prefix}_loss", loss, sync_dist=True)
        return loss

    def training_step(self, batch, batch_idx: int) -> torch.Tensor:        # embedding_classifier.py:64-82
        return self._step(batch, "train")

    def validation_step(self, batch, batch_idx: int) -> torch.Tensor:      # embedding_classifier.py:84-99
        return self._step(batch, "val")
