"""LatentClassifier: the reference's MLP head on latent vectors (/root/reference/ps_vae/latent_classifier.py:7-70).

Same constructor arguments, module layout (`layers` ModuleList with activations interleaved,
`output_layers` ModuleDict) and therefore the same state_dict keys.  It is tiny (64 -> C); SURVEY.md
section 8f ranks it "next" (N1).  `forward` is plain torch (any layer count, activation, multi-head); the head the
reference actually trains with -- single label, one or two Linear layers with ReLU -- also has a fused HIP
training step, `LatentClassifier.cross_entropy_step` (psvae_classifier_ce_f32: per-row kernel + one grouped GEMM
for the weight gradients), which `PseudoSpeakerVAE.training_step` uses when it applies.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _ffi


class LabelWatch:
    """Out-of-range labels without a host sync per step.

    The reference's F.cross_entropy raises on a label outside [0, C) (lightning.py:79 is fed the raw label maps, whose
    "unknown" is -1: utils.py:103,120,136) -- on a GPU as an asynchronous device-side assert.  The fused kernels turn such
    a row's loss term into NaN, so the step's loss IS the device-side error flag.  This watch folds every step's loss
    into a sticky accumulator on the device (torch.maximum propagates NaN: one tiny launch, no sync) and reads it back
    every `every` steps (`PSVAE_CHECK_LABELS_EVERY`, default 50) -- and on the very first step, where a wrongly mapped
    dataset shows at once.  A poisoned accumulator raises: with the offending labels when the current batch has them,
    else naming the window.  `PSVAE_CHECK_LABELS=sync` checks every batch before its launch (a device reduction and a
    host read per step: the round-2 behaviour); `PSVAE_CHECK_LABELS=0` switches the watch off."""

    def __init__(self):
        import os
        self.mode = os.environ.get("PSVAE_CHECK_LABELS", "1")
        self.every = max(1, int(os.environ.get("PSVAE_CHECK_LABELS_EVERY", "50")))
        self.acc = {}      # (device, head) -> sticky accumulator
        self.calls = {}

    @staticmethod
    def _raise_if_bad(y, C, what):
        bad_rows = (y < 0) | (y >= C)
        if bool(bad_rows.any()):
            raise _ffi.PsvaeError(f"{what}: label out of range [0, {C}): {y[bad_rows][:4].tolist()} "
                                  "(map unknown labels out of the batch first)")

    def before(self, y: torch.Tensor, C: int, what: str) -> None:
        if self.mode == "sync" or (self.mode != "0" and self.calls.get((y.device, what), 0) == 0):
            self._raise_if_bad(y, C, what)

    def after(self, loss: torch.Tensor, y: torch.Tensor, C: int, what: str) -> None:
        """`loss`: a device tensor that is NaN iff a row of this step had a label outside [0, C) (or the step diverged)"""
        if self.mode in ("0", "sync"):
            return
        dev = (y.device, what)
        acc = self.acc.get(dev)
        if acc is None:
            acc = self.acc[dev] = torch.zeros((), dtype=torch.float32, device=y.device)
        torch.maximum(acc, loss.detach().reshape(()), out=acc)
        n = self.calls[dev] = self.calls.get(dev, 0) + 1
        if n % self.every == 0 and bool(torch.isnan(acc)):
            acc.zero_()
            self._raise_if_bad(y, C, what)
            raise _ffi.PsvaeError(f"{what}: the loss became NaN within the last {self.every} steps: a label outside "
                                  f"[0, {C}) was fed (map unknown labels out of the batch first), or training diverged")


label_watch = LabelWatch()


def classifier_ce_raw(z, y, hidden, params, grad_scale: float = 1.0):
    """psvae_classifier_ce_f32 on detached tensors: returns loss2 = {mean CE, hits}, g_z and the parameter gradients
    (all times grad_scale)"""
    z = z.detach().float().contiguous()
    y = y.to(device=z.device, dtype=torch.int64).contiguous()
    B, Z = z.shape
    ps = [p.detach().float().contiguous() for p in params]
    W1, b1 = ps[0], ps[1]
    W2, b2 = (ps[2], ps[3]) if hidden else (None, None)
    C = (W2 if hidden else W1).shape[0]
    label_watch.before(y, C, "LatentClassifier cross entropy")
    lib = _ffi.lib()
    scratch = torch.empty(int(lib.psvae_classifier_scratch_bytes(B, hidden, C)), dtype=torch.uint8, device=z.device)
    loss2 = torch.empty(2, dtype=torch.float32, device=z.device)
    g_z = torch.empty_like(z)
    grads = [torch.empty_like(p) for p in ps]
    _ffi.check(lib.psvae_classifier_ce_f32(
        B, Z, hidden, C, _ffi.ptr(z), _ffi.ptr(y), _ffi.ptr(W1), _ffi.ptr(b1), _ffi.ptr(W2), _ffi.ptr(b2),
        float(grad_scale), _ffi.ptr(loss2), _ffi.ptr(g_z), _ffi.ptr(grads[0]), _ffi.ptr(grads[1]),
        _ffi.ptr(grads[2]) if hidden else None, _ffi.ptr(grads[3]) if hidden else None, _ffi.ptr(scratch),
        _ffi.stream_ptr()), "classifier_ce")
    label_watch.after(loss2[0], y, C, "LatentClassifier cross entropy")
    return loss2, g_z, grads


class _ClassifierCEFunction(torch.autograd.Function):
    """(mean CE, hits) = CE(classifier(z), y): forward runs the fused step with grad_scale = 1 and keeps the
    gradients; backward hands them out scaled by the upstream d(loss) on the device."""

    @staticmethod
    def forward(ctx, z, y, hidden, *params):
        loss2, g_z, grads = classifier_ce_raw(z, y, hidden, params)
        ctx.g_z, ctx.grads = g_z, grads
        loss, hits = loss2[0], loss2[1]
        ctx.mark_non_differentiable(hits)
        return loss, hits

    @staticmethod
    def backward(ctx, g_loss, _g_hits):
        s = g_loss.float()
        g_z = ctx.g_z.mul_(s) if ctx.needs_input_grad[0] else None
        grads = [g.mul_(s) for g in ctx.grads]
        ctx.g_z = ctx.grads = None
        return (g_z, None, None, *grads)

_ACTIVATIONS = {"relu": nn.ReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "leaky_relu": nn.LeakyReLU}


class LatentClassifier(nn.Module):
    def __init__(self, input_dim: int, num_classes, num_layers: int = 1, hidden_dim: int = 128,
                 activation: str = "relu"):
        super().__init__()
        if activation not in _ACTIVATIONS:
            raise ValueError(f"Unsupported activation: {activation}")
        act = _ACTIVATIONS[activation]
        self.single_label_mode = isinstance(num_classes, int)
        self.num_classes = num_classes
        self.layers = nn.ModuleList()
        self.output_layers = nn.ModuleDict()
        heads = None if self.single_label_mode else dict(num_classes)
        if num_layers == 1:
            if self.single_label_mode:
                self.layers.append(nn.Linear(input_dim, num_classes))
            else:
                for label, n in heads.items():
                    self.output_layers[label] = nn.Linear(input_dim, n)
            return
        self.layers.append(nn.Linear(input_dim, hidden_dim))
        for _ in range(num_layers - 2):
            self.layers.append(act())
            self.layers.append(nn.Linear(hidden_dim, hidden_dim))
        self.layers.append(act())
        if self.single_label_mode:
            self.layers.append(nn.Linear(hidden_dim, num_classes))
        else:
            for label, n in heads.items():
                self.output_layers[label] = nn.Linear(hidden_dim, n)

    def fused_params(self, input_dim=None):
        """(hidden, [W1, b1(, W2, b2)]) when the fused HIP step covers this head: single label, one or two Linear
        layers with ReLU between them, hidden <= 128, C <= 8, input dim in {16, 32, 64, 128}; else None"""
        if not self.single_label_mode:
            return None
        mods = list(self.layers)
        lins = [m for m in mods if isinstance(m, nn.Linear)]
        acts = [m for m in mods if not isinstance(m, nn.Linear)]
        if len(lins) not in (1, 2) or any(not isinstance(a, nn.ReLU) for a in acts):
            return None
        hidden = lins[0].out_features if len(lins) == 2 else 0
        if lins[-1].out_features > 8 or hidden > 128 or lins[0].in_features not in (16, 32, 64, 128):
            return None
        return hidden, [p for l in lins for p in (l.weight, l.bias)]

    def cross_entropy_step(self, z: torch.Tensor, y: torch.Tensor):
        """(mean cross entropy, number of correct rows) of this head on latent rows z[B, Z] against labels y[B] in
        [0, C), as ONE fused autograd node on the GPU (lightning.py:73-98 with F.cross_entropy's default mean)."""
        fp = self.fused_params()
        if fp is None or not z.is_cuda:
            raise _ffi.PsvaeError("cross_entropy_step covers single-label heads with one or two Linear layers and ReLU, "
                                  "on the GPU; use forward() + F.cross_entropy for the others")
        hidden, params = fp
        return _ClassifierCEFunction.apply(z, y, hidden, *params)

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        if self.single_label_mode:
            return x
        return {label: head(x) for label, head in self.output_layers.items()}
