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
_ACT_NAME = {nn.ReLU: "relu", nn.Tanh: "tanh", nn.Sigmoid: "sigmoid", nn.LeakyReLU: "leaky_relu"}


def classifier_heads_ce_raw(z, ys, hidden, activation, sizes, trunk, heads, grad_scale: float = 1.0):
    """psvae_classifier_heads_ce_f32 on detached tensors (any head LatentClassifier.general_spec covers): trunk = [W1, b1]
    or [] (no hidden layer), heads = [(W, b) per label]; ys = the labels' targets, one [B] tensor per label.  Returns
    loss_out = {mean over labels of the mean CE, hits per label, mean CE per label}, g_z, the trunk's gradients and the
    heads' gradients [(gW, gb) per label], all times grad_scale."""
    import ctypes as C
    z = z.detach().float().contiguous()
    B, Z = z.shape
    n = len(sizes)
    y = torch.stack([t.to(device=z.device, dtype=torch.int64) for t in ys]).contiguous()
    Wc = torch.cat([W.detach().float() for W, _ in heads]).contiguous()     # the labels' output Linears, stacked
    bc = torch.cat([b.detach().float() for _, b in heads]).contiguous()
    Ct = Wc.shape[0]
    for k, (t, c) in enumerate(zip(ys, sizes)):
        label_watch.before(t, c, f"LatentClassifier cross entropy (label {k})")
    lib = _ffi.lib()
    spec = _ffi.make_heads(sizes, None, activation)
    scratch = torch.empty(int(lib.psvae_classifier_scratch_bytes(B, hidden, Ct)), dtype=torch.uint8, device=z.device)
    loss_out = torch.empty(1 + 2 * n, dtype=torch.float32, device=z.device)
    g_z = torch.empty_like(z)
    if hidden:
        W1, b1 = (t.detach().float().contiguous() for t in trunk)
        gW1, gb1, gW2, gb2 = torch.empty_like(W1), torch.empty_like(b1), torch.empty_like(Wc), torch.empty_like(bc)
        args = (_ffi.ptr(W1), _ffi.ptr(b1), _ffi.ptr(Wc), _ffi.ptr(bc))
        gargs = (_ffi.ptr(gW1), _ffi.ptr(gb1), _ffi.ptr(gW2), _ffi.ptr(gb2))
    else:
        gW1, gb1 = torch.empty_like(Wc), torch.empty_like(bc)
        gW2 = gb2 = None
        args = (_ffi.ptr(Wc), _ffi.ptr(bc), None, None)
        gargs = (_ffi.ptr(gW1), _ffi.ptr(gb1), None, None)
    _ffi.check(lib.psvae_classifier_heads_ce_f32(B, Z, hidden, C.byref(spec), _ffi.ptr(z), _ffi.ptr(y), *args,
                                                 float(grad_scale), _ffi.ptr(loss_out), _ffi.ptr(g_z), *gargs,
                                                 _ffi.ptr(scratch), _ffi.stream_ptr()), "classifier_heads_ce")
    label_watch.after(loss_out[0], y, max(sizes), "LatentClassifier cross entropy (multi-label)")
    gWc, gbc = (gW2, gb2) if hidden else (gW1, gb1)
    head_grads = list(zip(gWc.split(list(sizes)), gbc.split(list(sizes))))
    return loss_out, g_z, ([gW1, gb1] if hidden else []), head_grads


class _ClassifierHeadsCEFunction(torch.autograd.Function):
    """loss_out = CE step of a general head (several labels / any of the four activations / up to 32 classes) as ONE
    autograd node: forward runs the fused step with grad_scale = 1, backward hands the gradients out scaled by the
    upstream d(loss_out[0]) on the device.  params = trunk (0 or 2 tensors) then (W, b) per label."""

    @staticmethod
    def forward(ctx, z, ys, meta, *params):
        hidden, activation, sizes = meta
        nt = 2 if hidden else 0
        trunk, flat = params[:nt], params[nt:]
        heads = [(flat[2 * i], flat[2 * i + 1]) for i in range(len(sizes))]
        loss_out, g_z, tg, hg = classifier_heads_ce_raw(z, ys, hidden, activation, sizes, trunk, heads)
        ctx.g_z = g_z
        ctx.grads = list(tg) + [t for pair in hg for t in pair]
        return loss_out

    @staticmethod
    def backward(ctx, g_out):
        s = g_out[0].float()  # only the mean loss is differentiable (hits and per-label losses are reports)
        g_z = ctx.g_z.mul_(s) if ctx.needs_input_grad[0] else None
        grads = [g.contiguous().mul_(s) for g in ctx.grads]
        ctx.g_z = ctx.grads = None
        return (g_z, None, None, *grads)



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

    def general_spec(self):
        """(hidden, activation, label names, sizes, trunk params, [(W, b) per label]) when the GENERAL fused kernels
        (csrc/heads_general.hip) cover this head: one or two Linear layers, any of the four activations, one label or
        several, at most 32 classes in all and 8 labels, hidden <= 128, input dim in {16, 32, 64, 128}; else None"""
        lins = [m for m in self.layers if isinstance(m, nn.Linear)]
        acts = [m for m in self.layers if not isinstance(m, nn.Linear)]
        if self.single_label_mode:
            if len(lins) not in (1, 2):
                return None
            trunk, heads, names = lins[:-1], [lins[-1]], [None]
        else:
            if len(lins) > 1:
                return None
            trunk, names = lins, list(self.output_layers.keys())
            heads = [self.output_layers[k] for k in names]
        if any(type(a) not in _ACT_NAME for a in acts) or len({type(a) for a in acts}) > 1:
            return None
        if acts and isinstance(acts[0], nn.LeakyReLU) and acts[0].negative_slope != 0.01:
            return None
        activation = _ACT_NAME[type(acts[0])] if acts else "relu"
        hidden = trunk[0].out_features if trunk else 0
        zin = (trunk[0] if trunk else heads[0]).in_features
        sizes = [h.out_features for h in heads]
        if sum(sizes) > 32 or len(sizes) > 8 or hidden > 128 or zin not in (16, 32, 64, 128):
            return None
        tp = [trunk[0].weight, trunk[0].bias] if trunk else []
        return hidden, activation, names, sizes, tp, [(h.weight, h.bias) for h in heads]

    def cross_entropy_heads_step(self, z: torch.Tensor, y):
        """The training step's classifier term for ANY head general_spec covers, as one fused autograd node
        (lightning.py:73-98): y a [B] tensor (single label) or a dict label -> [B] (multi-label, lightning.py:85-95: the
        MEAN of the labels' cross entropies).  Returns (loss, {label: hits}, {label: that label's mean CE})."""
        spec = self.general_spec()
        if spec is None or not z.is_cuda:
            raise _ffi.PsvaeError("cross_entropy_heads_step: a head LatentClassifier.general_spec() covers, on the GPU")
        hidden, activation, names, sizes, trunk, heads = spec
        ys = [y] if names == [None] else [y[k] for k in names]
        out = _ClassifierHeadsCEFunction.apply(z, ys, (hidden, activation, tuple(sizes)), *trunk, *[t for p in heads for t in p])
        n = len(sizes)
        return out[0], {k: out[1 + i] for i, k in enumerate(names)}, {k: out[1 + n + i] for i, k in enumerate(names)}

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
        fast = self._hip_forward(x)
        if fast is not None:
            return fast
        for layer in self.layers:
            x = layer(x)
        if self.single_label_mode:
            return x
        return {label: head(x) for label, head in self.output_layers.items()}

    def _hip_forward(self, x):
        """inference (gradients off, fp32 GPU tensors, ReLU stacks): the layers as MFMA GEMM launches with fused bias /
        bias + ReLU epilogues (ops.mlp_forward) instead of ATen; anything else returns None and forward() runs as written"""
        from . import ops
        mods = list(self.layers)
        lins = [m for m in mods if isinstance(m, nn.Linear)]
        heads = [] if self.single_label_mode else list(self.output_layers.values())
        if any(not isinstance(m, (nn.Linear, nn.ReLU)) for m in mods) or not ops.mlp_forward_ok(x, lins + heads):
            return None
        stack, i = [], 0
        while i < len(mods):  # (Linear [, ReLU]) pairs, in order
            if not isinstance(mods[i], nn.Linear):
                return None
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            stack.append((mods[i].weight, mods[i].bias, relu))
            i += 2 if relu else 1
        h = ops.mlp_forward(x, stack) if stack else x
        if self.single_label_mode:
            return h
        return {label: ops.mlp_forward(h, [(head.weight, head.bias, False)]) for label, head in self.output_layers.items()}
