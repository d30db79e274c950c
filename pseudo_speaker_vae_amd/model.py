"""VAEModel: the reference's nn.Module API (/root/reference/ps_vae/model.py:7-69) over the HIP hot path.

Same constructor, same `forward(x) -> (x_hat, mu, log_sigma)` / `decode(z)`, same 18 state_dict keys
and shapes, same default-init RNG consumption (the nn.Sequential containers are kept as parameter
holders only; their forward is never used).  All arithmetic runs in libpsvae_hip.so through two
torch.autograd.Functions; inputs must live on the GPU -- there is no CPU fallback.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed
import torch.nn as nn

from . import _ffi, api_graph, ops

HIDDEN = _ffi.HIDDEN


class _Buffers:
    """Per-(shape, device) cache of scratch + ctypes structs; activations are allocated per call
    (one flat tensor) because several forwards may be alive before their backward."""

    def __init__(self):
        self._scratch: Dict[tuple, torch.Tensor] = {}
        self.keep_last = False   # tests: keep the most recent call's context alive (its saved activations)
        self._last = None
        self.graphs = api_graph.GraphSteps()   # captured fused steps for the shapes training_step keeps seeing

    def remember(self, c):
        if self.keep_last:
            self._last = c

    def last_ctx(self):
        return self._last

    def scratch(self, B, D, Z, device) -> torch.Tensor:
        key = (B, D, Z, str(device))
        t = self._scratch.get(key)
        if t is None:
            if len(self._scratch) > 8:
                self._scratch.clear()
            t = torch.empty(_ffi.scratch_bytes(B, D, Z), dtype=torch.uint8, device=device)
            self._scratch[key] = t
        return t


_ACT_ORDER = ("h1m", "h2m", "h1s", "h2s", "d1", "d2", "eps", "z")


def _alloc_acts(B: int, D: int, Z: int, device, normalize: bool):
    """one allocation: 6 x [B,512] + 2 x [B,Z] (+ [B]) activations and the 3 outputs."""
    sizes = [B * HIDDEN] * 6 + [B * Z] * 2 + [B * D, B * Z, B * Z] + ([B] if normalize else [])
    pad = [(s + 63) // 64 * 64 for s in sizes]  # 256-B aligned carve
    flat = torch.empty(sum(pad), dtype=torch.float32, device=device)
    views, o = [], 0
    for s, p in zip(sizes, pad):
        views.append(flat[o:o + s])
        o += p
    acts = {k: v.view(B, -1) for k, v in zip(_ACT_ORDER, views[:8])}
    xh, mu, lv = views[8].view(B, D), views[9].view(B, Z), views[10].view(B, Z)
    if normalize:
        acts["ynorm"] = views[11]
    return acts, xh, mu, lv


def _alloc_grads(tensors):
    """one fresh tensor per gradient: autograd's AccumulateGrad takes such a tensor over as `.grad` without a copy
    (views of a flat buffer are copied, one small launch each)"""
    return [torch.empty_like(t) for t in tensors]


def _scale_grads(grads, s):
    """grads *= s (a device scalar), one multi-tensor launch"""
    torch._foreach_mul_(grads, s)


def _alloc_like_flat(tensors):
    """gradients for `tensors` carved out of ONE flat buffer (a single all-reduce payload)."""
    sizes = [t.numel() for t in tensors]
    pad = [(s + 63) // 64 * 64 for s in sizes]
    flat = torch.empty(sum(pad), dtype=torch.float32, device=tensors[0].device)
    out, o = [], 0
    for t, s, p in zip(tensors, sizes, pad):
        out.append(flat[o:o + s].view_as(t))
        o += p
    return out


_FUSED_AUTOGRAD = os.environ.get("PSVAE_AUTOGRAD_FUSED", "1") != "0"  # tuning switch


class _Ctx:
    """what backward needs, shared by both Functions"""
    __slots__ = ("B", "D", "Z", "flags", "acts", "acts_c", "scratch", "dims", "cond_tensors")


def _run_forward(x, eps, seed, offset, flags, params, buffers: _Buffers, cond_tensors=None):
    if not x.is_cuda:
        raise _ffi.PsvaeError("VAEModel runs on the GPU only (HIP kernels); move the module and its input to cuda")
    x = x.contiguous()
    if x.dtype != torch.float32:
        raise _ffi.PsvaeError(f"expected float32 input, got {x.dtype}")
    B, D = x.shape
    Z = params[4].shape[0]
    if params[0].shape[1] != D:
        raise _ffi.PsvaeError(f"input has {D} features, the model expects {params[0].shape[1]}")
    normalize = bool(flags & _ffi.NORMALIZE_DECODER)
    c = _Ctx()
    c.B, c.D, c.Z, c.flags = B, D, Z, flags
    c.acts, xh, mu, lv = _alloc_acts(B, D, Z, x.device, normalize)
    c.acts_c = _ffi.make_acts(c.acts)
    c.scratch = buffers.scratch(B, D, Z, x.device)
    c.dims = _ffi.Dims(B, D, Z)
    c.cond_tensors = cond_tensors
    ws = _WsView(c)
    cond = _make_cond(cond_tensors, None)
    buffers.remember(c)
    ops.vae_forward(ws, x, params, eps, seed, offset, flags, cond=cond, outs=(xh, mu, lv))
    return c, x, xh, mu, lv


def _run_fused_step(x, eps, seed, offset, beta, params, buffers: _Buffers, flags: int = 0, cond_tensors=None):
    """psvae_train_step_f32 on fresh buffers: returns ctx, outputs, the loss triple and the gradients (d total = 1):
    the 18 parameter gradients, then the three conditional-table gradients when cond_tensors is given"""
    if not x.is_cuda:
        raise _ffi.PsvaeError("VAEModel runs on the GPU only (HIP kernels); move the module and its input to cuda")
    x = x.contiguous()
    if x.dtype != torch.float32:
        raise _ffi.PsvaeError(f"expected float32 input, got {x.dtype}")
    B, D = x.shape
    Z = params[4].shape[0]
    if params[0].shape[1] != D:
        raise _ffi.PsvaeError(f"input has {D} features, the model expects {params[0].shape[1]}")
    c = _Ctx()
    c.B, c.D, c.Z, c.flags = B, D, Z, flags
    c.acts, xh, mu, lv = _alloc_acts(B, D, Z, x.device, bool(flags & _ffi.NORMALIZE_DECODER))
    c.acts_c = _ffi.make_acts(c.acts)
    c.scratch = buffers.scratch(B, D, Z, x.device)
    c.dims = _ffi.Dims(B, D, Z)
    c.cond_tensors = cond_tensors
    grads = _alloc_grads(params)
    gtabs = _alloc_grads(cond_tensors[0]) if cond_tensors is not None else []
    cond = _make_cond(cond_tensors, gtabs) if cond_tensors is not None else None
    loss3 = torch.empty(3, dtype=torch.float32, device=x.device)
    buffers.remember(c)
    ops.train_step(_WsView(c), x, params, grads, (xh, mu, lv), loss3, beta=beta, eps=eps, seed=seed, offset=offset,
                   cond=cond, flags=flags)
    return c, x, xh, mu, lv, loss3, grads + gtabs


def _make_cond(cond_tensors, gtabs):
    if cond_tensors is None:
        return None
    tabs, gender, age = cond_tensors
    return _ffi.make_cond(tabs, gtabs, gender, age)


class _WsView:
    """adapts a _Ctx to the ops.Workspace duck type"""

    def __init__(self, c: _Ctx):
        self.B, self.D, self.Z = c.B, c.D, c.Z
        self.dims, self.acts, self.acts_c, self.scratch = c.dims, c.acts, c.acts_c, c.scratch


class VAEForwardFunction(torch.autograd.Function):
    """(x_hat, mu, logvar) = VAE(x); differentiable w.r.t. the 18 parameters (and x if it asks)."""

    @staticmethod
    def forward(ctx, x, eps, meta, *params):
        seed, offset, flags, buffers = meta
        params = [p.detach() for p in params]
        c, x, xh, mu, lv = _run_forward(x.detach(), eps, seed, offset, flags, params, buffers)
        ctx.c, ctx.params, ctx.x = c, params, x
        ctx.save_for_backward(xh, mu, lv)
        ctx.set_materialize_grads(False)
        return xh, mu, lv

    @staticmethod
    def backward(ctx, g_xh, g_mu, g_lv):
        xh, mu, lv = ctx.saved_tensors
        c = ctx.c
        grads = _alloc_like_flat(ctx.params)
        g_x = torch.empty_like(ctx.x) if ctx.needs_input_grad[0] else None
        cg = lambda g: None if g is None else g.contiguous()
        if g_xh is None and not (c.flags & _ffi.NORMALIZE_DECODER):
            g_xh = torch.zeros_like(xh)  # decoder still needs a defined upstream gradient
        ops.vae_backward(_WsView(c), ctx.x, ctx.params, xh, mu, lv, grads, g_xhat=cg(g_xh), g_mu=cg(g_mu),
                         g_logvar=cg(g_lv), elbo_grad=False, g_x=g_x, flags=c.flags)
        return (g_x, None, None, *grads)


class ELBOStepFunction(torch.autograd.Function):
    """total, recon, kl = ELBO(VAE(x), x): forward + loss in one Function; backward runs the fused
    gradient path with the upstream d/d(total) read on the device (no host sync).  Returns
    (total, recon, kl, x_hat, mu, logvar); only `total` is differentiable."""

    @staticmethod
    def forward(ctx, x, eps, meta, *params):
        seed, offset, flags, buffers, beta, cond_tensors = meta[:6]
        world, want_outputs = meta[6:8] if len(meta) > 6 else (1, True)
        live_params = params
        params = [p.detach() for p in params]
        ncond = 3 if cond_tensors is not None else 0
        vae_params = params[:_ffi.NPARAM]
        if ncond:
            cond_tensors = (params[_ffi.NPARAM:], cond_tensors[1], cond_tensors[2])
        ctx.fused_grads = None
        ctx.slot = None
        if (_FUSED_AUTOGRAD and eps is None and not ncond and all(ctx.needs_input_grad[3:]) and x.is_cuda
                and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == params[0].shape[1]):
            # steady state of a training loop: the whole fused step as ONE captured HIP graph on static buffers
            # (api_graph.py); returns None whenever the call is not one it may serve, and the eager path below runs
            slot = buffers.graphs.lookup(x, params[4].shape[0], flags, beta, live_params, world)
            if slot is not None:
                token = object()
                loss3 = slot.run(x.detach(), seed, offset, token)
                ctx.slot, ctx.token, ctx.guard = slot, token, api_graph._Pending(slot, token)
                buffers.remember(slot)
                total, recon, kl = loss3[0], loss3[1], loss3[2]
                outs = tuple(t.clone() for t in slot.outs) if want_outputs else (None, None, None)
                ctx.mark_non_differentiable(recon, kl, *(t for t in outs if t is not None))
                return (total, recon, kl, *outs)
        if any(ctx.needs_input_grad[3:]) and _FUSED_AUTOGRAD:
            # parameters want gradients: run the fused train step now (forward + ELBO + backward with d(total) = 1; every
            # loss variant, with or without the conditional tables) and hand the gradients out, scaled, in backward()
            c, x, xh, mu, lv, loss3, grads = _run_fused_step(x.detach(), eps, seed, offset, beta, vae_params, buffers,
                                                             flags, cond_tensors if ncond else None)
            ctx.fused_grads = grads
        else:
            c, x, xh, mu, lv = _run_forward(x.detach(), eps, seed, offset, flags, vae_params, buffers, cond_tensors)
            loss3 = torch.empty(3, dtype=torch.float32, device=x.device)
            ops.elbo(_WsView(c), x, xh, mu, lv, beta, flags, loss3)
        ctx.c, ctx.params, ctx.x, ctx.beta = c, params, x, beta
        ctx.save_for_backward(xh, mu, lv)
        total, recon, kl = loss3[0], loss3[1], loss3[2]
        ctx.mark_non_differentiable(recon, kl, xh, mu, lv)
        return total, recon, kl, xh, mu, lv

    @staticmethod
    def backward(ctx, g_total, *_):
        if ctx.slot is not None:  # graph path: the gradients wait, unscaled, in the slot's flat buffer
            if ctx.slot.pending is not ctx.token:
                raise _ffi.PsvaeError("backward of a fused step whose gradients are gone: it ran already, or another "
                                      "step of the same shape has replaced them (PSVAE_API_GRAPH=0 keeps every step's "
                                      "gradients in buffers of its own)")
            return (None, None, None, *ctx.slot.scaled_grads(g_total.float()))
        if ctx.fused_grads is not None:  # gradients are linear in d(total): scale the precomputed ones on the device
            grads = ctx.fused_grads
            ctx.fused_grads = None
            _scale_grads(grads, g_total.float())
            return (None, None, None, *grads)
        xh, mu, lv = ctx.saved_tensors
        c = ctx.c
        grads = _alloc_like_flat(ctx.params)
        cond = _make_cond(c.cond_tensors, grads[_ffi.NPARAM:]) if c.cond_tensors is not None else None
        ops.vae_backward(_WsView(c), ctx.x, ctx.params[:_ffi.NPARAM], xh, mu, lv, grads[:_ffi.NPARAM],
                         elbo_grad=True, beta=ctx.beta, grad_loss=g_total.contiguous().float(), flags=c.flags,
                         cond=cond)
        return (None, None, None, *grads)


class ELBOClassifierStepFunction(torch.autograd.Function):
    """total = ELBO(VAE(x), x) + cw * CE(classifier(mu), y) + sw * CE(consistency(x_hat), y) as ONE autograd node, all on
    the HIP path (lightning.py:67-131 with a single-label LatentClassifier head on mu and / or an EmbeddingClassifier
    consistency head on x_hat): forward, ELBO, the heads' cross-entropy steps (whose d/dmu and d/dx_hat become the
    upstream gradients of the VAE backward) and the VAE backward run inside forward(); backward() hands the gradients
    out scaled by d(total).  Returns (total, recon, kl, ce, hits, cons, cons_hits, x_hat, mu, logvar); the entries of an
    absent head are zeros."""

    @staticmethod
    def forward(ctx, x, eps, y, meta, *params):
        from .embedding_classifier import consistency_ce_raw
        from .latent_classifier import classifier_ce_raw
        seed, offset, flags, buffers, beta, cw, hidden, n_clf, sw, n_cons = meta
        need = list(ctx.needs_input_grad[4:])
        params = [p.detach() for p in params]
        vae_params = params[:_ffi.NPARAM]
        clf_params = params[_ffi.NPARAM:_ffi.NPARAM + n_clf]
        cons_params = params[_ffi.NPARAM + n_clf:]
        c, x, xh, mu, lv = _run_forward(x.detach(), eps, seed, offset, flags, vae_params, buffers)
        loss3 = torch.empty(3, dtype=torch.float32, device=x.device)
        ops.elbo(_WsView(c), x, xh, mu, lv, beta, flags, loss3)
        z4 = torch.zeros(4, dtype=torch.float32, device=x.device)
        total, ce, hits, cons, cons_hits = loss3[0], z4[0], z4[1], z4[2], z4[3]
        g_mu = g_xh = None
        clf_grads, cons_grads = [], [None] * n_cons
        if n_clf:
            loss2, g_mu, clf_grads = classifier_ce_raw(mu, y, hidden, clf_params, grad_scale=cw)
            ce, hits = loss2[0], loss2[1]
            total = total + cw * ce
        if n_cons:
            want = any(need[_ffi.NPARAM + n_clf:])  # the head is frozen inside the VAE's step (lightning.py:48-50)
            loss2c, g_xh, cg = consistency_ce_raw(xh, y, cons_params, grad_scale=sw, want_param_grads=want)
            cons, cons_hits = loss2c[0], loss2c[1]
            total = total + sw * cons
            if want:
                cons_grads = cg
        ctx.all_grads = None
        if any(need):
            grads = [None] * _ffi.NPARAM
            if any(need[:_ffi.NPARAM]):
                grads = _alloc_grads(vae_params)
                ops.vae_backward(_WsView(c), x, vae_params, xh, mu, lv, grads, g_xhat=g_xh, g_mu=g_mu, elbo_grad=True,
                                 beta=beta, flags=flags)
            ctx.all_grads = list(grads) + list(clf_grads) + list(cons_grads)
        recon, kl = loss3[1], loss3[2]
        ctx.mark_non_differentiable(recon, kl, ce, hits, cons, cons_hits, xh, mu, lv)
        return total, recon, kl, ce, hits, cons, cons_hits, xh, mu, lv

    @staticmethod
    def backward(ctx, g_total, *_):
        grads = ctx.all_grads
        ctx.all_grads = None
        _scale_grads([g for g in grads if g is not None], g_total.float())
        return (None, None, None, None, *grads)


class ELBOTermsFunction(torch.autograd.Function):
    """(recon, kl) of lightning.py:110-117 from the forward outputs, on the HIP kernels (psvae_elbo_f32), differentiable
    w.r.t. x_hat, mu and logvar (psvae_elbo_grad_f32): lets training_step branches that combine the ELBO with other
    torch losses (multi-label heads, a frozen VAE) keep the loss itself off ATen."""

    @staticmethod
    def forward(ctx, x_hat, mu, lv, x, cos_loss, buffers):
        import ctypes as C
        if not x_hat.is_cuda:
            raise _ffi.PsvaeError("the ELBO terms run on the GPU only (HIP kernels)")
        x_hat, mu, lv, x = (t.detach().float().contiguous() for t in (x_hat, mu, lv, x))
        B, D = x_hat.shape
        Z = mu.shape[1]
        dims = _ffi.Dims(B, D, Z)
        flags = _ffi.COS_LOSS if cos_loss else 0
        loss3 = torch.empty(3, dtype=torch.float32, device=x_hat.device)
        scratch = buffers.scratch(B, D, Z, x_hat.device)
        _ffi.check(_ffi.lib().psvae_elbo_f32(C.byref(dims), _ffi.ptr(x), _ffi.ptr(x_hat), _ffi.ptr(mu), _ffi.ptr(lv), 1.0,
                                             _ffi.ptr(loss3), _ffi.ptr(scratch), flags, _ffi.stream_ptr()), "elbo")
        ctx.save_for_backward(x_hat, mu, lv, x)
        ctx.flags = flags
        return loss3[1], loss3[2]

    @staticmethod
    def backward(ctx, g_recon, g_kl):
        import ctypes as C
        x_hat, mu, lv, x = ctx.saved_tensors
        B, D = x_hat.shape
        dims = _ffi.Dims(B, D, mu.shape[1])
        zero = lambda g: torch.zeros((), dtype=torch.float32, device=x_hat.device) if g is None else g.float().contiguous()
        gr, gk = zero(g_recon), zero(g_kl)
        g_xh, g_mu, g_lv = torch.empty_like(x_hat), torch.empty_like(mu), torch.empty_like(lv)
        _ffi.check(_ffi.lib().psvae_elbo_grad_f32(C.byref(dims), _ffi.ptr(x), _ffi.ptr(x_hat), _ffi.ptr(mu), _ffi.ptr(lv),
                                                  _ffi.ptr(gr), _ffi.ptr(gk), _ffi.ptr(g_xh), _ffi.ptr(g_mu), _ffi.ptr(g_lv),
                                                  ctx.flags, _ffi.stream_ptr()), "elbo_grad")
        return g_xh, g_mu, g_lv, None, None, None


class DecodeFunction(torch.autograd.Function):
    """x_hat = decoder(z) on the HIP decode path.  The reference only ever detaches the result
    (inference.py:25,105), so the forward is always available; a backward through decode() is not
    part of the hot path and raises."""

    @staticmethod
    def forward(ctx, z, flags, *dec):
        if not z.is_cuda:
            raise _ffi.PsvaeError("VAEModel.decode runs on the GPU only (HIP kernels)")
        dec = [p.detach() for p in dec]
        return ops.decode(z.detach().float(), dec, dec[4].shape[0], flags)

    @staticmethod
    def backward(ctx, g):  # pragma: no cover
        raise _ffi.PsvaeError("backward through VAEModel.decode() is not implemented (the reference detaches it); "
                              "use forward()/elbo_step() for training")


class VAEModel(nn.Module):
    """Drop-in for ps_vae.model.VAEModel (model.py:7-69)."""

    def __init__(self, input_dim: int = 512, latent_dim: int = 64, normalize_decoder: bool = False):
        super().__init__()
        self.normalize_decoder = normalize_decoder
        # parameter containers only (identical key names / shapes / init order: model.py:14-36)
        def mlp(i, o):
            return nn.Sequential(nn.Linear(i, HIDDEN), nn.ReLU(), nn.Linear(HIDDEN, HIDDEN), nn.ReLU(),
                                 nn.Linear(HIDDEN, o))
        self.encoder_mu = mlp(input_dim, latent_dim)
        self.encoder_sigma = mlp(input_dim, latent_dim)
        self.decoder = mlp(latent_dim, input_dim)
        self.input_dim, self.latent_dim = input_dim, latent_dim
        self._scratch_cache = _Buffers()
        self._philox_offset = 0  # Philox blocks consumed so far (advances every forward)

    # ---- helpers -------------------------------------------------------------------------
    def ordered_params(self) -> List[torch.Tensor]:
        """the 18 tensors in C-ABI order (== creation order == state_dict order).  Cached: 18 container lookups per
        step cost ~40 us of host time; `.to()` and load_state_dict keep the Parameter objects, and the first and last
        entries are re-checked against the live modules on every call."""
        cached = self.__dict__.get("_ordered_cache")
        if cached is not None and cached[0] is self.encoder_mu[0].weight and cached[17] is self.decoder[4].bias:
            return cached
        out = []
        for blk in (self.encoder_mu, self.encoder_sigma, self.decoder):
            for i in (0, 2, 4):
                out += [blk[i].weight, blk[i].bias]
        self.__dict__["_ordered_cache"] = out
        return out

    @property
    def flags(self) -> int:
        return _ffi.NORMALIZE_DECODER if self.normalize_decoder else 0

    def _rank_world(self) -> Tuple[int, int]:
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_rank(), torch.distributed.get_world_size()
        return 0, 1

    def _next_rng(self, n_elems: int) -> Tuple[int, int]:
        """(seed, offset) of this call's epsilon stream: seed follows torch.manual_seed, the offset
        advances by the Philox blocks the call consumes, so successive forwards never reuse noise."""
        seed = torch.initial_seed() & ((1 << 64) - 1)
        nblk = (n_elems + 3) // 4
        # data-parallel ranks are usually seeded alike (Lightning's seed_everything): interleave their counter
        # ranges by rank so that shards never share noise (rank r takes the r-th block range of each step)
        rank, world = self._rank_world()
        off = self._philox_offset + rank * nblk
        self._philox_offset += world * nblk
        return seed, off

    # ---- reference API -------------------------------------------------------------------
    def forward(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None):
        """model.py:38-63.  Returns (x_hat, mu, log_sigma) where log_sigma is the log-variance.
        `eps` (optional, extension) injects the N(0,1) draw instead of the Philox stream."""
        seed, off = self._next_rng(x.shape[0] * self.latent_dim)
        meta = (seed, off, self.flags, self._scratch_cache)
        return VAEForwardFunction.apply(x, eps, meta, *self.ordered_params())

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """model.py:65-69."""
        return DecodeFunction.apply(z, self.flags, *self.ordered_params()[12:])

    def tune(self, batch_size: int, iters: int = 10) -> str:
        """Optional, once per batch size: autotune the GEMM launch plan for (batch_size, input_dim, latent_dim)
        (`psvae_tune_train_step_f32`, a few seconds on scratch buffers).  `forward` / `elbo_step` / their backward
        then launch with the tuned plan.  Returns the plan's text form."""
        from . import ops
        p0 = self.encoder_mu[0].weight
        ws = ops.Workspace(batch_size, self.input_dim, self.latent_dim, p0.device)
        params = [t.detach() for t in self.ordered_params()]
        grads = [torch.empty_like(t) for t in params]
        x = torch.randn(batch_size, self.input_dim, device=p0.device)
        loss3 = torch.empty(3, device=p0.device)
        self._scratch_cache.graphs.invalidate()   # captured steps hold the plan they were captured under
        return ops.tune_train_step(ws, x, params, grads, ws.new_outputs(), loss3, iters=iters)

    def tune_from_committed_plan(self, batch_size: int) -> Optional[str]:
        """install the launch plan committed for this GPU model and (batch_size, input_dim, latent_dim), if there is one
        (pseudo_speaker_vae_amd/launch_plans.json / $PSVAE_PLAN_CACHE; no measurement); returns its text or None"""
        from types import SimpleNamespace
        shape = SimpleNamespace(B=batch_size, D=self.input_dim, Z=self.latent_dim, device=self.encoder_mu[0].weight.device,
                                dims=_ffi.Dims(batch_size, self.input_dim, self.latent_dim))
        self._scratch_cache.graphs.invalidate()
        return ops.load_committed_plan(shape)

    def elbo_heads_step(self, x: torch.Tensor, y: torch.Tensor, classifier=None, consistency=None, beta: float = 1.0,
                        classifier_weight: float = 1.0, consistency_weight: float = 1.0, use_cos_loss: bool = False,
                        eps: Optional[torch.Tensor] = None):
        """fused forward + ELBO + cross entropy of a LatentClassifier on mu and / or of an EmbeddingClassifier on x_hat
        (lightning.py:67-131; heads the fused steps cover): returns
        (total, recon, kl, ce, hits, cons, cons_hits, x_hat, mu, logvar)."""
        hidden, clf_params, cons_params = 0, [], []
        if classifier is not None:
            fp = classifier.fused_params()
            if fp is None:
                raise _ffi.PsvaeError("elbo_heads_step needs a head LatentClassifier.fused_params() accepts")
            hidden, clf_params = fp
        if consistency is not None:
            if not consistency.fused_ok():
                raise _ffi.PsvaeError("elbo_heads_step needs a consistency head EmbeddingClassifier.fused_ok() accepts")
            cons_params = consistency.fused_params()
        seed, off = self._next_rng(x.shape[0] * self.latent_dim)
        flags = self.flags | (_ffi.COS_LOSS if use_cos_loss else 0)
        meta = (seed, off, flags, self._scratch_cache, float(beta), float(classifier_weight), hidden, len(clf_params),
                float(consistency_weight), len(cons_params))
        return ELBOClassifierStepFunction.apply(x, eps, y, meta, *self.ordered_params(), *clf_params, *cons_params)

    def elbo_classifier_step(self, x: torch.Tensor, y: torch.Tensor, classifier, beta: float = 1.0,
                             classifier_weight: float = 1.0, use_cos_loss: bool = False,
                             eps: Optional[torch.Tensor] = None):
        """fused forward + ELBO + classifier cross entropy on mu (lightning.py:67-131 with a single-label head the
        fused classifier step covers): returns (total, recon, kl, ce, hits, x_hat, mu, logvar)."""
        total, recon, kl, ce, hits, _, _, xh, mu, lv = self.elbo_heads_step(
            x, y, classifier, None, beta, classifier_weight, 1.0, use_cos_loss, eps)
        return total, recon, kl, ce, hits, xh, mu, lv

    def elbo_terms(self, x_hat: torch.Tensor, mu: torch.Tensor, log_sigma: torch.Tensor, x: torch.Tensor,
                   use_cos_loss: bool = False):
        """(recon, kl) of lightning.py:110-117 for given forward outputs, computed and differentiated by the HIP kernels"""
        return ELBOTermsFunction.apply(x_hat, mu, log_sigma, x, bool(use_cos_loss), self._scratch_cache)

    def elbo_step(self, x: torch.Tensor, beta: float = 1.0, use_cos_loss: bool = False,
                  eps: Optional[torch.Tensor] = None, outputs: bool = True):
        """fused forward + ELBO (lightning.py:110-124): returns (total, recon, kl, x_hat, mu, logvar).  A training loop
        that keeps calling this with one batch shape is served, from its third call on, by a captured HIP graph of the
        whole fused step (api_graph.py: same numbers, same epsilon stream, ~1/2 of the host time); `outputs=False`
        (what training_step passes: it only needs the losses) lets that path skip the copies of x_hat / mu / logvar
        and returns None for them."""
        seed, off = self._next_rng(x.shape[0] * self.latent_dim)
        flags = self.flags | (_ffi.COS_LOSS if use_cos_loss else 0)
        meta = (seed, off, flags, self._scratch_cache, float(beta), None, self._rank_world()[1], bool(outputs))
        return ELBOStepFunction.apply(x, eps, meta, *self.ordered_params())


PsVAE = VAEModel  # the name BASELINE.json uses
