"""torch.optim.Adam as the reference builds it (ps_vae/lightning.py:204-205: `Adam(self.parameters(), **hparams["optimizer"])`)
with step() as ONE HIP launch over all parameter tensors (psvae_adam_multi_f32) instead of torch's foreach sequence -- about
190 us of host time per step for the model's 18 tensors, more than the whole fused forward + backward.

Drop-in by construction: FusedAdam IS a torch.optim.Adam -- same constructor, same param_groups, same per-parameter state
({'step', 'exp_avg', 'exp_avg_sq'}), so schedulers (lightning.py:206: CosineAnnealingLR), Lightning's checkpointing of
optimizer state and state_dict() / load_state_dict() exchange with a plain torch Adam work unchanged.  Whatever the launch does
not cover (CPU or non-fp32 parameters, sparse gradients, amsgrad, maximize, capturable / differentiable, a tensor lr)
falls through to torch's own step().

Host cost is the point, so the common case -- every parameter of a group has a gradient, all have taken the same number of
steps -- runs from a per-group cache (pointer tables of the parameters and moments, one Python step counter); the
per-parameter 'step' tensors of the state are brought up to date whenever the state is looked at (state_dict)."""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from . import _ffi

MAX_TENSORS = 48  # PSVAE_ADAM_MAX_TENSORS (include/psvae.h)


class _GroupCache:
    __slots__ = ("params", "chunks", "step", "ptr_first", "ptr_last", "device_index")


class FusedAdam(torch.optim.Adam):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._fcache = {}

    # ---- what the launch covers ------------------------------------------------------------------------------------
    @staticmethod
    def _group_ok(group) -> bool:
        return not (group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable")
                    or isinstance(group["lr"], torch.Tensor))

    @staticmethod
    def _tensor_ok(p) -> bool:
        g = p.grad
        return (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_cuda and g.dtype == torch.float32
                and not g.is_sparse and g.is_contiguous() and g.device == p.device)

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:  # torch.optim.Adam's lazy state, same keys and kinds
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        m, v = st["exp_avg"], st["exp_avg_sq"]
        if not (m.is_contiguous() and v.is_contiguous() and m.device == p.device and m.dtype == torch.float32
                and v.device == p.device and v.dtype == torch.float32):
            raise _ffi.PsvaeError("FusedAdam: optimizer state tensors must be contiguous fp32 on the parameter's device")
        return st

    # ---- the cache ---------------------------------------------------------------------------------------------------
    def _sync_steps(self):
        """write the cached step counts into the per-parameter 'step' entries (torch's format) and drop the caches"""
        for c in self._fcache.values():
            for p in c.params:
                s = self.state[p]["step"]
                if isinstance(s, torch.Tensor):
                    s.fill_(float(c.step))
                else:
                    self.state[p]["step"] = float(c.step)
        self._fcache = {}

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self._fcache = {}
        return super().load_state_dict(state_dict)

    def add_param_group(self, param_group):
        if getattr(self, "_fcache", None):
            self._sync_steps()
        return super().add_param_group(param_group)

    def zero_grad(self, set_to_none: bool = True):
        """torch's zero_grad(set_to_none=True) -- every parameter's .grad becomes None -- without its profiler scope and per-call
        bookkeeping (21-28 us of host time per step for 18 tensors against ~3); set_to_none=False is torch's own"""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    def _build_cache(self, group):
        """every parameter of the group has a usable gradient and the same step count: pointer tables for the launches"""
        params = group["params"]
        steps = set()
        for p in params:
            if p.grad is None or not self._tensor_ok(p):
                return None
            steps.add(int(float(self._init_state(p)["step"])))
        if len(steps) != 1 or len({p.device for p in params}) != 1:
            return None
        c = _GroupCache()
        c.params = list(params)
        c.step = steps.pop()
        c.chunks = []
        for o in range(0, len(params), MAX_TENSORS):
            pl = params[o:o + MAX_TENSORS]
            arr = C.c_void_p * len(pl)
            c.chunks.append((pl, arr, arr(*[p.data_ptr() for p in pl]), arr(*[self.state[p]["exp_avg"].data_ptr() for p in pl]),
                             arr(*[self.state[p]["exp_avg_sq"].data_ptr() for p in pl]),
                             (C.c_int64 * len(pl))(*[p.numel() for p in pl])))
        c.ptr_first, c.ptr_last = params[0].data_ptr(), params[-1].data_ptr()
        c.device_index = params[0].device.index
        return c

    # ---- step --------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        if not all(self._group_ok(g) for g in self.param_groups):
            self._sync_steps()
            return super().step(closure)
        if any(id(g) not in self._fcache for g in self.param_groups):
            # (not every group runs from its cache yet: look at every tensor once -- anything the launch does not cover and
            #  the whole optimizer takes torch's own step, as it would have without this class)
            for g in self.param_groups:
                for p in g["params"]:
                    if p.grad is not None and not self._tensor_ok(p):
                        self._sync_steps()
                        return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _ffi.lib()
        for group in self.param_groups:
            params = group["params"]
            if not params:
                continue
            c = self._fcache.get(id(group))
            if c is not None and (len(c.params) != len(params) or c.ptr_first != params[0].data_ptr()
                                  or c.ptr_last != params[-1].data_ptr()):
                self._sync_steps()
                c = None
            if c is None:
                c = self._build_cache(group)
                if c is not None:
                    self._fcache[id(group)] = c
            fast = c is not None
            if fast:
                try:
                    gptrs = []
                    for p in c.params:
                        g = p.grad
                        if g.dtype is not torch.float32 or not g.is_cuda or not g.is_contiguous():
                            fast = False
                            break
                        gptrs.append(g.data_ptr())
                except AttributeError:  # a parameter without a gradient this step
                    fast = False
            if not fast:
                if c is not None:
                    self._sync_steps()
                self._general_step(L, group)
                continue
            c.step += 1
            beta1, beta2 = group["betas"]
            lr, eps, wd = float(group["lr"]), float(group["eps"]), float(group["weight_decay"])
            if torch.cuda.current_device() != c.device_index:  # (a model on another device than the current one: rare)
                with torch.cuda.device(c.device_index):
                    self._launch_chunks(L, c, gptrs, lr, beta1, beta2, eps, wd)
            else:
                self._launch_chunks(L, c, gptrs, lr, beta1, beta2, eps, wd)
        return loss

    @staticmethod
    def _launch_chunks(L, c, gptrs, lr, beta1, beta2, eps, wd):
        stream, o = _ffi.stream_ptr(), 0
        for pl, arr, P, M, V, N in c.chunks:
            G = arr(*gptrs[o:o + len(pl)])
            o += len(pl)
            _ffi.check(L.psvae_adam_multi_f32(len(pl), P, G, M, V, N, c.step, lr, float(beta1), float(beta2), eps, wd, 1.0, stream),
                       "adam_multi")

    def _general_step(self, L, group):
        """any subset of the group's parameters with gradients, any mix of step counts: one launch per distinct step count;
        tensors the launch does not cover make the whole optimizer take torch's own step"""
        by_step = {}
        for p in group["params"]:
            if p.grad is None:
                continue
            if not self._tensor_ok(p):
                raise _ffi.PsvaeError("FusedAdam: a parameter or gradient the fused launch does not cover (CPU, non-fp32, sparse "
                                      "or non-contiguous) -- construct torch.optim.Adam for this model, or set PSVAE_FUSED_ADAM=0")
            st = self._init_state(p)
            by_step.setdefault(int(float(st["step"])) + 1, []).append(p)
        beta1, beta2 = group["betas"]
        for step, plist in by_step.items():
            for o in range(0, len(plist), MAX_TENSORS):
                self._launch(L, plist[o:o + MAX_TENSORS], step, group, beta1, beta2)
            for p in plist:
                s = self.state[p]["step"]
                if isinstance(s, torch.Tensor):
                    s += 1
                else:
                    self.state[p]["step"] = s + 1

    def _launch(self, L, plist: List[torch.Tensor], step: int, group, beta1: float, beta2: float) -> None:
        n = len(plist)
        arr = C.c_void_p * n
        P = arr(*[p.data_ptr() for p in plist])
        G = arr(*[p.grad.data_ptr() for p in plist])
        M = arr(*[self.state[p]["exp_avg"].data_ptr() for p in plist])
        V = arr(*[self.state[p]["exp_avg_sq"].data_ptr() for p in plist])
        N = (C.c_int64 * n)(*[p.numel() for p in plist])
        with torch.cuda.device(plist[0].device):
            _ffi.check(L.psvae_adam_multi_f32(n, P, G, M, V, N, step, float(group["lr"]), float(beta1), float(beta2),
                                              float(group["eps"]), float(group["weight_decay"]), 1.0, _ffi.stream_ptr()),
                       "adam_multi")
