"""Functional layer over the C-ABI: torch tensors in / out, all compute in libpsvae_hip.so.

PyTorch is used for device memory and streams only.  Everything here requires CUDA(HIP) tensors;
there is deliberately no CPU path (see _ffi.ptr).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _ffi
from ._ffi import HIDDEN, NPARAM

GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_MASK, EPI_ACCUM = 0, 1, 2, 3, 4

ACT_SHAPES = {
    "h1m": lambda B, D, Z: (B, HIDDEN), "h2m": lambda B, D, Z: (B, HIDDEN),
    "h1s": lambda B, D, Z: (B, HIDDEN), "h2s": lambda B, D, Z: (B, HIDDEN),
    "eps": lambda B, D, Z: (B, Z), "z": lambda B, D, Z: (B, Z),
    "d1": lambda B, D, Z: (B, HIDDEN), "d2": lambda B, D, Z: (B, HIDDEN),
    "ynorm": lambda B, D, Z: (B,),
}


def _f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise _ffi.PsvaeError(f"{what} must be float32, got {t.dtype}")
    return t.contiguous()


def philox_normal(n: int, seed: int, offset: int, device, fast: bool = False) -> torch.Tensor:
    """n standard normals of the Philox stream (seed, offset); fast: Box-Muller on the hardware transcendentals
    (psvae_philox_normal_fast_f32) -- the form the Langevin chains and the fused decode kernels draw inside their loops"""
    out = torch.empty(n, dtype=torch.float32, device=device)
    fn = _ffi.lib().psvae_philox_normal_fast_f32 if fast else _ffi.lib().psvae_philox_normal_f32
    _ffi.check(fn(_ffi.ptr(out), n, seed, offset, _ffi.stream_ptr()), "philox_normal")
    return out


def gemm(kind: int, A: torch.Tensor, B: torch.Tensor, epi: int = EPI_NONE, aux: Optional[torch.Tensor] = None,
         want_colsum: bool = False, tile: int = 0, out: Optional[torch.Tensor] = None):
    """Test/bench hook: one grouped-kernel GEMM.  NT: A[M,K] B[N,K]; NN: A[M,K] B[K,N]; TN: A[K,M] B[K,N]."""
    # row-pitched views (stride(1) == 1) go through as they are: the C-ABI takes lda / ldb / ldc
    A, B = (t if (t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1) else _f32(t, n)
            for t, n in ((A, "A"), (B, "B")))
    if kind == GEMM_NT:
        (M, K), (N, K2) = A.shape, B.shape
    elif kind == GEMM_NN:
        (M, K), (K2, N) = A.shape, B.shape
    else:
        (K, M), (K2, N) = A.shape, B.shape
    assert K == K2, "contraction mismatch"
    Cm = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=A.device)
    cs = torch.empty(M, dtype=torch.float32, device=A.device) if want_colsum else None
    _ffi.check(_ffi.lib().psvae_gemm_f32(kind, M, N, K, _ffi.ptr(A, True), A.stride(0), _ffi.ptr(B, True), B.stride(0),
                                         _ffi.ptr(Cm, True), Cm.stride(0), _ffi.ptr(aux), _ffi.ptr(cs), epi, tile,
                                         _ffi.stream_ptr()), "gemm")
    return (Cm, cs) if want_colsum else Cm


def mlp_forward(x: torch.Tensor, layers) -> torch.Tensor:
    """y = Linear_n(... relu(Linear_1(x))) on the MFMA GEMM with fused bias / bias + ReLU epilogues, one launch per layer:
    `layers` = [(weight [out, in], bias [out], relu: bool), ...].  Inference only (no autograd): what the classifier heads'
    forward() takes when gradients are off (latent_classifier.py:55-70, embedding_classifier.py:49-62)."""
    h = _f32(x, "x")
    for W, b, relu in layers:
        h = gemm(GEMM_NT, h, W.detach(), EPI_BIAS_RELU if relu else EPI_BIAS, aux=_f32(b.detach(), "bias"))
    return h


def mlp_forward_ok(x: torch.Tensor, linears) -> bool:
    """whether mlp_forward applies: a 2-D fp32 CUDA batch, fp32 CUDA Linear layers, gradients not being recorded"""
    if torch.is_grad_enabled() and (x.requires_grad or any(l.weight.requires_grad for l in linears)):
        return False
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] > 0
            and all(l.weight.is_cuda and l.weight.dtype == torch.float32 and l.bias is not None for l in linears))


class Workspace:
    """Caller-owned buffers of one batch shape: saved activations, outputs, scratch, gradients."""

    def __init__(self, B: int, D: int, Z: int, device, normalize: bool = False):
        self.B, self.D, self.Z = B, D, Z
        self.device = torch.device(device)
        self.dims = _ffi.Dims(B, D, Z)
        self.acts: Dict[str, torch.Tensor] = {
            k: torch.empty(f(B, D, Z), dtype=torch.float32, device=device)
            for k, f in ACT_SHAPES.items() if (k != "ynorm" or normalize)}
        self.acts_c = _ffi.make_acts(self.acts)
        nbytes = _ffi.scratch_bytes(B, D, Z)
        self.scratch = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.loss3 = torch.zeros(3, dtype=torch.float32, device=device)

    def new_outputs(self):
        B, D, Z = self.B, self.D, self.Z
        return (torch.empty(B, D, dtype=torch.float32, device=self.device),
                torch.empty(B, Z, dtype=torch.float32, device=self.device),
                torch.empty(B, Z, dtype=torch.float32, device=self.device))


def vae_forward(ws: Workspace, x: torch.Tensor, params: Sequence[torch.Tensor], eps: Optional[torch.Tensor] = None,
                seed: int = 0, offset: int = 0, flags: int = 0, cond=None,
                outs: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None,
                rng_state: Optional[torch.Tensor] = None):
    """VAEModel.forward on the GPU: returns (x_hat, mu, logvar); activations land in ws.acts."""
    x = _f32(x, "x")
    assert x.shape == (ws.B, ws.D), (x.shape, ws.B, ws.D)
    xh, mu, lv = outs if outs is not None else ws.new_outputs()
    P = _ffi.make_params(params)
    _ffi.check(_ffi.lib().psvae_forward_f32(
        C.byref(ws.dims), _ffi.ptr(x), C.byref(P), C.byref(cond) if cond is not None else None,
        _ffi.ptr(_f32(eps, "eps")) if eps is not None else None, seed, offset, _ffi.ptr(rng_state),
        C.byref(ws.acts_c),
        _ffi.ptr(xh), _ffi.ptr(mu), _ffi.ptr(lv), _ffi.ptr(ws.scratch), flags, _ffi.stream_ptr()), "forward")
    return xh, mu, lv


def elbo(ws: Workspace, x, xh, mu, lv, beta: float, flags: int = 0, loss3: Optional[torch.Tensor] = None):
    loss3 = ws.loss3 if loss3 is None else loss3
    _ffi.check(_ffi.lib().psvae_elbo_f32(C.byref(ws.dims), _ffi.ptr(x), _ffi.ptr(xh), _ffi.ptr(mu), _ffi.ptr(lv),
                                         beta, _ffi.ptr(loss3), _ffi.ptr(ws.scratch), flags, _ffi.stream_ptr()), "elbo")
    return loss3


def vae_backward(ws: Workspace, x, params: Sequence[torch.Tensor], xh, mu, lv, grads: Sequence[torch.Tensor],
                 g_xhat=None, g_mu=None, g_logvar=None, elbo_grad: bool = False, beta: float = 1.0,
                 grad_loss: Optional[torch.Tensor] = None, g_x: Optional[torch.Tensor] = None, flags: int = 0,
                 phases: int = _ffi.PHASE_ALL, cond=None):
    P = _ffi.make_params(params)
    G = _ffi.make_grads(grads)
    _ffi.check(_ffi.lib().psvae_backward_f32(
        C.byref(ws.dims), _ffi.ptr(x), C.byref(P), C.byref(cond) if cond is not None else None,
        C.byref(ws.acts_c), _ffi.ptr(xh), _ffi.ptr(mu), _ffi.ptr(lv),
        _ffi.ptr(g_xhat), _ffi.ptr(g_mu), _ffi.ptr(g_logvar), int(elbo_grad), beta, _ffi.ptr(grad_loss),
        C.byref(G), _ffi.ptr(g_x), _ffi.ptr(ws.scratch), flags, phases, _ffi.stream_ptr()), "backward")
    return grads


def train_step(ws: Workspace, x, params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor], outs, loss3,
               beta: float = 1.0, eps: Optional[torch.Tensor] = None, seed: int = 0, offset: int = 0,
               rng_state: Optional[torch.Tensor] = None, grad_loss: Optional[torch.Tensor] = None,
               phases: int = _ffi.PHASE_ALL, params_kmajor: Optional[Sequence[torch.Tensor]] = None, cond=None,
               flags: int = 0):
    """fused forward + ELBO + backward: psvae_train_step_f32.  params_kmajor: the k-major ([in, out]) copies of the
    18-slot parameter list (weights transposed, bias slots ignored) when the caller maintains them (TrainEngine does);
    cond: a _ffi.Cond (one-hot tables); flags: NORMALIZE_DECODER | COS_LOSS."""
    P = _ffi.make_params(params)
    G = _ffi.make_grads(grads)
    PK = _ffi.make_params(params_kmajor) if params_kmajor is not None else None
    opts = _ffi.make_step_opts(PK, cond, flags) if (PK is not None or cond is not None or flags) else None
    xh, mu, lv = outs
    _ffi.check(_ffi.lib().psvae_train_step_f32(
        C.byref(ws.dims), _ffi.ptr(x), C.byref(P), _ffi.ptr(eps), seed, offset, _ffi.ptr(rng_state),
        C.byref(ws.acts_c), _ffi.ptr(xh), _ffi.ptr(mu), _ffi.ptr(lv), beta, _ffi.ptr(grad_loss), C.byref(G),
        _ffi.ptr(loss3), _ffi.ptr(ws.scratch), phases, C.byref(opts) if opts is not None else None,
        _ffi.stream_ptr()), "train_step")


def last_step_launches() -> int:
    """kernel launches issued by this thread's most recent train_step call"""
    return int(_ffi.lib().psvae_last_step_launches())


def transpose_params(ws: Workspace, params: Sequence[torch.Tensor], params_kmajor: Sequence[torch.Tensor]) -> None:
    """(re)make the k-major weight copies: params_kmajor[2k] <- params[2k]^T (one launch)"""
    P, PK = _ffi.make_params(params), _ffi.make_params(params_kmajor)
    _ffi.check(_ffi.lib().psvae_transpose_params_f32(C.byref(ws.dims), C.byref(P), C.byref(PK), _ffi.stream_ptr()),
               "transpose_params")


def tune_train_step(ws: Workspace, x, params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor], outs, loss3,
                    iters: int = 10, params_kmajor: Optional[Sequence[torch.Tensor]] = None) -> str:
    """Autotune the fused step's launch plan for this (B, D, Z) (psvae_tune_train_step_f32): times the whole step
    under each candidate plan on the current stream -- a few seconds -- and keeps the fastest for every later
    train_step call of this process, but only if it beats the heuristic plan by a margin in the same session.
    Overwrites ws.acts / outs / grads / loss3.  Returns the plan's text form."""
    P = _ffi.make_params(params)
    G = _ffi.make_grads(grads)
    PK = _ffi.make_params(params_kmajor) if params_kmajor is not None else None
    opts = _ffi.make_step_opts(PK) if PK is not None else None
    xh, mu, lv = outs
    _ffi.check(_ffi.lib().psvae_tune_train_step_f32(
        C.byref(ws.dims), _ffi.ptr(x), C.byref(P), C.byref(ws.acts_c), _ffi.ptr(xh), _ffi.ptr(mu), _ffi.ptr(lv),
        C.byref(G), _ffi.ptr(loss3), _ffi.ptr(ws.scratch), iters, C.byref(opts) if opts is not None else None,
        _ffi.stream_ptr()), "tune_train_step")
    return describe_plan(ws)


PLAN_FILE = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)),
                                       "launch_plans.json")


def plan_key(ws: Workspace) -> str:
    """plans are keyed by GPU architecture + CU count and the batch shape (the marketing name is not stable)"""
    pr = torch.cuda.get_device_properties(ws.device)
    arch = getattr(pr, "gcnArchName", "gpu").split(":")[0]
    return f"{arch}/{pr.multi_processor_count}cu|{ws.B}x{ws.D}x{ws.Z}"


def _read_plans(path):
    import json, os
    if path and os.path.exists(path):
        try:
            return json.load(open(path))
        except (OSError, ValueError):
            pass
    return {}


def load_committed_plan(ws: Workspace, cache_path: Optional[str] = None) -> Optional[str]:
    """install the plan recorded for this GPU model and (B, D, Z): $PSVAE_PLAN_CACHE / cache_path first, then the
    committed pseudo_speaker_vae_amd/launch_plans.json.  Returns its text form, or None if there is no entry."""
    import os
    key = plan_key(ws)
    for path in (cache_path or os.environ.get("PSVAE_PLAN_CACHE"), PLAN_FILE):
        text = _read_plans(path).get(key)
        if text:
            try:
                set_plan(ws, text)
                return describe_plan(ws)
            except _ffi.PsvaeError:
                continue  # written by another library version
    return None


def tune_or_load_plan(ws: Workspace, x, params, grads, outs, loss3, iters: int = 10, cache_path: Optional[str] = None,
                      force: bool = False, params_kmajor=None, step_kwargs=None) -> str:
    """The recorded plan for this GPU model and shape if there is one (so every run uses the SAME plan); else -- or
    with force=True -- tune_train_step, and remember the result in cache_path / $PSVAE_PLAN_CACHE when one is given."""
    import json, os
    if not force:
        got = load_committed_plan(ws, cache_path)
        if got is not None:
            return got
    plan = tune_train_step(ws, x, params, grads, outs, loss3, iters=iters, params_kmajor=params_kmajor)
    cache_path = cache_path or os.environ.get("PSVAE_PLAN_CACHE")
    if cache_path:
        cache = _read_plans(cache_path)
        cache[plan_key(ws)] = plan
        tmp = f"{cache_path}.tmp{os.getpid()}"
        with open(tmp, "w") as f:
            json.dump(cache, f, indent=1)
        os.replace(tmp, cache_path)
    return plan


def describe_plan(ws: Workspace) -> str:
    buf = C.create_string_buffer(1024)
    _ffi.check(_ffi.lib().psvae_describe_plan(C.byref(ws.dims), buf, 1024), "describe_plan")
    return buf.value.decode()


def set_plan(ws: Workspace, text: str) -> None:
    """install a launch plan from describe_plan's text form ("untuned" removes it)"""
    _ffi.check(_ffi.lib().psvae_set_plan(C.byref(ws.dims), text.encode()), "set_plan")


class Bf16Workspace:
    """buffers of one batch shape for the bf16 training pass (scratch is zero-filled once: see psvae.h)"""

    def __init__(self, B: int, D: int, Z: int, device):
        self.B, self.D, self.Z = B, D, Z
        self.dims = _ffi.Dims(B, D, Z)
        nbytes = int(_ffi.lib().psvae_bf16_scratch_bytes(C.byref(self.dims)))
        self.scratch = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
        self.eps, self.z = f(B, Z), f(B, Z)
        self.x_hat, self.mu, self.logvar = f(B, D), f(B, Z), f(B, Z)


def train_step_bf16(ws: Bf16Workspace, x, params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor], loss3,
                    beta: float = 1.0, eps: Optional[torch.Tensor] = None, seed: int = 0, offset: int = 0,
                    rng_state: Optional[torch.Tensor] = None, grad_loss: Optional[torch.Tensor] = None, cond=None,
                    flags: int = 0):
    """one training pass with bf16 MFMA operands (config 3): psvae_train_step_bf16; flags: NORMALIZE_DECODER | COS_LOSS |
    BF16_PACKS_CURRENT (the weight packs in ws.scratch are current: pack_params_bf16)"""
    P = _ffi.make_params(params)
    G = _ffi.make_grads(grads)
    _ffi.check(_ffi.lib().psvae_train_step_bf16(
        C.byref(ws.dims), _ffi.ptr(_f32(x, "x")), C.byref(P), C.byref(cond) if cond is not None else None,
        _ffi.ptr(eps), seed, offset, _ffi.ptr(rng_state), _ffi.ptr(ws.eps), _ffi.ptr(ws.z), _ffi.ptr(ws.x_hat),
        _ffi.ptr(ws.mu), _ffi.ptr(ws.logvar), beta, _ffi.ptr(grad_loss), C.byref(G), _ffi.ptr(loss3),
        _ffi.ptr(ws.scratch), flags, _ffi.stream_ptr()), "train_step_bf16")


def pack_params_bf16(ws: Bf16Workspace, params: Sequence[torch.Tensor]) -> None:
    """refresh the bf16 working copies of the weights inside ws.scratch (psvae_pack_params_bf16): after every parameter
    update when the passes run with flags | BF16_PACKS_CURRENT"""
    P = _ffi.make_params(params)
    _ffi.check(_ffi.lib().psvae_pack_params_bf16(C.byref(ws.dims), C.byref(P), _ffi.ptr(ws.scratch), _ffi.stream_ptr()),
               "pack_params_bf16")


def wgrad_bf16(dY: torch.Tensor, X: torch.Tensor, split: int = 1, bias: bool = True):
    """dW = dY^T X (fp32 [out, in]) and db = column sums of dY from bf16 operands [B, out], [B, in]: psvae_wgrad_bf16, the
    weight-gradient launch of the bf16 pass by itself"""
    if dY.dtype != torch.bfloat16 or X.dtype != torch.bfloat16 or dY.dim() != 2 or X.dim() != 2 or dY.shape[0] != X.shape[0]:
        raise _ffi.PsvaeError("wgrad_bf16: bf16 [B, out] and [B, in]")
    if dY.stride(1) != 1 or X.stride(1) != 1:
        raise _ffi.PsvaeError("wgrad_bf16: rows must be contiguous")
    B, out, inn = dY.shape[0], dY.shape[1], X.shape[1]
    dW = torch.empty(out, inn, dtype=torch.float32, device=dY.device)
    db = torch.empty(out, dtype=torch.float32, device=dY.device) if bias else None
    nbytes = int(_ffi.lib().psvae_wgrad_bf16_scratch_bytes(out, inn, split))
    scratch = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dY.device)
    _ffi.check(_ffi.lib().psvae_wgrad_bf16(dY.data_ptr(), dY.stride(0), X.data_ptr(), X.stride(0), B, out, inn, _ffi.ptr(dW),
                                           _ffi.ptr(db), split, _ffi.ptr(scratch), _ffi.stream_ptr()), "wgrad_bf16")
    return dW, db


def decode(z: torch.Tensor, dec_params: Sequence[torch.Tensor], D: int, flags: int = 0,
           cond_tab=None, gender=None, age=None, out: Optional[torch.Tensor] = None,
           scratch: Optional[torch.Tensor] = None) -> torch.Tensor:
    """VAEModel.decode on the GPU (decoder tensors = params[12:18])."""
    z = _f32(z, "z")
    N, Z = z.shape
    xh = out if out is not None else torch.empty(N, D, dtype=torch.float32, device=z.device)
    nbytes = int(_ffi.lib().psvae_decode_scratch_bytes(N))
    if scratch is None or scratch.numel() < nbytes:
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=z.device)
    arr = (C.c_void_p * 6)(*[_ffi.ptr(_f32(t, "decoder param")) for t in dec_params])
    _ffi.check(_ffi.lib().psvae_decode_f32(N, D, Z, _ffi.ptr(z), arr, _ffi.ptr(cond_tab), _ffi.ptr(gender),
                                           _ffi.ptr(age), _ffi.ptr(xh), _ffi.ptr(scratch), flags,
                                           _ffi.stream_ptr()), "decode")
    return xh


def sample_decode(N: int, Z: int, dec_params: Sequence[torch.Tensor], D: int, seed: int, offset: int, flags: int = 0,
                  bf16: bool = False, out: Optional[torch.Tensor] = None, scratch: Optional[torch.Tensor] = None,
                  z_out: Optional[torch.Tensor] = None, want_z: bool = False):
    """z ~ N(0, I) [N, Z] from the Philox stream (seed, offset) and x_hat = decoder(z) in ONE call
    (psvae_sample_decode_f32 / _bf16; inference.py:22-25).  On the fused decode's shapes z is drawn inside the decode
    kernel and never touches memory unless `want_z` / `z_out` asks for it; returns (x_hat, z or None)."""
    dev = dec_params[0].device
    xh = out if out is not None else torch.empty(N, D, dtype=torch.float32, device=dev)
    L = _ffi.lib()
    nbytes = int(L.psvae_decode_bf16_scratch_bytes(N, D, Z) if bf16 else L.psvae_decode_scratch_bytes(N))
    if scratch is None or scratch.numel() < nbytes:
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    arr = (C.c_void_p * 6)(*[_ffi.ptr(_f32(t, "decoder param")) for t in dec_params])
    fn = L.psvae_sample_decode_bf16 if bf16 else L.psvae_sample_decode_f32
    z = z_out if z_out is not None else (torch.empty(N, Z, dtype=torch.float32, device=dev) if want_z else None)
    rc = fn(N, D, Z, seed, offset, arr, _ffi.ptr(z), _ffi.ptr(xh), _ffi.ptr(scratch), flags, _ffi.stream_ptr())
    if rc == -1 and z is None:  # a shape the fused kernels do not cover: the draw needs a buffer
        z = torch.empty(N, Z, dtype=torch.float32, device=dev)
        rc = fn(N, D, Z, seed, offset, arr, _ffi.ptr(z), _ffi.ptr(xh), _ffi.ptr(scratch), flags, _ffi.stream_ptr())
        z = z if want_z else None
    _ffi.check(rc, "sample_decode")
    return xh, z


def last_step_launches_bf16() -> int:
    """launches of this thread's most recent bf16 pass (7 = the row-chain path at its default loss)"""
    return int(_ffi.lib().psvae_last_step_launches_bf16())


def last_decode_launches() -> int:
    """launches of this thread's most recent decode call (4 + normalisation = the fused three-layer kernel ran)"""
    return int(_ffi.lib().psvae_last_decode_launches())


def forward_bf16(ws: Bf16Workspace, x, params: Sequence[torch.Tensor], eps: Optional[torch.Tensor] = None, seed: int = 0,
                 offset: int = 0, rng_state: Optional[torch.Tensor] = None, grad_loss: Optional[torch.Tensor] = None,
                 cond=None, flags: int = 0):
    """forward half of the bf16 pass (psvae_forward_bf16): fills ws.x_hat / mu / logvar / eps / z and the scratch"""
    P = _ffi.make_params(params)
    _ffi.check(_ffi.lib().psvae_forward_bf16(
        C.byref(ws.dims), _ffi.ptr(_f32(x, "x")), C.byref(P), C.byref(cond) if cond is not None else None, _ffi.ptr(eps),
        seed, offset, _ffi.ptr(rng_state), _ffi.ptr(ws.eps), _ffi.ptr(ws.z), _ffi.ptr(ws.x_hat), _ffi.ptr(ws.mu),
        _ffi.ptr(ws.logvar), _ffi.ptr(grad_loss), _ffi.ptr(ws.scratch), flags, _ffi.stream_ptr()), "forward_bf16")


def backward_bf16(ws: Bf16Workspace, x, params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor], loss3,
                  beta: float = 1.0, rng_state: Optional[torch.Tensor] = None, grad_loss: Optional[torch.Tensor] = None,
                  cond=None, flags: int = 0):
    """backward half (psvae_backward_bf16) on the workspace a forward_bf16 call just filled (same flags)"""
    P = _ffi.make_params(params)
    G = _ffi.make_grads(grads)
    _ffi.check(_ffi.lib().psvae_backward_bf16(
        C.byref(ws.dims), _ffi.ptr(_f32(x, "x")), C.byref(P), C.byref(cond) if cond is not None else None,
        _ffi.ptr(rng_state), _ffi.ptr(ws.eps), _ffi.ptr(ws.z), _ffi.ptr(ws.x_hat), _ffi.ptr(ws.mu), _ffi.ptr(ws.logvar),
        beta, _ffi.ptr(grad_loss), C.byref(G), _ffi.ptr(loss3), _ffi.ptr(ws.scratch), flags, _ffi.stream_ptr()),
        "backward_bf16")


def decode_bf16(z: torch.Tensor, dec_params: Sequence[torch.Tensor], D: int, flags: int = 0,
                out: Optional[torch.Tensor] = None, scratch: Optional[torch.Tensor] = None) -> torch.Tensor:
    """VAEModel.decode with bf16 operands / fp32 accumulation (psvae_decode_bf16): the fast sampling mode"""
    z = _f32(z, "z")
    N, Z = z.shape
    xh = out if out is not None else torch.empty(N, D, dtype=torch.float32, device=z.device)
    nbytes = int(_ffi.lib().psvae_decode_bf16_scratch_bytes(N, D, Z))
    if scratch is None or scratch.numel() < nbytes:
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=z.device)
    arr = (C.c_void_p * 6)(*[_ffi.ptr(_f32(t, "decoder param")) for t in dec_params])
    _ffi.check(_ffi.lib().psvae_decode_bf16(N, D, Z, _ffi.ptr(z), arr, _ffi.ptr(xh), _ffi.ptr(scratch), flags,
                                            _ffi.stream_ptr()), "decode_bf16")
    return xh


def adam_step(param, grad, exp_avg, exp_avg_sq, step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
              weight_decay: float = 0.0, grad_scale: float = 1.0):
    n = param.numel()
    _ffi.check(_ffi.lib().psvae_adam_step_f32(_ffi.ptr(param), _ffi.ptr(grad), _ffi.ptr(exp_avg), _ffi.ptr(exp_avg_sq),
                                              n, step, lr, betas[0], betas[1], eps, weight_decay, grad_scale,
                                              _ffi.stream_ptr()), "adam_step")


def adam_step_dev(layout: "_ffi.AdamDev", state: torch.Tensor, betas=(0.9, 0.999), eps: float = 1e-8,
                  weight_decay: float = 0.0, grad_scale: float = 1.0):
    """psvae_adam_step_dev_f32: Adam with {lr, step} on the device (state: 24-byte uint8 tensor holding a
    psvae_adam_state) that also maintains the k-major weight copies; graph-capturable."""
    _ffi.check(_ffi.lib().psvae_adam_step_dev_f32(C.byref(layout), _ffi.ptr(state), betas[0], betas[1], eps,
                                                  weight_decay, grad_scale, _ffi.stream_ptr()), "adam_step_dev")
