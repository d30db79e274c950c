"""ctypes binding of libpsvae_hip.so (include/psvae.h).  Thin: pointers in, error codes out.

The library is REQUIRED: there is no CPU or eager-PyTorch fallback anywhere in the product path.
A missing / unloadable library raises at first use.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

PKG = os.path.dirname(os.path.abspath(__file__))
# PSVAE_LIB: tuning only -- load another build of the same ABI (tools/build_variant.sh)
LIB_PATH = os.environ.get("PSVAE_LIB") or os.path.join(PKG, "libpsvae_hip.so")

NPARAM = 18
HIDDEN = 512
COND_DIM = 6
NORMALIZE_DECODER = 1
COS_LOSS = 2
BF16_PACKS_CURRENT = 0x100  # psvae_*_bf16: the weight packs in scratch are current (psvae_pack_params_bf16)
PHASE_DECODER, PHASE_ENCODER, PHASE_ALL = 1, 2, 3
ABI_VERSION = 4

# every symbol include/psvae.h declares (tests check the .so exports each of them)
SYMBOLS = (
    "psvae_version", "psvae_debug_force_tile", "psvae_debug_force_wgrad_split", "psvae_strerror", "psvae_scratch_bytes", "psvae_philox_normal_f32",
    "psvae_forward_f32", "psvae_elbo_f32", "psvae_backward_f32", "psvae_train_step_f32",
    "psvae_tune_train_step_f32", "psvae_describe_plan", "psvae_set_plan",
    "psvae_classifier_scratch_bytes", "psvae_classifier_ce_f32", "psvae_consistency_scratch_bytes", "psvae_consistency_ce_f32", "psvae_decode_bf16_scratch_bytes", "psvae_decode_bf16", "psvae_forward_bf16", "psvae_backward_bf16",
    "psvae_bf16_scratch_bytes", "psvae_train_step_bf16", "psvae_pack_params_bf16", "psvae_philox_normal_fast_f32", "psvae_adam_multi_f32", "psvae_wgrad_bf16_scratch_bytes", "psvae_wgrad_bf16", "psvae_decode_scratch_bytes",
    "psvae_decode_f32", "psvae_gemm_f32", "psvae_langevin_f32", "psvae_gather_rows_f32", "psvae_adam_step_f32",
    "psvae_last_step_launches", "psvae_transpose_params_f32", "psvae_adam_step_dev_f32", "psvae_elbo_grad_f32",
    "psvae_last_decode_launches", "psvae_last_step_launches_bf16", "psvae_sample_decode_f32", "psvae_sample_decode_bf16",
    "psvae_ipc_blob_bytes", "psvae_ipc_comm_create", "psvae_ipc_comm_connect", "psvae_ipc_allreduce_avg_f32",
    "psvae_ipc_comm_destroy", "psvae_langevin_heads_f32", "psvae_classifier_heads_ce_f32",
)


class Dims(C.Structure):
    _fields_ = [("B", C.c_int64), ("D", C.c_int64), ("Z", C.c_int64)]


class Params(C.Structure):
    _fields_ = [("p", C.c_void_p * NPARAM)]


class Grads(C.Structure):
    _fields_ = [("g", C.c_void_p * NPARAM)]


class Cond(C.Structure):
    _fields_ = [("tab", C.c_void_p * 3), ("gtab", C.c_void_p * 3), ("gender", C.c_void_p), ("age", C.c_void_p)]


class StepOpts(C.Structure):
    _fields_ = [("params_kmajor", C.POINTER(Params)), ("cond", C.POINTER(Cond)), ("flags", C.c_uint32)]


class AdamState(C.Structure):  # device-resident: lr, completed steps
    _fields_ = [("lr", C.c_double), ("step", C.c_int64), ("reserved", C.c_uint32 * 2)]


class AdamDev(C.Structure):
    _fields_ = [("param", C.c_void_p), ("param_kmajor", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p),
                ("exp_avg_sq", C.c_void_p), ("mat_off", C.c_int64 * 9), ("mat_rows", C.c_int64 * 9),
                ("mat_cols", C.c_int64 * 9), ("vec_off", C.c_int64 * 16), ("vec_len", C.c_int64 * 16),
                ("n_vec", C.c_int32)]


class Heads(C.Structure):
    """psvae_heads: the labels of a (multi-label) LatentClassifier head, output Linears stacked"""
    _fields_ = [("n_heads", C.c_int32), ("size", C.c_int32 * 8), ("target", C.c_int32 * 8), ("activation", C.c_int32),
                ("prior_scale", C.c_float)]


ACTIVATION_IDS = {"relu": 0, "leaky_relu": 1, "tanh": 2, "sigmoid": 3}


def make_heads(sizes, targets=None, activation: str = "relu", prior_scale: float = 1.0) -> Heads:
    h = Heads()
    h.prior_scale = float(prior_scale)
    h.n_heads = len(sizes)
    for i, n in enumerate(sizes):
        h.size[i] = int(n)
        h.target[i] = int(targets[i]) if targets is not None else 0
    h.activation = ACTIVATION_IDS[activation]
    return h


class Mlp3Grads(C.Structure):
    _fields_ = [("g", C.c_void_p * 6)]


ACT_FIELDS = ("h1m", "h2m", "h1s", "h2s", "eps", "z", "d1", "d2", "ynorm")


class Acts(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ACT_FIELDS]


class PsvaeError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PsvaeError(
            f"{LIB_PATH} is missing: build it with `python -m pseudo_speaker_vae_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    L = C.CDLL(LIB_PATH)
    L.psvae_version.restype = C.c_int
    L.psvae_strerror.restype = C.c_char_p
    L.psvae_strerror.argtypes = [C.c_int]
    L.psvae_scratch_bytes.restype = C.c_size_t
    L.psvae_scratch_bytes.argtypes = [C.POINTER(Dims)]
    L.psvae_decode_scratch_bytes.restype = C.c_size_t
    L.psvae_decode_scratch_bytes.argtypes = [C.c_int64]
    vp, u64, i64, u32, f32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_uint32, C.c_float
    L.psvae_philox_normal_f32.argtypes = [vp, i64, u64, u64, vp]
    L.psvae_philox_normal_fast_f32.argtypes = [vp, i64, u64, u64, vp]
    L.psvae_philox_normal_fast_f32.restype = C.c_int
    L.psvae_forward_f32.argtypes = [C.POINTER(Dims), vp, C.POINTER(Params), C.POINTER(Cond), vp, u64, u64, vp,
                                    C.POINTER(Acts), vp, vp, vp, vp, u32, vp]
    L.psvae_elbo_f32.argtypes = [C.POINTER(Dims), vp, vp, vp, vp, f32, vp, vp, u32, vp]
    L.psvae_backward_f32.argtypes = [C.POINTER(Dims), vp, C.POINTER(Params), C.POINTER(Cond), C.POINTER(Acts),
                                     vp, vp, vp, vp, vp, vp, C.c_int, f32, vp, C.POINTER(Grads), vp, vp, u32, u32, vp]
    L.psvae_tune_train_step_f32.argtypes = [C.POINTER(Dims), vp, C.POINTER(Params), C.POINTER(Acts), vp, vp, vp,
                                            C.POINTER(Grads), vp, vp, C.c_int, C.POINTER(StepOpts), vp]
    L.psvae_tune_train_step_f32.restype = C.c_int
    L.psvae_describe_plan.argtypes = [C.POINTER(Dims), C.c_char_p, C.c_size_t]
    L.psvae_describe_plan.restype = C.c_int
    L.psvae_set_plan.argtypes = [C.POINTER(Dims), C.c_char_p]
    L.psvae_set_plan.restype = C.c_int
    L.psvae_train_step_f32.argtypes = [C.POINTER(Dims), vp, C.POINTER(Params), vp, u64, u64, vp, C.POINTER(Acts),
                                       vp, vp, vp, f32, vp, C.POINTER(Grads), vp, vp, u32, C.POINTER(StepOpts), vp]
    L.psvae_elbo_grad_f32.argtypes = [C.POINTER(Dims), vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, vp]
    L.psvae_elbo_grad_f32.restype = C.c_int
    L.psvae_last_step_launches.restype = C.c_int
    L.psvae_last_step_launches.argtypes = []
    L.psvae_last_decode_launches.restype = C.c_int
    L.psvae_last_decode_launches.argtypes = []
    L.psvae_last_step_launches_bf16.restype = C.c_int
    L.psvae_last_step_launches_bf16.argtypes = []
    L.psvae_transpose_params_f32.argtypes = [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Params), vp]
    L.psvae_transpose_params_f32.restype = C.c_int
    L.psvae_adam_step_dev_f32.argtypes = [C.POINTER(AdamDev), vp, C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.c_double, vp]
    L.psvae_adam_step_dev_f32.restype = C.c_int
    L.psvae_bf16_scratch_bytes.restype = C.c_size_t
    L.psvae_bf16_scratch_bytes.argtypes = [C.POINTER(Dims)]
    L.psvae_train_step_bf16.argtypes = [C.POINTER(Dims), vp, C.POINTER(Params), C.POINTER(Cond), vp, u64, u64, vp, vp,
                                        vp, vp, vp, vp, f32, vp, C.POINTER(Grads), vp, vp, u32, vp]
    L.psvae_train_step_bf16.restype = C.c_int
    L.psvae_pack_params_bf16.argtypes = [C.POINTER(Dims), C.POINTER(Params), vp, vp]
    L.psvae_pack_params_bf16.restype = C.c_int
    L.psvae_adam_multi_f32.argtypes = [C.c_int, vp, vp, vp, vp, vp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                       C.c_double, C.c_double, vp]
    L.psvae_adam_multi_f32.restype = C.c_int
    L.psvae_wgrad_bf16_scratch_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    L.psvae_wgrad_bf16_scratch_bytes.restype = C.c_size_t
    L.psvae_wgrad_bf16.argtypes = [vp, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp]
    L.psvae_wgrad_bf16.restype = C.c_int
    L.psvae_decode_f32.argtypes = [i64, i64, i64, vp, C.POINTER(vp), vp, vp, vp, vp, vp, u32, vp]
    L.psvae_gemm_f32.argtypes = [C.c_int, i64, i64, i64, vp, i64, vp, i64, vp, i64, vp, vp, C.c_int, C.c_int, vp]
    L.psvae_langevin_f32.argtypes = [i64, i64, i64, i64, vp, vp, vp, vp, vp, C.c_int32, f32, f32, i64, vp, u64, u64, vp, vp]
    L.psvae_langevin_f32.restype = C.c_int
    L.psvae_classifier_scratch_bytes.argtypes = [i64, i64, i64]
    L.psvae_classifier_scratch_bytes.restype = C.c_size_t
    L.psvae_classifier_ce_f32.argtypes = [i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.psvae_classifier_ce_f32.restype = C.c_int
    L.psvae_consistency_scratch_bytes.argtypes = [i64, i64, i64]
    L.psvae_consistency_scratch_bytes.restype = C.c_size_t
    L.psvae_consistency_ce_f32.argtypes = [i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp,
                                           C.POINTER(Mlp3Grads), vp, vp]
    L.psvae_consistency_ce_f32.restype = C.c_int
    L.psvae_decode_bf16_scratch_bytes.argtypes = [i64, i64, i64]
    L.psvae_decode_bf16_scratch_bytes.restype = C.c_size_t
    L.psvae_decode_bf16.argtypes = [i64, i64, i64, vp, C.POINTER(vp), vp, vp, u32, vp]
    L.psvae_decode_bf16.restype = C.c_int
    L.psvae_forward_bf16.argtypes = [C.POINTER(Dims), vp, C.POINTER(Params), vp, vp, u64, u64, vp, vp, vp, vp, vp, vp, vp, vp, u32, vp]
    L.psvae_forward_bf16.restype = C.c_int
    L.psvae_backward_bf16.argtypes = [C.POINTER(Dims), vp, C.POINTER(Params), vp, vp, vp, vp, vp, vp, vp, f32, vp,
                                      C.POINTER(Grads), vp, vp, u32, vp]
    L.psvae_backward_bf16.restype = C.c_int
    for fn in (L.psvae_sample_decode_f32, L.psvae_sample_decode_bf16):
        fn.argtypes = [i64, i64, i64, u64, u64, C.POINTER(vp), vp, vp, vp, u32, vp]
        fn.restype = C.c_int
    L.psvae_langevin_heads_f32.argtypes = [i64, i64, i64, C.POINTER(Heads), vp, vp, vp, vp, vp, f32, f32, i64, vp, u64, u64, vp, vp]
    L.psvae_langevin_heads_f32.restype = C.c_int
    L.psvae_classifier_heads_ce_f32.argtypes = [i64, i64, i64, C.POINTER(Heads), vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp,
                                                vp, vp, vp, vp]
    L.psvae_classifier_heads_ce_f32.restype = C.c_int
    L.psvae_ipc_blob_bytes.restype = C.c_size_t
    L.psvae_ipc_blob_bytes.argtypes = []
    L.psvae_ipc_comm_create.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_char_p, C.POINTER(vp), vp]
    L.psvae_ipc_comm_create.restype = C.c_int
    L.psvae_ipc_comm_connect.argtypes = [vp, vp]
    L.psvae_ipc_comm_connect.restype = C.c_int
    L.psvae_ipc_allreduce_avg_f32.argtypes = [vp, vp, C.c_double, vp]
    L.psvae_ipc_allreduce_avg_f32.restype = C.c_int
    L.psvae_ipc_comm_destroy.argtypes = [vp]
    L.psvae_ipc_comm_destroy.restype = C.c_int
    L.psvae_gather_rows_f32.argtypes = [vp, i64, i64, vp, i64, vp, vp]
    L.psvae_gather_rows_f32.restype = C.c_int
    f64 = C.c_double
    L.psvae_adam_step_f32.argtypes = [vp, vp, vp, vp, i64, i64, f64, f64, f64, f64, f64, f64, vp]
    for fn in ("psvae_philox_normal_f32", "psvae_forward_f32", "psvae_elbo_f32", "psvae_backward_f32",
               "psvae_train_step_f32", "psvae_debug_force_tile", "psvae_debug_force_wgrad_split",
               "psvae_decode_f32", "psvae_gemm_f32", "psvae_adam_step_f32"):
        getattr(L, fn).restype = C.c_int
    if L.psvae_version() != ABI_VERSION:
        raise PsvaeError(f"libpsvae_hip.so ABI {L.psvae_version()} != binding {ABI_VERSION}: rebuild")
    _lib = L
    return L


def check(code: int, what: str = "") -> None:
    if code != 0:
        raise PsvaeError(f"{what}: {lib().psvae_strerror(code).decode()} (code {code})")


def ptr(t: Optional[torch.Tensor], pitched: bool = False) -> Optional[int]:
    """Device pointer of a contiguous fp32/int32 CUDA(HIP) tensor, None for None.  `pitched` also admits a
    2-D row-pitched view (unit inner stride) for the entry points that take a leading dimension."""
    if t is None:
        return None
    if not t.is_cuda:
        raise PsvaeError("psvae kernels need tensors on the GPU (cuda:N); got a CPU tensor and there is no CPU path")
    if pitched and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t.data_ptr()
    if not t.is_contiguous():
        raise PsvaeError("psvae kernels need contiguous tensors")
    return t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def make_params(tensors: Sequence[torch.Tensor]) -> Params:
    assert len(tensors) == NPARAM
    p = Params()
    for i, t in enumerate(tensors):
        if t.dtype != torch.float32:
            raise PsvaeError("parameters must be float32")
        p.p[i] = ptr(t)
    return p


def make_step_opts(params_kmajor: Optional[Params] = None, cond: Optional["Cond"] = None, flags: int = 0):
    """psvae_step_opts for psvae_train_step_f32; the returned object keeps its members alive"""
    o = StepOpts()
    o._keep = (params_kmajor, cond)
    o.params_kmajor = C.pointer(params_kmajor) if params_kmajor is not None else None
    o.cond = C.pointer(cond) if cond is not None else None
    o.flags = int(flags)
    return o


def make_grads(tensors: Sequence[torch.Tensor]) -> Grads:
    assert len(tensors) == NPARAM
    g = Grads()
    for i, t in enumerate(tensors):
        g.g[i] = ptr(t)
    return g


def make_acts(d: dict) -> Acts:
    a = Acts()
    for n in ACT_FIELDS:
        setattr(a, n, ptr(d.get(n)))
    return a


def make_cond(tabs, gtabs, gender, age) -> Optional[Cond]:
    if tabs is None:
        return None
    c = Cond()
    for i in range(3):
        c.tab[i] = ptr(tabs[i])
        c.gtab[i] = ptr(gtabs[i]) if gtabs is not None else None
    c.gender = ptr(gender)
    c.age = ptr(age)
    return c


def scratch_bytes(B: int, D: int, Z: int) -> int:
    return int(lib().psvae_scratch_bytes(C.byref(Dims(B, D, Z))))
