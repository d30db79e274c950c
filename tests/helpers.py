"""Shared helpers for the parity tests (fixture loading, comparison against the oracle)."""
import glob
import os

import numpy as np
import torch

from oracle import psvae_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REL_TOL = 1e-4  # BASELINE.json north_star: "within 1e-4 rel fp32"


def step_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN, "step_*.npz")))


def load(path):
    return dict(np.load(path, allow_pickle=False))


def rel_err(a, b):
    """max |a-b| / max|b| -- scale-relative error of a whole tensor (fp32 GEMM outputs)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / den)


def elementwise_excess(a, b, tol):
    """worst |a-b| / (tol |b| + tol rms(b)): <= 1 when EVERY element is within tol relative to itself plus an absolute
    slack of tol times the tensor's rms.  The slack is what an fp32 dot product owes: its rounding error scales with
    the magnitude of the summed terms (~ the tensor's typical entry), not with a result that cancelled to near zero."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    rms = max(float(np.sqrt(np.mean(b * b))), 1e-30)
    return float((np.abs(a - b) / (tol * np.abs(b) + tol * rms)).max())


def assert_close(a, b, tol=REL_TOL, what="", elementwise=None):
    """two checks: the tensor-scaled one (max error against the largest entry) and an element-wise one in which small
    entries are NOT hidden behind the largest: |a - b| <= tol |b| + tol rms(b) for every element.
    The element-wise check is on for the parity tolerances (tol >= 5e-5: the north_star 1e-4 asserts against the
    oracle / the reference's fixtures).  Tighter calls (1e-5 .. 1e-6) compare two fp32 evaluation ORDERS of the same
    sum: an element that cancels to near zero legitimately differs by ~1e-7 of the summed magnitudes, which is not small
    against |b| + rms(b) times 2e-6 -- those stay bounded against the tensor scale only; so do the bf16 emulation
    checks (a bf16-neighbour flip is 2^-8 of the ELEMENT by construction): they pass elementwise=False."""
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    e = rel_err(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
    if elementwise is None:
        elementwise = tol >= 5e-5
    if elementwise:
        x = elementwise_excess(a, b, tol)
        assert x <= 1.0, f"{what}: element-wise error {x:.2f} x the allowed {tol:.1e} (|b| + rms(b))"


MASK_KEYS = ("h1m", "h2m", "h1s", "h2s", "d1", "d2")


def check_masks(gpu_masks, oracle_acts):
    """ReLU masks of the HIP path may differ from the oracle's only at activations within fp32
    rounding of zero (the ReLU kink), and only for a vanishing fraction of elements."""
    for k in MASK_KEYS:
        ref = oracle_acts[k]
        diff = gpu_masks[k] != (ref > 0)
        n = int(diff.sum())
        assert n <= max(2, 2e-5 * ref.numel()), f"{k}: {n} mask flips"
        if n:
            assert float(ref[diff].abs().max()) <= 2e-5 * float(ref.abs().max()), f"{k}: flip at a non-tiny activation"


def masks_of(acts):
    """ReLU masks (activation > 0) of a workspace's / context's saved activations, on the CPU"""
    return {k: (acts[k] > 0).cpu() for k in MASK_KEYS}


def fixture_params(fx):
    p = O.init_params(int(fx["D"]), int(fx["Z"]), int(fx["pseed"]))
    for k in O.PARAM_KEYS:  # parameters are regenerated from the seed: check the stored checksum
        a = p[k].double().numpy().ravel()
        np.testing.assert_allclose([a.sum(), (a * a).sum()], fx[f"pck/{k}"], rtol=1e-12, atol=1e-12)
    return p


def check_grads_against_fixture(grads, fx, tol=REL_TOL):
    """grads: dict key -> torch/np array of full parameter-gradient tensors."""
    for k in O.PARAM_KEYS:
        g = np.asarray(grads[k].detach().cpu() if torch.is_tensor(grads[k]) else grads[k], dtype=np.float64)
        ref_sample = fx[f"g/{k}"].astype(np.float64)
        scale = max(np.sqrt(fx[f"gck/{k}"][1] / g.size), 1e-30)  # rms of the reference grad tensor
        got = g.reshape(-1)[fx[f"gi/{k}"]] if f"gi/{k}" in fx else g.reshape(-1)
        err = np.abs(got - ref_sample).max() / max(np.abs(ref_sample).max(), scale)
        assert err <= tol, f"grad {k}: rel err {err:.3e}"
        # whole-tensor checksum: sum within tol of the l2 mass
        l2 = np.sqrt(fx[f"gck/{k}"][1])
        assert abs(g.sum() - fx[f"gck/{k}"][0]) <= 10 * tol * max(l2 * np.sqrt(g.size), 1e-30), f"grad sum {k}"
        assert abs((g * g).sum() - fx[f"gck/{k}"][1]) <= 10 * tol * max(fx[f"gck/{k}"][1], 1e-30), f"grad l2 {k}"
