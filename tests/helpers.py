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


def assert_close(a, b, tol=REL_TOL, what=""):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"


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
