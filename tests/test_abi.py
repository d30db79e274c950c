"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/psvae.h declares; argument errors are reported as codes (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pseudo_speaker_vae_amd import _ffi, build
    build.build(verbose=False)
    return _ffi.lib()


def test_header_symbols_all_exported(lib):
    from pseudo_speaker_vae_amd import _ffi
    hdr = open(os.path.join(REPO, "include", "psvae.h")).read()
    declared = set(re.findall(r"\b(psvae_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_ffi.SYMBOLS), declared ^ set(_ffi.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_version_and_strerror(lib):
    assert lib.psvae_version() == 1
    assert b"NULL" in lib.psvae_strerror(-1)
    assert lib.psvae_strerror(0) == b"ok"


def test_argument_errors_before_any_launch(lib):
    from pseudo_speaker_vae_amd import _ffi
    assert lib.psvae_philox_normal_f32(None, 16, 0, 0, None) == -1
    assert lib.psvae_gemm_f32(0, 4, 4, 4, None, 4, None, 4, None, 4, None, None, 0, 0, None) == -1
    assert lib.psvae_gemm_f32(7, 4, 4, 4, 16, 4, 16, 4, 16, 4, None, None, 0, 0, None) == -4
    assert lib.psvae_gemm_f32(0, 4, 4, 4, 16, 4, 16, 4, 16, 4, None, None, 0, 99, None) == -4
    assert lib.psvae_debug_force_tile(99) == -4 and lib.psvae_debug_force_tile(0) == 0
    assert lib.psvae_gemm_f32(0, 0, 4, 4, 16, 4, 16, 4, 16, 4, None, None, 0, 0, None) == -2
    d = _ffi.Dims(0, 256, 64)
    assert lib.psvae_scratch_bytes(C.byref(d)) == 0
    d = _ffi.Dims(1024, 256, 64)
    assert lib.psvae_scratch_bytes(C.byref(d)) > 1024 * 512 * 4 * 6
    assert lib.psvae_decode_scratch_bytes(10) >= 2 * 10 * 512 * 4
    assert lib.psvae_forward_f32(C.byref(d), None, None, None, None, 0, 0, None, None, None, None, None, None, 0, None) == -1


def test_cpu_tensors_are_refused_loudly():
    import torch
    from pseudo_speaker_vae_amd import _ffi, ops
    with pytest.raises(_ffi.PsvaeError, match="no CPU path"):
        ops.gemm(0, torch.zeros(4, 4), torch.zeros(4, 4))
