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
    assert lib.psvae_version() == 4
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


def test_launch_plan_text_roundtrip_and_errors(lib):
    """psvae_set_plan / psvae_describe_plan are host-only: a plan survives the text round trip, garbage is refused."""
    from pseudo_speaker_vae_amd import _ffi
    d = _ffi.Dims(777, 256, 64)
    buf = C.create_string_buffer(1024)
    assert lib.psvae_describe_plan(C.byref(d), buf, 1024) == 0 and buf.value == b"untuned"
    plan = b"F1:0/6/0/1/1 F2:0/2/0/2/2 F3:0/8/0/0/0 F5:0/4/0/2/1 F6:0/0/0/0/0 F7:0/1/0/1/2 " \
           b"B1:2/6/2/2/2 B2:1/6/1/1/1 B3:8/8/4/2/1 B4:8/1/1/2/2 B5:6/6/1/1/1 B6:1/1/1/2/1"
    assert lib.psvae_set_plan(C.byref(d), plan) == 0
    assert lib.psvae_describe_plan(C.byref(d), buf, 1024) == 0 and buf.value == plan
    other = _ffi.Dims(778, 256, 64)  # plans are per (B, D, Z)
    assert lib.psvae_describe_plan(C.byref(other), buf, 1024) == 0 and buf.value == b"untuned"
    for bad in (b"F1:0/6/0/1", b"nonsense", b"F1:0/99/0/1/1 " + plan[13:], plan.replace(b"B3:8/8/4", b"B3:8/8/9")):
        assert lib.psvae_set_plan(C.byref(d), bad) == -4, bad
    assert lib.psvae_set_plan(C.byref(d), None) == -1
    assert lib.psvae_set_plan(C.byref(d), b"untuned") == 0
    assert lib.psvae_describe_plan(C.byref(d), buf, 1024) == 0 and buf.value == b"untuned"
    assert lib.psvae_tune_train_step_f32(C.byref(d), None, None, None, None, None, None, None, None, None, 0, None, None) == -4


def test_header_is_plain_c_and_links_against_the_library(tmp_path, lib):
    """include/psvae.h is a C header (C99, -pedantic clean) and a C program linked against libpsvae_hip.so resolves the
    host-only entry points (no GPU needed: version, strerror, scratch sizes, plan text API)"""
    import shutil
    import subprocess
    from pseudo_speaker_vae_amd import _ffi
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "c_abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "psvae.h"
int main(void) {
    psvae_dims d = {1024, 256, 64};
    char buf[64];
    if (psvae_version() != PSVAE_ABI_VERSION) return 1;
    if (strcmp(psvae_strerror(PSVAE_OK), "ok") != 0) return 2;
    if (psvae_scratch_bytes(&d) == 0 || psvae_decode_scratch_bytes(10) == 0) return 3;
    if (psvae_describe_plan(&d, buf, sizeof buf) != 0 || strcmp(buf, "untuned") != 0) return 4;
    if (psvae_set_plan(&d, "garbage") != PSVAE_ERR_ARG) return 5;
    if (psvae_train_step_f32(&d, NULL, NULL, NULL, 0, 0, NULL, NULL, NULL, NULL, NULL, 1.0f, NULL, NULL, NULL, NULL,
                             PSVAE_PHASE_ALL, NULL, NULL) != PSVAE_ERR_NULL) return 6;
    {
        psvae_step_opts o = {NULL, NULL, PSVAE_COS_LOSS};
        psvae_adam_state st = {1e-3, 0, {0u, 0u}};
        if (o.flags != 2u || sizeof st != 24 || psvae_last_step_launches() != 0) return 7;
        if (psvae_adam_step_dev_f32(NULL, NULL, 0.9, 0.999, 1e-8, 0.0, 1.0, NULL) != PSVAE_ERR_NULL) return 8;
        if (psvae_transpose_params_f32(&d, NULL, NULL, NULL) != PSVAE_ERR_NULL) return 9;
    }
    printf("c abi ok\n");
    return 0;
}
''')
    exe = tmp_path / "c_abi"
    libdir = os.path.dirname(_ffi.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(REPO, "include"), str(src),
           "-o", str(exe), "-L", libdir, "-l:libpsvae_hip.so", "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "c abi ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_build_refuses_a_spilling_asm_load_kernel():
    """build.py's guard for kernels with inline-asm register loads (a spilled destination register would be stored before
    the data lands); no kernel needs it at present, the mechanism stays tested"""
    import pytest as _pytest
    from pseudo_speaker_vae_amd import build
    ok = ("x.hip:1:1: remark: Function Name: _ZN5psvae24decode_fused_bf16_kernelILi2ELi16ELi8EEEv [-Rpass-analysis=kernel-resource-usage]\n"
          "x.hip:1:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]\n"
          "x.hip:1:1: remark: Function Name: _ZN5psvae19decode_fused_kernelILi8ELi32EEEv [-Rpass-analysis=kernel-resource-usage]\n"
          "x.hip:1:1: remark:     VGPRs Spill: 37 [-Rpass-analysis=kernel-resource-usage]\n")
    build.NO_SPILL["x.hip"] = "decode_fused_bf16_kernel"
    try:
        build._check_no_spill("x.hip", ok)  # the other kernel's loads are compiler-tracked: its spills are legal
        bad = ok.replace("VGPRs Spill: 0", "VGPRs Spill: 11")
        with _pytest.raises(RuntimeError, match="spills 11"):
            build._check_no_spill("x.hip", bad)
    finally:
        del build.NO_SPILL["x.hip"]
