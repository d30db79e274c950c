"""The fused decode (csrc/decode_fused.hip: VAEModel.decode, /root/reference/ps_vae/model.py:65-69, all three layers in
one kernel for N >= 16,384 rows) against the oracle and against the per-layer path it replaces.  The oracle decodes
slices (it is a CPU loop over 1,024 output columns per row); the per-layer HIP path -- itself oracle-checked in
test_gpu_parity.py / test_gpu_api.py -- is the all-rows checker."""
import os

import pytest
import torch

from helpers import assert_close
from oracle import psvae_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _decode(ops, z, dec, D, fused, **kw):
    if fused:
        os.environ.pop("PSVAE_DECODE_UNFUSED", None)
    else:
        os.environ["PSVAE_DECODE_UNFUSED"] = "1"
    try:
        out = ops.decode(z, dec, D, **kw)
        torch.cuda.synchronize()
        return out, ops.last_decode_launches()
    finally:
        os.environ.pop("PSVAE_DECODE_UNFUSED", None)


@pytest.mark.parametrize("D,Z,N", [(256, 64, 16384), (256, 64, 16384 + 37), (512, 128, 20001), (128, 32, 16400),
                                   (512, 32, 16385), (128, 128, 17000), (256, 32, 30000), (256, 128, 16447),
                                   (512, 64, 16391), (128, 64, 65536 + 1)])
def test_fused_decode_matches_oracle_and_per_layer_path(D, Z, N):
    from pseudo_speaker_vae_amd import ops
    p = O.init_params(D, Z, N % 7)
    dec = [p[k].to(DEV) for k in O.PARAM_KEYS[12:]]
    z = ops.philox_normal(N * Z, 3, 0, DEV).reshape(N, Z)
    out = torch.full((N + 1, D), float("nan"), device=DEV)  # one guard row behind the output
    fused, nf = _decode(ops, z, dec, D, True, out=out[:N])
    assert nf == 4, "the fused kernel did not run"
    assert torch.isnan(out[N]).all(), "wrote past the last row"
    assert torch.isfinite(fused).all()
    per_layer, npl = _decode(ops, z, dec, D, False)
    assert npl == 3 * ((N + 65535) // 65536)
    assert_close(fused.cpu(), per_layer.cpu(), 2e-6, "fused vs per-layer, all rows")
    for lo, hi in ((0, 70), (N // 2 - 5, N // 2 + 60), (N - 45, N)):  # incl. the ragged last tile
        ref = O.decode(p, z[lo:hi].cpu())
        assert_close(fused[lo:hi].cpu(), ref, 1e-4, f"rows {lo}:{hi} vs oracle")


def test_fused_decode_with_normalised_output():
    """normalize_decoder_output (model.py:38,69): rows scaled to unit L2 norm after the fused kernel"""
    from pseudo_speaker_vae_amd import ops
    from pseudo_speaker_vae_amd._ffi import NORMALIZE_DECODER as PSVAE_NORMALIZE_DECODER
    D, Z, N = 256, 64, 70001  # two normalisation chunks
    p = O.init_params(D, Z, 1)
    dec = [p[k].to(DEV) for k in O.PARAM_KEYS[12:]]
    z = ops.philox_normal(N * Z, 8, 0, DEV).reshape(N, Z)
    out, n = _decode(ops, z, dec, D, True, flags=PSVAE_NORMALIZE_DECODER)
    assert n == 4 + 2
    assert_close(out.norm(dim=1).cpu(), torch.ones(N), 1e-5, "unit rows")
    for lo, hi in ((0, 64), (65530, 65600), (N - 30, N)):
        ref = O.decode(p, z[lo:hi].cpu())
        ref = ref / ref.norm(dim=1, keepdim=True)
        assert_close(out[lo:hi].cpu(), ref, 1e-4, f"rows {lo}:{hi} vs oracle")


def test_small_and_conditional_decodes_keep_the_per_layer_path():
    """below 16,384 rows the persistent kernel cannot fill the chip, and the conditional table rides the decoder.0
    epilogue of the per-layer path: both must not take the fused kernel"""
    from pseudo_speaker_vae_amd import ops
    D, Z = 256, 64
    p = O.init_params(D, Z, 2)
    dec = [p[k].to(DEV) for k in O.PARAM_KEYS[12:]]
    z = ops.philox_normal(4096 * Z, 8, 0, DEV).reshape(4096, Z)
    _, n = _decode(ops, z, dec, D, True)
    assert n == 3
    N = 20000
    z = ops.philox_normal(N * Z, 9, 0, DEV).reshape(N, Z)
    g = torch.Generator().manual_seed(0)
    tab = torch.randn(6, 512, generator=g).to(DEV)
    gender = torch.randint(0, 2, (N,), generator=g, dtype=torch.int32).to(DEV)
    age = torch.randint(0, 4, (N,), generator=g, dtype=torch.int32).to(DEV)
    out, n = _decode(ops, z, dec, D, True, cond_tab=tab, gender=gender, age=age)
    assert n == 3
    plain, n2 = _decode(ops, z, dec, D, True)
    assert n2 == 4
    assert not torch.equal(out, plain)


def test_fused_decode_through_the_model_api():
    """VAEModel.decode / .sample (model.py:65-69, inference.py:25) at sampling sizes go through the fused kernel and
    match the oracle"""
    from pseudo_speaker_vae_amd import ops
    from pseudo_speaker_vae_amd.model import VAEModel
    D, Z, N = 256, 64, 50_000
    torch.manual_seed(0)
    m = VAEModel(D, Z).to(DEV)
    z = torch.randn(N, Z, device=DEV)
    with torch.no_grad():
        out = m.decode(z)
    torch.cuda.synchronize()
    assert ops.last_decode_launches() == 4
    p = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = O.decode(p, z[-100:].cpu())
    assert_close(out[-100:].cpu(), ref, 1e-4, "model.decode tail vs oracle")


BF16_TOL = 4e-3  # tests/test_gpu_bf16.py: a bf16-neighbour flip of a hidden activation is 2^-8 of that element


@pytest.mark.parametrize("D,Z,N", [(256, 64, 32768), (256, 64, 32768 + 77), (128, 32, 40001), (512, 64, 33000),
                                   (128, 64, 70001), (512, 32, 32769), (256, 32, 131072 + 5), (256, 128, 40000), (512, 128, 33001),
                                   (128, 128, 32768 + 129)])
def test_fused_bf16_decode_matches_emulation_and_per_layer_path(D, Z, N):
    """psvae_decode_bf16 at N >= 32,768 rows runs the fused bf16 kernel (same rounding points as the per-layer path: z,
    weights and the two hidden activations to bf16 round-to-nearest-even, fp32 accumulation, biases and x_hat); against
    the oracle's bf16 emulation on slices and the per-layer path on all rows.  The two HIP paths sum in different orders,
    so a hidden activation near a rounding boundary may land on the other bf16 neighbour: the bound is the bf16 one, and
    the mean difference shows such flips are rare."""
    from pseudo_speaker_vae_amd import ops
    p = O.init_params(D, Z, N % 5)
    dec = [p[k].to(DEV) for k in O.PARAM_KEYS[12:]]
    z = ops.philox_normal(N * Z, 4, 0, DEV).reshape(N, Z)
    out = torch.full((N + 1, D), float("nan"), device=DEV)
    os.environ.pop("PSVAE_DECODE_UNFUSED", None)
    fused = ops.decode_bf16(z, dec, D, out=out[:N])
    torch.cuda.synchronize()
    assert torch.isnan(out[N]).all(), "wrote past the last row"
    assert torch.isfinite(fused).all()
    os.environ["PSVAE_DECODE_UNFUSED"] = "1"
    try:
        per_layer = ops.decode_bf16(z, dec, D)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("PSVAE_DECODE_UNFUSED", None)
    assert not torch.equal(fused, per_layer), "both calls took the same path"
    # the two tile heights of the fused kernel (64 rows, two workgroups per CU: the default; 128 rows, one): the same
    # arithmetic per row -- identical bits
    os.environ["PSVAE_DECODE_BF16_RG"] = "8"
    try:
        tall = ops.decode_bf16(z, dec, D)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("PSVAE_DECODE_BF16_RG", None)
    assert torch.equal(tall, fused), "64-row and 128-row tiles disagree"
    assert_close(fused.cpu(), per_layer.cpu(), BF16_TOL, "fused vs per-layer bf16, all rows", elementwise=False)
    assert float((fused - per_layer).abs().mean() / per_layer.pow(2).mean().sqrt()) < 2e-5
    for lo, hi in ((0, 140), (N // 2 - 3, N // 2 + 130), (N - 150, N)):  # incl. the ragged last tile
        ref = O.decode_bf16(p, z[lo:hi].cpu(), False)
        assert_close(fused[lo:hi].cpu(), ref, BF16_TOL, f"rows {lo}:{hi} vs bf16 emulation", elementwise=False)


def test_fused_bf16_decode_small_calls_keep_the_per_layer_path():
    from pseudo_speaker_vae_amd import ops
    D, Z, N = 256, 128, 20000
    p = O.init_params(D, Z, 3)
    dec = [p[k].to(DEV) for k in O.PARAM_KEYS[12:]]
    z = ops.philox_normal(N * Z, 4, 0, DEV).reshape(N, Z)
    a = ops.decode_bf16(z, dec, D)
    os.environ["PSVAE_DECODE_UNFUSED"] = "1"
    try:
        b = ops.decode_bf16(z, dec, D)
    finally:
        os.environ.pop("PSVAE_DECODE_UNFUSED", None)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    ref = O.decode_bf16(p, z[-64:].cpu(), False)
    assert_close(a[-64:].cpu(), ref, BF16_TOL, "tail vs bf16 emulation", elementwise=False)


@pytest.mark.parametrize("bf16,D,Z,N", [(False, 256, 64, 16384 + 37), (False, 512, 128, 20001), (False, 128, 32, 16400),
                                        (True, 256, 64, 32768 + 77), (True, 512, 128, 40001), (True, 128, 32, 33000)])
def test_sample_decode_draws_z_inside_the_fused_kernel(bf16, D, Z, N):
    """psvae_sample_decode_*: unconditional_synthesis (inference.py:22-25) as ONE call -- z ~ N(0, I) drawn from the Philox
    stream INSIDE the fused decode kernel.  The z it reports must be the stream psvae_philox_normal_f32 writes (same
    blocks, same element order), and x_hat must be, bit for bit, what the same kernel decodes from that z read from
    memory; rows past N are never written (guard rows behind both outputs)."""
    from pseudo_speaker_vae_amd import ops
    p = O.init_params(D, Z, 2)
    dec = [p[k].to(DEV) for k in O.PARAM_KEYS[12:]]
    seed, off = 99, 12345
    xg = torch.full((N + 1, D), float("nan"), device=DEV)
    zg = torch.full((N + 1, Z), float("nan"), device=DEV)
    xh, z = ops.sample_decode(N, Z, dec, D, seed, off, bf16=bf16, out=xg[:N], z_out=zg[:N])
    torch.cuda.synchronize()
    assert ops.last_decode_launches() == 4 or bf16          # (the fp32 entry counts its launches: fused = 4)
    assert torch.isnan(xg[N]).all() and torch.isnan(zg[N]).all(), "wrote past the last row"
    z_ref = ops.philox_normal(N * Z, seed, off, DEV).reshape(N, Z)
    assert_close(z.cpu(), z_ref.cpu(), 1e-6, "in-kernel Philox draw vs psvae_philox_normal_f32")
    assert_close(z[:64].cpu(), torch.from_numpy(O.philox_normal(64 * Z, seed, off).reshape(64, Z)), 1e-5, "draw vs oracle")
    ref = (ops.decode_bf16 if bf16 else ops.decode)(z.contiguous(), dec, D)
    torch.cuda.synchronize()
    assert torch.equal(xh, ref), "decode of the drawn z differs from decode of the same z read from memory"
    # without a z buffer the result is the same (z never touches memory)
    xh2, none = ops.sample_decode(N, Z, dec, D, seed, off, bf16=bf16)
    assert none is None and torch.equal(xh2, xh)
    # another offset draws other rows
    xh3, _ = ops.sample_decode(N, Z, dec, D, seed, off + N * Z // 4, bf16=bf16)
    assert not torch.equal(xh3, xh)


@pytest.mark.parametrize("bf16", [False, True])
def test_sample_decode_on_small_batches_draws_first(bf16):
    """below the fused kernels' row counts the entry draws z with the Philox launch, then decodes: same numbers"""
    from pseudo_speaker_vae_amd import ops
    D, Z, N = 256, 64, 300
    p = O.init_params(D, Z, 4)
    dec = [p[k].to(DEV) for k in O.PARAM_KEYS[12:]]
    xh, z = ops.sample_decode(N, Z, dec, D, 5, 40, bf16=bf16, want_z=True)
    z_ref = ops.philox_normal(N * Z, 5, 40, DEV).reshape(N, Z)
    assert torch.equal(z, z_ref)
    ref = (ops.decode_bf16 if bf16 else ops.decode)(z_ref, dec, D)
    assert torch.equal(xh, ref)
    xh2, none = ops.sample_decode(N, Z, dec, D, 5, 40, bf16=bf16)
    assert none is None and torch.equal(xh2, xh)
    if not bf16:
        assert_close(xh.cpu(), O.decode(p, z_ref.cpu()), 1e-4, "vs oracle")
