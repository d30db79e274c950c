"""GPU kernel tests (run with -m gpu on an MI355X): each HIP kernel through the C-ABI vs a plain
fp32/fp64 torch restatement of the same op."""
import numpy as np
import pytest
import torch

from helpers import assert_close, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from pseudo_speaker_vae_amd import ops as o
    return o


def _ref_gemm(kind, A, B):
    A, B = A.double().cpu(), B.double().cpu()
    if kind == 0:
        return A @ B.t()
    return A @ B if kind == 1 else A.t() @ B


TILES = list(range(1, 10))  # every GEMM config id (csrc/gemm_f32.hip)


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("tile", TILES)
def test_gemm_tiles_exact_shapes(ops, kind, tile):
    """integer-valued operands: every product and sum is exact in fp32 -> must match bit for bit.
    Asymmetric B catches a transposed C write (cdna guide section 3)."""
    torch.manual_seed(kind * 16 + tile)
    M, N, K = 256, 192, 160
    dev = "cuda:0"
    shapes = {0: ((M, K), (N, K)), 1: ((M, K), (K, N)), 2: ((K, M), (K, N))}[kind]
    A = torch.randint(-3, 4, shapes[0], device=dev).float()
    B = torch.randint(-3, 4, shapes[1], device=dev).float()
    Cm = ops.gemm(kind, A, B, tile=tile)
    torch.cuda.synchronize()
    ref = _ref_gemm(kind, A, B).float()
    assert torch.equal(Cm.cpu(), ref), f"kind {kind} tile {tile:#x}: max diff {(Cm.cpu() - ref).abs().max()}"


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("shape", [(1, 20, 512), (37, 64, 64), (130, 262, 70), (8, 512, 784), (33, 17, 5), (64, 50, 31), (100, 36, 1000)])
@pytest.mark.parametrize("tile", [0, 1, 2, 5, 6, 8])
def test_gemm_ragged_shapes(ops, kind, shape, tile):
    """edge tiles, K not a multiple of 32, leading dimensions that are not multiples of 4 (scalar path)."""
    M, N, K = shape
    torch.manual_seed(M * 7 + N * 3 + K)
    dev = "cuda:0"
    shapes = {0: ((M, K), (N, K)), 1: ((M, K), (K, N)), 2: ((K, M), (K, N))}[kind]
    A = torch.randn(shapes[0], device=dev)
    B = torch.randn(shapes[1], device=dev)
    Cm = ops.gemm(kind, A, B, tile=tile)
    torch.cuda.synchronize()
    assert_close(Cm.cpu(), _ref_gemm(kind, A, B), 2e-6, f"gemm kind {kind} {shape} tile {tile:#x}")


def test_gemm_epilogues(ops):
    dev = "cuda:0"
    torch.manual_seed(0)
    M, N, K = 96, 160, 128
    A, W, bias = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
    ref = (A.double() @ W.double().t() + bias.double()).cpu()
    assert_close(ops.gemm(0, A, W, ops.EPI_BIAS, bias).cpu(), ref, 2e-6, "bias")
    assert_close(ops.gemm(0, A, W, ops.EPI_BIAS_RELU, bias).cpu(), ref.clamp_min(0), 2e-6, "bias+relu")
    act = torch.randn(M, N, device=dev).clamp_min(0)
    Wn = torch.randn(K, N, device=dev)
    refm = ((A.double() @ Wn.double()) * (act.double() > 0)).cpu()
    assert_close(ops.gemm(1, A, Wn, ops.EPI_MASK, act).cpu(), refm, 2e-6, "relu mask")
    base = torch.randn(M, N, device=dev)
    out = base.clone()
    ops.gemm(1, A, Wn, ops.EPI_ACCUM, out=out)
    assert_close(out.cpu(), (base.double() + A.double() @ Wn.double()).cpu(), 2e-6, "accumulate")
    # wgrad with fused bias-gradient column sums, every tile config
    dY, X = torch.randn(200, 96, device=dev), torch.randn(200, 72, device=dev)
    for tile in range(0, 10):
        dW, db = ops.gemm(2, dY, X, want_colsum=True, tile=tile)
        assert_close(dW.cpu(), (dY.double().t() @ X.double()).cpu(), 2e-6, f"wgrad {tile:#x}")
        assert_close(db.cpu(), dY.double().sum(0).cpu(), 2e-6, f"bias grad {tile:#x}")


def test_gemm_bad_arguments(ops):
    from pseudo_speaker_vae_amd import _ffi
    A = torch.randn(4, 4, device="cuda:0")
    with pytest.raises(_ffi.PsvaeError):
        ops.gemm(0, A, A, tile=77)
    with pytest.raises(_ffi.PsvaeError):
        ops.gemm(0, A, A, epi=ops.EPI_BIAS, aux=None)
    with pytest.raises(_ffi.PsvaeError):
        ops.gemm(0, A.cpu(), A.cpu())


def test_philox_matches_oracle_stream(ops):
    from oracle import psvae_oracle as O
    for n, seed, off in ((4096, 0, 0), (1000, 1234, 77), (7, 2**40 + 5, 2**33), (65536, 42, 1 << 20)):
        got = ops.philox_normal(n, seed, off, "cuda:0").cpu().numpy()
        want = O.philox_normal(n, seed, off)
        assert np.abs(got - want).max() < 2e-5, (n, seed, off)
    big = ops.philox_normal(1 << 22, 7, 0, "cuda:0")
    assert abs(big.mean().item()) < 3e-3 and abs(big.std().item() - 1) < 3e-3
    assert abs((big ** 4).mean().item() - 3) < 0.05


def test_adam_matches_oracle(ops):
    from oracle import psvae_oracle as O
    torch.manual_seed(0)
    n = 100003
    p0, m0, v0 = torch.randn(n), torch.zeros(n), torch.zeros(n)
    p, m, v = p0.cuda(), m0.cuda(), v0.cuda()
    pc, mc, vc = p0.clone(), m0.clone(), v0.clone()
    for step in range(1, 5):
        g = torch.randn(n)
        ops.adam_step(p, g.cuda(), m, v, step, 1e-3, weight_decay=0.01, grad_scale=0.5)
        pc, mc, vc = O.adam_step(pc, 0.5 * g, mc, vc, step, 1e-3, weight_decay=0.01)
    assert_close(p.cpu(), pc, 1e-6, "adam param")
    assert_close(v.cpu(), vc, 1e-6, "adam v")


def test_philox_fast_stream_is_the_exact_stream_to_hardware_accuracy():
    DEV = "cuda:0"
    """psvae_philox_normal_fast_f32 (Box-Muller on v_log / v_sqrt / v_sin / v_cos): the same Philox blocks as
    psvae_philox_normal_f32, every value within 2e-6 of the stream's largest magnitude; standard-normal moments; and what
    the samplers draw inside their kernels (tests/test_gpu_properties.py, test_gpu_decode_fused.py inject it bit for bit)"""
    from pseudo_speaker_vae_amd import ops
    n = 1 << 20
    a = ops.philox_normal(n, 77, 1234, DEV).cpu().double()
    b = ops.philox_normal(n, 77, 1234, DEV, fast=True).cpu().double()
    assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())
    assert abs(float(b.mean())) < 4e-3 and abs(float(b.var()) - 1.0) < 6e-3
    assert abs(float((b ** 4).mean()) - 3.0) < 5e-2
    odd = ops.philox_normal(1001, 77, 1234, DEV, fast=True).cpu().double()   # ragged tail, same values
    assert torch.equal(odd, b[:1001])
