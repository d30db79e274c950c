"""The bf16 weight-gradient launch (csrc/wgrad_bf16.hip) by itself, at full batch sizes, on a size-independent property:
with integer-valued operands every product and every fp32 partial sum is an integer below 2^24, so the result must equal
the integer matrix product BIT FOR BIT -- whatever the tile, the batch split, the order of the slab sums or the raggedness
of the last tile.  (The pass-level parity of the same launch against the oracle's bf16 emulation: tests/test_gpu_bf16.py.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ints(shape, lo, hi, seed):
    return torch.randint(lo, hi + 1, shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("B,out,inn,split,form", [
    (4096, 512, 512, 5, "default"),        # cfg3's big layers as the pass runs them
    (4100, 256, 512, 3, "default"),        # ragged last tile in the last slice
    (8192, 512, 64, 8, "default"),         # half-width tiles, the deepest split
    (1000, 6, 512, 1, "default"),          # a conditional table's gradient: 6 rows (pitch 8), no split
    (70, 64, 256, 4, "default"),           # two k tiles for four slices: empty slices write zero slabs
    (4096, 512, 256, 5, "loader-wave"),    # the alternative forms of the launch
    (1030, 128, 128, 2, "loader-wave"),
    (4096, 512, 512, 5, "three-buffers"),
    (4096, 512, 512, 5, "64-row-tiles"),
    (1030, 256, 128, 3, "64-row-tiles"),
])
def test_wgrad_bf16_is_exact_on_integer_operands(B, out, inn, split, form, monkeypatch):
    from pseudo_speaker_vae_amd import ops
    if form == "loader-wave":
        monkeypatch.setenv("PSVAE_WGRAD_LW", "1")
    if form == "three-buffers":
        monkeypatch.setenv("PSVAE_WGRAD_NBUF", "3")
    if form == "64-row-tiles":
        monkeypatch.setenv("PSVAE_WGRAD_BK", "64")
    dY = _ints((B, out), -3, 3, 1)
    X = _ints((B, inn), -3, 3, 2)
    # pitches: multiples of 8 elements (the one-hot matrix of the pass is [B][8] with 6 valid columns)
    ldy = (out + 7) // 8 * 8
    dYd = torch.zeros(B, ldy, dtype=torch.bfloat16, device=DEV)
    dYd[:, :out] = dY.to(torch.bfloat16)
    dW, db = ops.wgrad_bf16(dYd[:, :out], X.to(torch.bfloat16).to(DEV), split=split)
    torch.cuda.synchronize()
    ref = dY.t().to(torch.int64) @ X.to(torch.int64)
    assert int(ref.abs().max()) < 2 ** 24
    assert torch.equal(dW.cpu().to(torch.int64), ref), "dW differs from the integer product"
    assert torch.equal(dW.cpu(), ref.to(torch.float32))
    assert torch.equal(db.cpu().to(torch.int64), dY.sum(0).to(torch.int64)), "db differs from the integer column sums"


def test_wgrad_bf16_rejects_bad_layouts():
    from pseudo_speaker_vae_amd import _ffi, ops
    dY = torch.zeros(64, 36, dtype=torch.bfloat16, device=DEV)   # pitch 36: not a multiple of 8
    X = torch.zeros(64, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_ffi.PsvaeError):
        ops.wgrad_bf16(dY, X)
    with pytest.raises(_ffi.PsvaeError):
        ops.wgrad_bf16(X, X, split=9)
