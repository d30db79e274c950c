"""fused (decode_fused.hip) against per-layer decode: equality within summation-order noise, and time.
usage: python tools/decode_ab.py [N D Z] [bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import psvae_oracle as O
from pseudo_speaker_vae_amd import ops

bf = "bf16" in sys.argv
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
N, D, Z = nums if len(nums) == 3 else (1_000_000, 256, 64)
dev = "cuda:0"
p = O.init_params(D, Z, 9)
dec = [p[k].to(dev) for k in O.PARAM_KEYS[12:]]
z = ops.philox_normal(N * Z, 5, 0, dev).reshape(N, Z)
fn = ops.decode_bf16 if bf else ops.decode


def run(unfused):
    if unfused:
        os.environ["PSVAE_DECODE_UNFUSED"] = "1"
    else:
        os.environ.pop("PSVAE_DECODE_UNFUSED", None)
    out = fn(z, dec, D)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        out = fn(z, dec, D)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return out, min(ts)


a, ta = run(True)
b, tb = run(False)
err = float((a - b).abs().max() / a.abs().max())
fl = 2.0 * N * (Z * 512 + 512 * 512 + 512 * D)
print(f"{'bf16' if bf else 'fp32'} N={N} D={D} Z={Z}: per-layer {ta * 1e3:.3f} ms ({fl / ta * 1e-12:.1f} TF)  fused {tb * 1e3:.3f} ms "
      f"({fl / tb * 1e-12:.1f} TF)  max rel diff {err:.2e}  mean abs diff / rms {float((a - b).abs().mean() / a.pow(2).mean().sqrt()):.2e}", flush=True)
oracle = O.decode_bf16 if bf else O.decode
for lo, hi in ((0, 256), (N - 100, N)):
    ref = oracle(p, z[lo:hi].cpu())
    print(f"fused vs oracle rows {lo}:{hi}:", float((b[lo:hi].cpu() - ref).abs().max() / ref.abs().max()))
