"""fused (decode_fused.hip) against per-layer decode: equality within fp32 summation-order noise, and time.
usage: python tools/decode_ab.py [N] [D] [Z]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import psvae_oracle as O
from pseudo_speaker_vae_amd import ops

N, D, Z = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1_000_000, 256, 64)
dev = "cuda:0"
p = O.init_params(D, Z, 9)
dec = [p[k].to(dev) for k in O.PARAM_KEYS[12:]]
z = ops.philox_normal(N * Z, 5, 0, dev).reshape(N, Z)


def run(unfused):
    if unfused:
        os.environ["PSVAE_DECODE_UNFUSED"] = "1"
    else:
        os.environ.pop("PSVAE_DECODE_UNFUSED", None)
    out = ops.decode(z, dec, D)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        out = ops.decode(z, dec, D)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return out, min(ts)


a, ta = run(True)
b, tb = run(False)
err = float((a - b).abs().max() / a.abs().max())
fl = 2.0 * N * (Z * 512 + 512 * 512 + 512 * D)
print(f"N={N} D={D} Z={Z}: per-layer {ta * 1e3:.2f} ms ({fl / ta * 1e-12:.1f} TF)  fused {tb * 1e3:.2f} ms "
      f"({fl / tb * 1e-12:.1f} TF)  max rel diff {err:.2e}", flush=True)
ref = O.decode(p, z[:256].cpu())
print("fused vs oracle (256 rows):", float((b[:256].cpu() - ref).abs().max() / ref.abs().max()))
tail = O.decode(p, z[N - 100:].cpu())
print("fused vs oracle (last 100 rows):", float((b[N - 100:].cpu() - tail).abs().max() / tail.abs().max()))
