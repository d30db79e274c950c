import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from oracle import psvae_oracle as O
from pseudo_speaker_vae_amd import ops
D, Z = 256, 64
p = O.init_params(D, Z, 9)
dec = [p[k].to("cuda:0") for k in O.PARAM_KEYS[12:]]
for N in (4096, 8192, 16384):
    z = ops.philox_normal(N * Z, 5, 0, "cuda:0").reshape(N, Z)
    for fused in (False, True):
        os.environ["PSVAE_DECODE_FUSED_MIN"] = "0" if fused else "100000000"
        out = ops.decode_bf16(z, dec, D)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                ops.decode_bf16(z, dec, D, out=out)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        print(N, "fused" if fused else "per-layer", f"{(time.perf_counter() - t0) / 200 * 1e6:.1f} us per decode (incl. weight cast/pack launches)")
