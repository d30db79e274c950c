"""Time the production GEMM (psvae_gemm_f32) per config on the VAE's shapes, 50 launches in one HIP graph."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudo_speaker_vae_amd import ops
dev = "cuda:0"
SHAPES = [("NT", 0, 1024, 512, 512), ("NN", 1, 1024, 512, 512), ("TN", 2, 512, 512, 1024),
          ("TN", 2, 256, 512, 1024), ("NN", 1, 1024, 512, 256), ("TN", 2, 512, 64, 1024), ("NN", 1, 1024, 64, 512),
          ("NT", 0, 1024, 512, 64), ("NT", 0, 1024, 64, 512)]
cfgs = [int(a) for a in sys.argv[1:]] or list(range(1, 10))
for name, kind, M, N, K in SHAPES:
    sa = {0: (M, K), 1: (M, K), 2: (K, M)}[kind]
    sb = {0: (N, K), 1: (K, N), 2: (K, N)}[kind]
    A, B = torch.randn(sa, device=dev), torch.randn(sb, device=dev)
    C = torch.empty(M, N, device=dev)
    line = f"{name} {M:4d}x{N:4d}x{K:4d}: "
    for cfg in cfgs:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): ops.gemm(kind, A, B, tile=cfg, out=C)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(50): ops.gemm(kind, A, B, tile=cfg, out=C)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): g.replay()
        e1.record(); torch.cuda.synchronize()
        line += f" c{cfg}:{e0.elapsed_time(e1) * 1000 / 200:5.1f}"
    print(line, flush=True)
