"""Does a power-of-two row pitch cost bandwidth?  Time one GEMM shape with padded operand pitches (graph of 50 launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudo_speaker_vae_amd import ops
dev = "cuda:0"
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for name, kind, M, N, K in [("NN", 1, 2048, 512, 512), ("NT", 0, 2048, 512, 512), ("TN", 2, 512, 1024, 1024), ("NN", 1, 1024, 512, 512)]:
    line = f"{name} {M}x{N}x{K} c{cfg}:"
    for pad in (0, 16, 32, 48, 64, 96):
        sa = {0: (M, K), 1: (M, K), 2: (K, M)}[kind]
        sb = {0: (N, K), 1: (K, N), 2: (K, N)}[kind]
        Ab = torch.randn(sa[0], sa[1] + pad, device=dev); Bb = torch.randn(sb[0], sb[1] + pad, device=dev)
        A, B = Ab[:, :sa[1]], Bb[:, :sb[1]]
        Cb = torch.empty(M, N + pad, device=dev); C = Cb[:, :N]
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
        line += f"  pad{pad}:{e0.elapsed_time(e1) * 1000 / 200:5.1f}"
    print(line, flush=True)
