"""Hot-L2 vs cold-L2 timing of one GEMM shape: a graph of 48 launches that either reuse one operand set (hot: operands
stay in the XCDs' L2s when they fit) or rotate over `nsets` sets (cold L2; MALL/HBM-resident)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudo_speaker_vae_amd import ops
dev = "cuda:0"
KINDS = {"NT": 0, "NN": 1, "TN": 2}
def run(name, M, N, K, cfg, nsets):
    kind = KINDS[name]
    sa = {0: (M, K), 1: (M, K), 2: (K, M)}[kind]
    sb = {0: (N, K), 1: (K, N), 2: (K, N)}[kind]
    sets = [(torch.randn(sa, device=dev), torch.randn(sb, device=dev), torch.empty(M, N, device=dev)) for _ in range(nsets)]
    for A, B, C in sets: ops.gemm(kind, A, B, tile=cfg, out=C)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(48):
            A, B, C = sets[i % nsets]
            ops.gemm(kind, A, B, tile=cfg, out=C)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / (4 * 48)
for spec in sys.argv[1:]:
    name, M, N, K, cfg = spec.split(",")
    M, N, K, cfg = int(M), int(N), int(K), int(cfg)
    print(f"{name} {M}x{N}x{K} c{cfg}: " + "  ".join(f"sets{n}:{run(name, M, N, K, cfg, n):5.1f}" for n in (1, 4, 16)), flush=True)
