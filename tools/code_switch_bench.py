"""Cold-instruction-fetch cost of a GEMM launch: a 48-launch graph over 16 rotating operand sets (cold L2 either way)
that runs ONE config throughout (warm code after the first launch) against the same graph rotating over several
configs (a different kernel every launch, more code than the 64 KB instruction cache holds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudo_speaker_vae_amd import ops
dev = "cuda:0"
KINDS = {"NT": 0, "NN": 1, "TN": 2}
def run(kind, M, N, K, cfgs, nsets=16, launches=48):
    sa = {0: (M, K), 1: (M, K), 2: (K, M)}[kind]
    sb = {0: (N, K), 1: (K, N), 2: (K, N)}[kind]
    sets = [(torch.randn(sa, device=dev), torch.randn(sb, device=dev), torch.empty(M, N, device=dev)) for _ in range(nsets)]
    for c in cfgs:
        ops.gemm(kind, *sets[0][:2], tile=c, out=sets[0][2])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(launches):
            A, B, C = sets[i % nsets]
            ops.gemm(kind, A, B, tile=cfgs[i % len(cfgs)], out=C)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / (6 * launches)
CFGS = [1, 2, 3, 4, 5, 6, 8, 9]
for spec in sys.argv[1:] or ["NT,2048,512,512", "NN,1024,512,512", "TN,512,512,1024"]:
    name, M, N, K = spec.split(",")
    M, N, K = int(M), int(N), int(K)
    single = {c: run(KINDS[name], M, N, K, [c]) for c in CFGS}
    rot = run(KINDS[name], M, N, K, CFGS)
    two = run(KINDS[name], M, N, K, [6, 2])
    mean = sum(single.values()) / len(single)
    print(f"{name} {M}x{N}x{K}: one config per graph " + " ".join(f"c{c}:{t:.1f}" for c, t in single.items()) +
          f" | mean {mean:.2f} us; rotating all eight {rot:.2f} us/launch (+{rot - mean:.2f}); "
          f"alternating c6/c2 {two:.2f} vs {(single[6] + single[2]) / 2:.2f}", flush=True)
