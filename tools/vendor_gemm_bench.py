"""Sanity reference for the hand-written GEMM: what does the vendor library (torch.mm -> rocBLAS/hipBLASLt, fp32) take
for the step's GEMM shapes, 48 launches in one HIP graph, hot L2?  (Not used by the product: comparison only.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudo_speaker_vae_amd import ops
dev = "cuda:0"
torch.backends.cuda.matmul.allow_tf32 = False

def graph_time(fn, n=48):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / (4 * n)

print("shape (M x N x K)        vendor NN   vendor NT   vendor TN | ours NN  ours NT  ours TN (best config)")
for M, N, K in [(2048, 512, 512), (1024, 512, 512), (2048, 512, 256), (1024, 256, 512), (512, 512, 1024), (512, 1024, 1024),
                (4096, 512, 512), (65536, 512, 512)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); Bt = torch.randn(N, K, device=dev)
    At = torch.randn(K, M, device=dev); C = torch.empty(M, N, device=dev)
    v_nn = graph_time(lambda: torch.mm(A, B, out=C))
    v_nt = graph_time(lambda: torch.mm(A, Bt.t(), out=C))
    v_tn = graph_time(lambda: torch.mm(At.t(), B, out=C))
    def best(kind, a, b):
        return min(graph_time(lambda: ops.gemm(kind, a, b, tile=t, out=C)) for t in (1, 2, 6, 7, 9))
    o_nn, o_nt, o_tn = best(1, A, B), best(0, A, Bt), best(2, At, B)
    print(f"{M:6d} x {N:4d} x {K:4d}:   {v_nn:8.1f}   {v_nt:8.1f}   {v_tn:8.1f}  | {o_nn:7.1f}  {o_nt:7.1f}  {o_tn:7.1f}   us", flush=True)
