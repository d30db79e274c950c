"""psvae_gather_rows_f32 (batch assembly from the packed embedding table): time and bytes/s for a training
batch, a large batch and a whole shuffled epoch of the CommonVoice-sized table (674,068 x 256 fp32 = 690 MB)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudo_speaker_vae_amd.data import PackedEmbeddingTable
dev = "cuda:0"
N, D = 674068, 256
tab = PackedEmbeddingTable(torch.randn(N, D), device=dev)
g = torch.Generator().manual_seed(0)
for B in (1024, 4096, 65536, N):
    idx = torch.randperm(N, generator=g)[:B].to(dev)
    out = torch.empty(B, D, device=dev)
    for _ in range(3): tab.batch(idx, out)
    torch.cuda.synchronize()
    reps = 200 if B <= 65536 else 20
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10): tab.batch(idx, out)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps // 10): gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    ref = tab.table[idx]
    assert torch.equal(ref, out)
    e0.record()
    for _ in range(reps): torch.index_select(tab.table, 0, idx, out=out)
    e1.record(); torch.cuda.synchronize()
    us_t = e0.elapsed_time(e1) * 1000 / reps
    print(f"B={B:7d}: {us:8.1f} us  {2 * B * D * 4 / us / 1e3:7.1f} GB/s (read + write)   torch.index_select {us_t:8.1f} us", flush=True)
