"""Measure the per-launch floor of tiny dependent kernels on the box: eager vs HIP graph."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudo_speaker_vae_amd import ops, _ffi

dev = "cuda:0"
out = torch.empty(256, device=dev)
def k():
    _ffi.lib().psvae_philox_normal_f32(out.data_ptr(), 256, 1, 0, torch.cuda.current_stream().cuda_stream)
for _ in range(100): k()
torch.cuda.synchronize()
for n in (1000, 5000):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): k()
    e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"eager n={n}: wall {1e6*(t1-t0)/n:.2f} us/launch, device {1e3*e0.elapsed_time(e1)/n:.2f} us/launch")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    k()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(100): k()
for _ in range(3): g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"graph: device {1e3*e0.elapsed_time(e1)/5000:.2f} us/launch")
# torch's own tiny kernel for comparison
a = torch.zeros(256, device=dev)
for _ in range(100): a.add_(1.0)
torch.cuda.synchronize()
e0.record()
for _ in range(5000): a.add_(1.0)
e1.record(); torch.cuda.synchronize()
print(f"torch add_: device {1e3*e0.elapsed_time(e1)/5000:.2f} us/launch")
