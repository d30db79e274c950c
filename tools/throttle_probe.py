"""Is a sustained back-to-back GEMM stream clock/power limited?  Same graph (12 launches) replayed (a) back to back
for ~0.5 s, (b) one replay at a time with idle gaps; per-launch time from HIP events around each replay."""
import sys, os, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudo_speaker_vae_amd import ops
dev = "cuda:0"
name, M, N, K, cfg = sys.argv[1].split(","); M, N, K, cfg = int(M), int(N), int(K), int(cfg)
kind = {"NT": 0, "NN": 1, "TN": 2}[name]
sa = {0: (M, K), 1: (M, K), 2: (K, M)}[kind]; sb = {0: (N, K), 1: (K, N), 2: (K, N)}[kind]
A, B, C = torch.randn(sa, device=dev), torch.randn(sb, device=dev), torch.empty(M, N, device=dev)
ops.gemm(kind, A, B, tile=cfg, out=C); torch.cuda.synchronize()
NL = 12
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(NL): ops.gemm(kind, A, B, tile=cfg, out=C)
g.replay(); torch.cuda.synchronize()
def timed(nrep):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nrep): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / (nrep * NL)
time.sleep(1.0)
print("after idle, 1 replay :", " ".join(f"{timed(1):5.1f}" for _ in range(1)))
time.sleep(1.0)
print("idle gaps (50 ms)    :", end=" ")
for _ in range(6):
    time.sleep(0.05); print(f"{timed(1):5.1f}", end=" ")
print()
print("back to back x2000   :", " ".join(f"{timed(2000):5.1f}" for _ in range(3)))
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        return " | ".join(l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l))
    except Exception as e:
        return f"rocm-smi failed: {e}"
print("idle smi:", smi())
import threading
res = []
t = threading.Thread(target=lambda: res.append(smi()))
t.start()
x = timed(20000)
t.join()
print(f"under load ({x:5.1f} us/launch) smi:", res[0])
