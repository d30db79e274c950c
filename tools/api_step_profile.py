"""Host-time breakdown of the drop-in API step (PseudoSpeakerVAE.training_step + loss.backward(), cfg2 shape).

Each segment is timed on the HOST with perf_counter and NO device sync inside the loop (the question is what the host
spends per step; the GPU runs behind), over 300 steps after warm-up; the whole-step wall time (with a sync at the end)
is printed beside the sum.  PSVAE_API_GRAPH=0 shows the eager path for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudo_speaker_vae_amd.lightning import PseudoSpeakerVAE

B, D, Z = 1024, 256, 64
dev = "cuda:0"
torch.manual_seed(0)
m = PseudoSpeakerVAE(model={"input_dim": D, "latent_dim": Z}, optimizer=dict(lr=1e-3), scheduler=dict(T_max=10)).to(dev)
m.model.tune_from_committed_plan(B)
opt = m.configure_optimizers()["optimizer"]
xs = [torch.randn(B, D, device=dev) for _ in range(8)]
y = torch.zeros(B, dtype=torch.long, device=dev)
with_adam = "--with-adam" in sys.argv
seg = {"zero_grad": 0.0, "training_step": 0.0, "backward": 0.0, "optimizer.step": 0.0}

def step(i, acc=None):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    t1 = time.perf_counter()
    out = m.training_step((xs[i % 8], y), i)
    t2 = time.perf_counter()
    out["loss"].backward()
    t3 = time.perf_counter()
    if with_adam:
        opt.step()
    t4 = time.perf_counter()
    if acc is not None:
        acc["zero_grad"] += t1 - t0; acc["training_step"] += t2 - t1; acc["backward"] += t3 - t2; acc["optimizer.step"] += t4 - t3

for i in range(300):
    step(i)
torch.cuda.synchronize()
N = 300
# (a) host time per segment with the GPU allowed to fall behind only a little: sync every 10 steps outside the timers
for i in range(N):
    step(i, seg)
    if i % 10 == 9:
        torch.cuda.synchronize()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    step(i)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e6
served = m.model._scratch_cache.graphs.served
print(f"API step, B={B} D={D} Z={Z}, graph path {'ON' if served else 'OFF'} ({served} steps served){' +Adam' if with_adam else ''}")
for k, v in seg.items():
    print(f"  host {k:16s} {v / N * 1e6:7.1f} us")
print(f"  host sum              {sum(seg.values()) / N * 1e6:7.1f} us")
print(f"  wall per step         {wall:7.1f} us  ({B / wall:.2f} M emb/s)")
# finer: inside training_step / backward via the autograd profiler's CPU totals would perturb the numbers; instead time
# the pieces of the graph path in isolation
slot = next(iter(m.model._scratch_cache.graphs.slots.values()), None)
if slot is not None:
    def t(f, n=300):
        torch.cuda.synchronize(); a = time.perf_counter()
        for _ in range(n): f()
        b = time.perf_counter(); torch.cuda.synchronize(); return (b - a) / n * 1e6
    params = m.model.ordered_params()
    print(f"  piece: slot.x.copy_(x)        {t(lambda: slot.x.copy_(xs[0])):6.1f} us host")
    print(f"  piece: graph.replay()         {t(lambda: slot.graph.replay()):6.1f} us host (GPU-bound when > host)")
    print(f"  piece: loss3.clone()          {t(lambda: slot.loss3.clone()):6.1f} us host")
    print(f"  piece: flat.mul_(g)           {t(lambda: slot.flat.mul_(slot.loss3[0])):6.1f} us host")
    print(f"  piece: 18 fresh grad views    {t(lambda: slot._views()):6.1f} us host")
    print(f"  piece: usable() alias check   {t(lambda: slot.usable(params)):6.1f} us host")
    print(f"  piece: ordered_params()       {t(lambda: m.model.ordered_params()):6.1f} us host")
    print(f"  piece: _next_rng              {t(lambda: m.model._next_rng(B * Z)):6.1f} us host")
