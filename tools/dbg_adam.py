import sys, torch
sys.path.insert(0, ".")
from oracle import psvae_oracle as O
from pseudo_speaker_vae_amd.engine import TrainEngine
eng = TrainEngine(256, 64, 64, device="cuda:0", seed=1, lr=1e-3, weight_decay=0.01)
g = torch.Generator().manual_seed(2)
cur = {k: v.cpu().clone() for k, v in eng.state_dict().items()}
grads = [torch.randn(t.shape, generator=g) * 0.1 for t in eng.params]
for dst, src in zip(eng.grads, grads):
    dst.copy_(src)
eng.optimizer_step(lr=1e-3)
torch.cuda.synchronize()
for k, gr, q in zip(O.PARAM_KEYS, grads, eng.params):
    want, m, v = O.adam_step(cur[k], gr, torch.zeros_like(gr), torch.zeros_like(gr), 1, 1e-3, weight_decay=0.01)
    d = (q.cpu() - want).abs()
    i = d.argmax()
    print(k, float(d.max()), float(want.abs().max()), "at", int(i), float(gr.flatten()[i]), float(cur[k].flatten()[i]), float(want.flatten()[i]), float(q.cpu().flatten()[i]))
