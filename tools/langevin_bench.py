"""Fused Langevin sampler vs the general torch-op loop on the same GPU (rows x steps per second)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudo_speaker_vae_amd import sampling
from ps_vae.lightning import PseudoSpeakerVAE
dev = "cuda:0"
torch.manual_seed(0)
mod = PseudoSpeakerVAE(model=dict(input_dim=256, latent_dim=64),
                       classifier=dict(input_dim=64, num_classes=3, num_layers=2, hidden_dim=128, activation="relu")).to(dev)
fus = sampling._fusable_classifier(mod.classifier, 1)
for N, T in ((16, 5000), (4096, 500), (125000, 100), (1000000, 100)):
    z = torch.randn(N, 64, device=dev)
    sampling.langevin_fused(fus, z, 1, 0.01, 1.0, 2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sampling.langevin_fused(fus, z, 1, 0.01, 1.0, T)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    flop = N * T * 2 * 2 * (128 * 64 + 3 * 128)
    print(f"fused   N={N:8d} T={T:5d}: {dt*1e3:9.2f} ms  {N*T/dt/1e6:9.1f} M row-steps/s  {flop/dt/1e12:6.2f} TFLOP/s", flush=True)
for N, T in ((16, 500), (4096, 100)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    # general path = the reference's loop with torch ops (force by passing a dict-free but non-fusable target type)
    z = torch.randn(N, 64, device=dev).requires_grad_()
    for _ in range(T):
        lp = torch.log_softmax(mod.classifier(z), -1)[:, 1] - 0.5 * (z ** 2).sum(1)
        g = torch.autograd.grad(lp.sum(), z)[0]
        z = (z + 0.5 * 1e-4 * g + 0.01 * torch.randn_like(z)).detach().requires_grad_()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"torch   N={N:8d} T={T:5d}: {dt*1e3:9.2f} ms  {N*T/dt/1e6:9.1f} M row-steps/s", flush=True)
