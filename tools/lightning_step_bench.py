import sys, time, torch
sys.path.insert(0, "/root/repo")
from pseudo_speaker_vae_amd.lightning import PseudoSpeakerVAE
m = PseudoSpeakerVAE(model={"input_dim": 256, "latent_dim": 64}, lr=1e-3).to("cuda:0")
x = torch.randn(1024, 256, device="cuda:0"); y = torch.zeros(1024, dtype=torch.long, device="cuda:0")
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
def step():
    opt.zero_grad(set_to_none=True)
    out = m.training_step((x, y), 0)
    out["loss"].backward()
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print("lightning-style training_step + backward: %.1f us/step" % ((time.perf_counter() - t0) / 200 * 1e6))
