"""cProfile of the Lightning-style step (host side): where do the ~450 us go?"""
import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseudo_speaker_vae_amd.lightning import PseudoSpeakerVAE
m = PseudoSpeakerVAE(model={"input_dim": 256, "latent_dim": 64}, lr=1e-3).to("cuda:0")
x = torch.randn(1024, 256, device="cuda:0"); y = torch.zeros(1024, dtype=torch.long, device="cuda:0")
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
def step():
    opt.zero_grad(set_to_none=True)
    out = m.training_step((x, y), 0)
    out["loss"].backward()
for _ in range(30): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
