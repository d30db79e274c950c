"""Host-bound cost of the Lightning-style step (training_step + loss.backward()), without and with a classifier head
(--classifier) and a frozen EmbeddingClassifier consistency head on x_hat (--consistency);
PSVAE_AUTOGRAD_FUSED=0 / --torch-classifier / --torch-consistency select the unfused paths for comparison."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseudo_speaker_vae_amd.lightning import PseudoSpeakerVAE
with_clf = "--classifier" in sys.argv
kw = dict(classifier=dict(input_dim=64, num_classes=3, num_layers=2, hidden_dim=128, activation="relu")) if with_clf else {}
m = PseudoSpeakerVAE(model={"input_dim": 256, "latent_dim": 64}, lr=1e-3, **kw).to("cuda:0")
if "--consistency" in sys.argv:
    from pseudo_speaker_vae_amd.embedding_classifier import EmbeddingClassifier
    head = EmbeddingClassifier(256, 3, 128).to("cuda:0").eval()
    for q in head.parameters():
        q.requires_grad = False
    m.consistency_classifier = head
    if "--torch-consistency" in sys.argv:
        head.fused_ok = lambda: False
if "--torch-classifier" in sys.argv:
    m.classifier.fused_params = lambda *a, **k: None
x = torch.randn(1024, 256, device="cuda:0"); y = torch.randint(0, 3, (1024,), device="cuda:0")
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
def step():
    opt.zero_grad(set_to_none=True)
    out = m.training_step((x, y), 0)
    out["loss"].backward()
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("lightning-style training_step + backward%s: %.1f us/step" % ((" (classifier head)" if with_clf else "") + (" (consistency head)" if "--consistency" in sys.argv else ""), (time.perf_counter() - t0) / 200 * 1e6))
