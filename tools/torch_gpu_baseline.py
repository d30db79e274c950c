"""Context number: the same training pass (3-Linear encoders x2, reparameterise, 3-Linear decoder, MSE/10 + KL, backward)
written with stock PyTorch modules on the same GPU -- rocBLAS / hipBLASLt GEMMs + elementwise kernels -- eager and
captured in a HIP graph.  This is what the reference's model code does on an MI355X; it is NOT the product path."""
import sys, time, torch, torch.nn as nn
B, D, Z, H = 1024, 256, 64, 512
if len(sys.argv) > 3: B, D, Z = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = "cuda:0"
torch.backends.cuda.matmul.allow_tf32 = False
torch.manual_seed(0)
mlp = lambda i, o: nn.Sequential(nn.Linear(i, H), nn.ReLU(), nn.Linear(H, H), nn.ReLU(), nn.Linear(H, o))
enc_mu, enc_sig, dec = mlp(D, Z).to(dev), mlp(D, Z).to(dev), mlp(Z, D).to(dev)
params = [p for m in (enc_mu, enc_sig, dec) for p in m.parameters()]
x = torch.randn(B, D, device=dev)

def step():
    for p in params: p.grad = None
    mu, lv = enc_mu(x), enc_sig(x)
    z = mu + torch.exp(0.5 * lv) * torch.randn_like(lv)
    xh = dec(z)
    recon = nn.functional.mse_loss(xh, x) / 10
    kl = -0.5 * torch.mean(torch.sum(1 + lv - mu.pow(2) - lv.exp(), dim=-1))
    (recon + kl).backward()

for _ in range(5): step()
torch.cuda.synchronize()
n = 200
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(n): step()
e1.record(); torch.cuda.synchronize()
print(f"stock PyTorch eager        : {1e6 * (time.perf_counter() - t0) / n:7.1f} us/step wall, {1e3 * e0.elapsed_time(e1) / n:7.1f} us/step device")
# graph capture of the same step (static grads)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
for p in params: p.grad = None
with torch.cuda.graph(g):
    step()
for _ in range(5): g.replay()
torch.cuda.synchronize()
e0.record()
for _ in range(n): g.replay()
e1.record(); torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / n
print(f"stock PyTorch, HIP graph   : {us:7.1f} us/step  = {B / us:.2f} M embeddings/s  (B={B}, D={D}, Z={Z})")

# decode-only (cfg5): 1,000,000 latent rows through the decoder, no grad, 65,536-row chunks like the product path
if "--decode" in sys.argv:
    N = 1_000_000
    zs = torch.randn(N, Z, device=dev)
    out = torch.empty(N, D, device=dev)
    def decode_all():
        with torch.no_grad():
            for r0 in range(0, N, 65536):
                out[r0:r0 + 65536] = dec(zs[r0:r0 + 65536])
    decode_all(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): decode_all()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"stock PyTorch decode 1M rows: {ms:7.2f} ms = {N / ms / 1e3:.1f} M embeddings/s")
    # same decode with bf16 weights / activations (stock PyTorch, for psvae_decode_bf16's context)
    decb = dec.to(torch.bfloat16)
    zsb = zs.to(torch.bfloat16)
    outb = torch.empty(N, D, device=dev)
    def decode_all_bf16():
        with torch.no_grad():
            for r0 in range(0, N, 131072):
                outb[r0:r0 + 131072] = decb(zsb[r0:r0 + 131072]).float()
    decode_all_bf16(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): decode_all_bf16()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"stock PyTorch bf16 decode 1M rows (z already bf16): {ms:7.2f} ms = {N / ms / 1e3:.1f} M embeddings/s")
