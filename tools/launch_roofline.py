"""Per-launch roofline of the fused step from a kernel trace (tools/profile_step.sh output): algorithmic FLOP of each of
the 10 launches (cfg2 shapes by default) over its average duration.  usage: launch_roofline.py TRACE_DIR [B D Z]"""
import csv, glob, re, sys
d = sys.argv[1]
B, D, Z = (int(a) for a in sys.argv[2:5]) if len(sys.argv) >= 5 else (1024, 256, 64)
H = 512
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "psvae" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] + r["Grid_Size_X"] for r in rows]
per = next(p for p in range(8, 60) if names[-p:] == names[-2 * p:-p] == names[-3 * p:-2 * p])
NP = min(100, len(rows) // per - 1)
mm = lambda m, n, k: 2.0 * m * n * k
# launches of psvae_train_step_f32 in order; a trailing slab_reduce (split weight gradients) has no FLOP
sites = [("F1  encoders layer 0 (x2)", 2 * mm(B, H, D)),
         ("F2  encoders layer 1 (x2) + heads as second stage (partials)", 2 * mm(B, H, H) + 2 * mm(B, Z, H)),
         ("bottleneck fwd: sum partials, reparam, KL, decoder layer 0", mm(B, H, Z)),
         ("F6  decoder layer 1", mm(B, H, H)),
         ("F7  decoder layer 2 + MSE gradient", mm(B, D, H)),
         ("B1  dW decoder.4 + dgrad", mm(D, H, B) + mm(B, H, D)),
         ("B2  dW decoder.2 + dgrad + decoder.0 dgrad as second stage", mm(H, H, B) + mm(B, H, H) + mm(B, Z, H)),
         ("bottleneck bwd: g_mu, g_logvar, heads' dgrad (x2), loss", 2 * mm(B, H, Z)),
         ("B5  dW encoders layer 1 (x2) + dgrad (x2)", 2 * mm(H, H, B) + 2 * mm(B, H, H)),
         ("B6  dW encoders layer 0 (x2) + dW heads (x2) + dW decoder.0", 2 * mm(H, D, B) + 2 * mm(Z, H, B) + mm(H, Z, B))]
if per != len(sites):
    print(f"(trace has {per} launches per step, the table below assumes the 10-launch fused sequence)")
print(f"| launch | GFLOP | avg us | TFLOP/s | of 157.3 |\n|---|---|---|---|---|")
tot_f = tot_t = 0.0
for i in range(per):
    ds = [int(rows[-(k + 1) * per + i]["End_Timestamp"]) - int(rows[-(k + 1) * per + i]["Start_Timestamp"]) for k in range(NP)]
    us = sum(ds) / len(ds) / 1e3
    name, fl = sites[i] if i < len(sites) else ("slab_reduce (split weight gradients)", 0.0)
    tot_f += fl; tot_t += us
    tf = fl / us * 1e-6 if us else 0.0
    print(f"| {name} | {fl / 1e9:.3f} | {us:.1f} | {tf:.1f} | {tf / 157.3:.0%} |" if fl else f"| {name} | – | {us:.1f} | – | – |")
print(f"| **step (sum of launches)** | {tot_f / 1e9:.3f} | {tot_t:.1f} | {tot_f / tot_t * 1e-6:.1f} | {tot_f / tot_t * 1e-6 / 157.3:.0%} |")
