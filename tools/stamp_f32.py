"""In-kernel phase stamps of the fp32 grouped GEMM launches of the cfg2 step (diagnostic library built with
`tools/build_variant.sh stamp -DPSVAE_STAMP`; run with PSVAE_LIB=variants/libpsvae_stamp.so).  Wave 0 of every workgroup
prints: cycles from entry to descriptor decoded | k loop | LDS exchange barrier | row loop | tail, plus its entry and exit on
the 100 MHz real-time clock.  This script runs three eager steps, groups the last step's lines by launch and prints, per
launch: workgroups, spread of the entry times, median phase lengths (shader cycles) and first-entry-to-last-exit in us."""
import os, re, subprocess, sys, statistics as st
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from pseudo_speaker_vae_amd.engine import TrainEngine
    B, D, Z = 1024, 256, 64
    eng = TrainEngine(D, Z, B, device="cuda:0", seed=0)
    x = torch.randn(B, D, device="cuda:0")
    eng.tune(x)
    for i in range(3):
        torch.cuda.synchronize(); print(f"STEP {i}", flush=True)
        eng.forward_backward(x)
        torch.cuda.synchronize()
    sys.exit(0)
out = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True).stdout
steps = out.split("STEP ")
last = steps[-1]
pat = re.compile(r"S kind=(\d+) M=(\d+) N=(\d+) K=(\d+) cfg=(\d+) epi=(\d+) s2=(\d+) wg=(\d+) rt0=(\d+) rt1=(\d+) d=(\d+),(\d+),(\d+),(\d+),(\d+)")
rows = [tuple(map(int, m.groups())) for m in pat.finditer(last)]
# a launch = a maximal run of rows whose entry times lie within 40 us of each other
rows.sort(key=lambda r: r[8])
launches, cur = [], []
for r in rows:
    if cur and r[8] - cur[0][8] > 4000:   # 100 MHz ticks: 40 us
        launches.append(cur); cur = []
    cur.append(r)
if cur: launches.append(cur)
print(f"{len(rows)} workgroup records in the last step, {len(launches)} launches")
for L in launches:
    t0 = min(r[8] for r in L); t1 = max(r[9] for r in L)
    kinds = sorted({(r[0], r[1], r[2], r[3], r[4], r[6]) for r in L})
    med = [st.median(r[10 + i] for r in L) for i in range(5)]
    mx = [max(r[10 + i] for r in L) for i in range(5)]
    entry_spread = (max(r[8] for r in L) - t0) / 100.0
    exit_spread = (t1 - min(r[9] for r in L)) / 100.0
    print(f"launch {len(L):4d} wgs  span {(t1 - t0) / 100.0:6.2f} us  entry spread {entry_spread:5.2f} us  exit spread {exit_spread:5.2f} us")
    print(f"    median cycles: decode {med[0]:6.0f} | k loop {med[1]:7.0f} | lds barrier {med[2]:6.0f} | row loop {med[3]:6.0f} | tail {med[4]:6.0f}   (max {mx[0]:.0f} | {mx[1]:.0f} | {mx[2]:.0f} | {mx[3]:.0f} | {mx[4]:.0f})")
    for k in kinds:
        n = sum(1 for r in L if (r[0], r[1], r[2], r[3], r[4], r[6]) == k)
        sub = [r for r in L if (r[0], r[1], r[2], r[3], r[4], r[6]) == k]
        print(f"      {n:4d} x kind {k[0]} M={k[1]} N={k[2]} K={k[3]} cfg {k[4]} stage2 {k[5]}: k loop median {st.median(r[11] for r in sub):.0f}, wg life median {st.median((r[9]-r[8]) for r in sub)/100.0:.2f} us")
