"""Parse a rocprofv3 kernel trace of tools/tile_sweep.py into a launches x tile-config table (us)."""
import csv, glob, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm_grouped" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
NL = 12  # gemm launches per step
steps = [rows[i:i + NL] for i in range(0, len(rows), NL)]
cfgs = ["c1", "c2", "c3", "c4", "c5", "c6", "c7", "c8", "c9", "auto"]
per = len(steps) // len(cfgs)
tab = {}
for ci, c in enumerate(cfgs):
    ss = steps[ci * per + 4:(ci + 1) * per]  # skip warm-up steps
    for li in range(NL):
        d = sorted(int(s[li]["End_Timestamp"]) - int(s[li]["Start_Timestamp"]) for s in ss)
        name = re.search(r"<([\d, ]+)>", ss[0][li]["Kernel_Name"])
        tab[(li, c)] = (d[len(d) // 2] / 1000, name.group(1).replace(", ", ""), int(ss[0][li]["Grid_Size_X"]) // 256)
names = ["F1 enc1", "F2 enc2", "F3 enc3", "F5 dec1", "F6 dec2", "F7 dec3", "B1 dec3", "B2 dec2", "B3 dec1", "B4 enc3", "B5 enc2", "B6 enc1"]
print("launch     " + "".join(f"{c:>13}" for c in cfgs))
tot = defaultdict(float)
best = 0
for li in range(NL):
    line = f"{names[li]:10} "
    m = min(tab[(li, c)][0] for c in cfgs[:-1])
    best += m
    for c in cfgs:
        t, nm, g = tab[(li, c)]
        tot[c] += t
        line += f"{t:6.1f}{'*' if t == m else ' '}({g:4d})"
    print(line + ("  auto=" + tab[(li, 'auto')][1]))
print("total      " + "".join(f"{tot[c]:13.1f}" for c in cfgs), " best-per-launch", round(best, 1))
