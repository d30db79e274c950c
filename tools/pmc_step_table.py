"""Per-launch table of rocprofv3 --pmc counters over the steps of a bench run: the step's period is detected from the
sequence of kernel names, counters are averaged per position.  usage: pmc_step_table.py DIR [DIR ...]  (one DIR per pass)"""
import csv, glob, re, sys
from collections import defaultdict

def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    by, names, grid = defaultdict(lambda: defaultdict(float)), {}, {}
    for r in csv.DictReader(open(f)):
        i = int(r["Dispatch_Id"])
        by[i][r["Counter_Name"]] += float(r["Counter_Value"])
        names[i] = r["Kernel_Name"]
        grid[i] = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
    ids = [i for i in sorted(by) if "psvae" in names[i]]
    return by, names, grid, ids

def short(n):
    return re.sub(r"\(.*", "", n).replace("void psvae::", "").replace("psvae::", "")[:52]

tabs, order, wgs = defaultdict(dict), None, None
for d in sys.argv[1:]:
    by, names, grid, ids = load(d)
    seq = [names[i] + str(grid[i]) for i in ids]
    per = next(p for p in range(4, 40) if seq[-p:] == seq[-2 * p:-p] == seq[-3 * p:-2 * p])
    nper = min(8, len(seq) // per - 1)
    for pos in range(per):
        acc = defaultdict(float)
        for k in range(nper):
            for c, v in by[ids[-(k + 1) * per + pos]].items():
                acc[c] += v / nper
        tabs[pos].update(acc)
    order = [short(names[i]) for i in ids[-per:]]
    wgs = [grid[i] for i in ids[-per:]]
cols = sorted({c for t in tabs.values() for c in t})
print("| # | launch | workgroups | " + " | ".join(cols) + " |")
print("|---|---|---|" + "---|" * len(cols))
tot = defaultdict(float)
for pos, nm in enumerate(order):
    print(f"| {pos} | `{nm}` | {wgs[pos]} | " + " | ".join(f"{tabs[pos].get(c, 0):.0f}" for c in cols) + " |")
    for c in cols:
        tot[c] += tabs[pos].get(c, 0)
print("| | **step** | | " + " | ".join(f"**{tot[c]:.0f}**" for c in cols) + " |")
