"""Average of rocprofv3 --pmc counters per kernel name over the dispatches of a run.  usage: pmc_kernel.py DIR [substr]"""
import csv, glob, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else "psvae"
by = defaultdict(lambda: defaultdict(float)); names = {}
for r in csv.DictReader(open(f)):
    d = int(r["Dispatch_Id"]); by[d][r["Counter_Name"]] += float(r["Counter_Value"]); names[d] = r["Kernel_Name"]
agg = defaultdict(lambda: defaultdict(list))
for d, c in by.items():
    if sub in names[d]:
        nm = re.sub(r"\(.*", "", names[d]).replace("void psvae::", "")[:60]
        for k, v in c.items():
            agg[nm][k].append(v)
for nm, c in agg.items():
    print(nm, " ".join(f"{k}={sum(v)/len(v):.1f} (n={len(v)})" for k, v in c.items()))
