"""Summarise rocprofv3 --pmc counter_collection.csv per kernel name (median over dispatches of one step position)."""
import csv, glob, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = defaultdict(lambda: defaultdict(list))
order = []
for r in rows:
    did = int(r["Dispatch_Id"])
    by[did][r["Counter_Name"]].append(float(r["Counter_Value"]))
    by[did]["_name"] = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void psvae::", "").replace("psvae::", "")
    by[did]["_grid"] = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
ids = sorted(by)
psv = [i for i in ids if "gemm" in by[i]["_name"]]
per = 12
last = psv[-per:]
cnames = sorted({k for i in last for k in by[i] if not k.startswith("_")})
print(f"{'kernel':28} {'wgs':>5} " + " ".join(f"{c[-14:]:>14}" for c in cnames))
for i in last:
    print(f"{by[i]['_name'][:28]:28} {by[i]['_grid']:5d} " + " ".join(f"{sum(by[i][c]):14.0f}" for c in cnames))
