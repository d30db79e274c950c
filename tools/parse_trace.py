"""Per-launch timeline of the step from a rocprofv3 kernel trace: period auto-detected, durations averaged (and min)
over the last periods.  usage: parse_trace.py DIR [nperiods]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = [r for r in csv.DictReader(open(f)) if "psvae" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] + r["Grid_Size_X"] for r in rows]
per = next((p for p in range(5, 60) if names[-p:] == names[-2 * p:-p] == names[-3 * p:-2 * p]), 18)
NP = min(NP, len(rows) // per - 1)
tot = 0
for i in range(per):
    ds = [int(rows[-(k + 1) * per + i]["End_Timestamp"]) - int(rows[-(k + 1) * per + i]["Start_Timestamp"]) for k in range(NP)]
    r = rows[-per + i]
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void psvae::", "").replace("psvae::", "")
    wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    avg = sum(ds) / len(ds) / 1000
    tot += avg
    print(f"{avg:7.1f}us (min {min(ds)/1000:5.1f}) grid {wgs:5d} vgpr {r['VGPR_Count']:>3} lds {r['LDS_Block_Size']:>6} {nm[:56]}")
print("launches", per, "periods", NP, "sum", round(tot, 1))
