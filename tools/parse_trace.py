"""Print the last step's kernel timeline from a rocprofv3 kernel trace (period auto-detected)."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "psvae" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
per = next((p for p in range(8, 60) if names[-p:] == names[-2 * p:-p]), 18)
seg = rows[-per:]
tot = 0
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot += d
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void psvae::", "").replace("psvae::", "")
    wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    print(f"{d/1000:7.1f}us grid {wgs:5d} vgpr {r['VGPR_Count']:>3} {nm[:64]}")
print("launches", per, "sum", tot / 1000)
