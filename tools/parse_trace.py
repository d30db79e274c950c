import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "psvae" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = 18
seg = rows[-per:]
t0 = int(seg[0]["Start_Timestamp"]); tot = 0
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot += d
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void psvae::", "").replace("psvae::", "")
    print(f"{(int(r['Start_Timestamp'])-t0)/1000:8.1f}us dur {d/1000:6.1f}us grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d} lds {r['LDS_Block_Size']:>6} vgpr {r['VGPR_Count']:>3} {nm}")
print("sum", tot / 1000, "span", (int(seg[-1]["End_Timestamp"]) - t0) / 1000)
