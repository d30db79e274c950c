"""Turn a tools/profile_step.sh run into the markdown summary + pmc.json committed under profiles/.
usage: summarize_profile.py OUT_DIR RAW_DIR      (RAW_DIR holds trace/, pmc_sq/, pmc_fetch/, pmc_write/)"""
import csv, glob, json, re, sys
from collections import defaultdict
out, raw = sys.argv[1], sys.argv[2]


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void psvae::", "").replace("psvae::", "")
    return n[:60]


print(f"# rocprofv3 summary ({out})\n")
b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
print("bench.py (un-profiled):", json.dumps({k: b[k] for k in ("value", "unit", "ms_per_step", "roofline", "cpu_baseline") if k in b}), "\n")
print("launch plan of every run below:", "`" + b["config"]["launch_plan"] + "`", "\n")
nl = b["config"].get("launches_per_step") or 10
st = glob.glob(raw + "/trace/**/*_kernel_stats.csv", recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    steps = json.loads(open(out + "/bench_under_trace.json").read().strip().splitlines()[-1])
    print("## `rocprofv3 --kernel-trace --stats -- python3 bench.py` (launch form: " + str(steps.get("config", {}).get("launch")) + "; the run also holds the calibration's launches of the other form: same kernels)\n")
    print("| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|")
    tot = 0
    for r in rows:
        if "psvae" not in r["Name"]:
            continue
        tot += int(r["TotalDurationNs"])
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {int(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.1f} |")
    g = [r for r in rows if "gemm_grouped" in r["Name"]]
    gt = sum(int(r["TotalDurationNs"]) for r in g)
    gc = sum(int(r["Calls"]) for r in g)
    nst = max(int(r["Calls"]) for r in rows if "bottleneck_fwd" in r["Name"])  # one forward bottleneck launch per step
    print(f"\nGEMM kernel (all pair instantiations): {gc} launches, average {gt/gc/1e3:.2f} us per launch, "
          f"{gt/1e3/nst:.1f} us per step ({gc/nst:.0f} launches per step); all psvae kernels {tot/1e3/nst:.1f} us per step "
          f"(bench under the tracer: {steps['ms_per_step']*1e3:.1f} us/step wall).\n")
pmc = {}
for tag, title in (("pmc_sq", "SQ / TCC counters (summed over the launches of one step)"),
                   ("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = glob.glob(raw + f"/{tag}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    rows = list(csv.DictReader(open(f[0])))
    by = defaultdict(lambda: defaultdict(float))
    names = {}
    for r in rows:
        did = int(r["Dispatch_Id"])
        by[did][r["Counter_Name"]] += float(r["Counter_Value"])
        names[did] = r["Kernel_Name"]
    ids = [i for i in sorted(by) if "psvae" in names[i]]
    # one step = the launches from an encoder-layer-0 GEMM (two launches before a forward bottleneck kernel) to the
    # last launch before the next one; take the last COMPLETE step
    bn = [k for k, i in enumerate(ids) if "bottleneck_fwd" in names[i]]
    if len(bn) < 2:
        continue
    lo, hi = bn[-2] - 2, bn[-1] - 2
    step_ids = ids[lo:hi]
    print(f"## {title}\n\n({len(step_ids)} launches: {', '.join(short(names[i]).split('<')[0] for i in step_ids)})\n")
    agg = defaultdict(float)
    for i in step_ids:
        for k, v in by[i].items():
            agg[k] += v
    for k, v in sorted(agg.items()):
        extra = ""
        if k == "FETCH_SIZE":
            extra = f"  (KB; x2 gfx950 wide-read correction = {2*v/1024:.1f} MB per step)"
            pmc["fetch_size_kb"] = v
        if k == "WRITE_SIZE":
            extra = f"  (KB = {v/1024:.1f} MB per step)"
            pmc["write_size_kb"] = v
        print(f"- {k}: {v:,.0f}{extra}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in agg and "SQ_WAVE_CYCLES" in agg:
        print(f"- MFMA busy cycles per SIMD: {agg['SQ_VALU_MFMA_BUSY_CYCLES']/1024:,.0f} (= {agg['SQ_VALU_MFMA_BUSY_CYCLES']/1024/2.4e3:.1f} us at 2.4 GHz)")
        if agg.get("TCC_HIT_sum"):
            print(f"- L2 hit rate: {agg['TCC_HIT_sum']/(agg['TCC_HIT_sum']+agg['TCC_MISS_sum']):.2f}")
    print()
if "fetch_size_kb" in pmc and "write_size_kb" in pmc:
    pmc["hbm_bytes_per_step"] = int((2 * pmc["fetch_size_kb"] + pmc["write_size_kb"]) * 1024)
    pmc["launch_plan"] = b["config"]["launch_plan"]
    pmc["source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `python bench.py --mode eager`, all launches of one step"
    pmc["note"] = ("FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); Infinity-Cache "
                   "hits are counted, so this is L2-fill traffic: an upper bound on HBM bytes")
    json.dump(pmc, open(out + "/pmc.json", "w"), indent=1)
