"""Turn a tools/profile_step.sh output directory into the markdown summary committed under profiles/."""
import csv, glob, json, re, sys
from collections import defaultdict
out = sys.argv[1]


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void psvae::", "").replace("psvae::", "")
    return n[:46]


print(f"# rocprofv3 summary ({out})\n")
try:
    b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
    print("bench.py (un-profiled):", json.dumps({k: b[k] for k in ("value", "unit", "ms_per_step", "roofline", "cpu_baseline") if k in b}), "\n")
except Exception as e:
    print("bench.json missing:", e)
st = glob.glob(out + "/trace/**/*_kernel_stats.csv", recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    tr = list(csv.DictReader(open(glob.glob(out + "/trace/**/*_kernel_trace.csv", recursive=True)[0])))
    steps = json.loads(open(out + "/bench_under_trace.json").read().strip().splitlines()[-1])
    nsteps = steps["steps"] + steps["warmup"] + 3  # + capture warm-ups and the capture itself
    print("## `rocprofv3 --kernel-trace --stats -- python3 bench.py` (graph mode, cfg2)\n")
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
    nst = max(int(r["Calls"]) for r in rows if "transpose_f32" in r["Name"])  # one weight transpose per step
    print(f"\nGEMM kernel (all pair instantiations): {gc} launches, average {gt/gc/1e3:.2f} us per launch, "
          f"{gt/1e3/nst:.1f} us per step ({gc/nst:.0f} launches per step); all psvae kernels {tot/1e3/nst:.1f} us per step "
          f"(bench under the tracer: {steps['ms_per_step']*1e3:.1f} us/step wall).\n")
for tag, title in (("pmc_sq", "SQ / TCC counters (summed over the MFMA-carrying launches of the last step)"),
                   ("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = glob.glob(out + f"/{tag}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    rows = list(csv.DictReader(open(f[0])))
    by = defaultdict(lambda: defaultdict(float))
    names = {}
    for r in rows:
        did = int(r["Dispatch_Id"])
        by[did][r["Counter_Name"]] += float(r["Counter_Value"])
        names[did] = r["Kernel_Name"]
    ids = sorted(by)
    last_t = max(i for i in ids if "transpose_f32" in names[i])  # the last step starts at its weight transpose
    gemm = [i for i in ids if i > last_t and ("gemm_grouped" in names[i] or "reparam_linear" in names[i])]
    if len([i for i in ids if i > last_t]) < 10:  # the run ended inside that step: take the one before
        prev_t = max(i for i in ids if "transpose_f32" in names[i] and i < last_t)
        gemm = [i for i in ids if prev_t < i < last_t and ("gemm_grouped" in names[i] or "reparam_linear" in names[i])]
    print(f"({len(gemm)} launches)\n")
    agg = defaultdict(float)
    for i in gemm:
        for k, v in by[i].items():
            agg[k] += v
    print(f"## {title}\n")
    for k, v in sorted(agg.items()):
        extra = ""
        if k == "FETCH_SIZE":
            extra = f"  (KB; x2 gfx950 wide-read correction = {2*v/1024:.1f} MB per step)"
        if k == "WRITE_SIZE":
            extra = f"  (KB = {v/1024:.1f} MB per step)"
        print(f"- {k}: {v:,.0f}{extra}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in agg and "SQ_WAVE_CYCLES" in agg:
        print(f"- MFMA busy cycles per SIMD: {agg['SQ_VALU_MFMA_BUSY_CYCLES']/1024:,.0f} (= {agg['SQ_VALU_MFMA_BUSY_CYCLES']/1024/2.4e3:.1f} us at 2.4 GHz; the fp32 roofline time of the step)")
        if agg.get("TCC_HIT_sum"):
            print(f"- L2 hit rate: {agg['TCC_HIT_sum']/(agg['TCC_HIT_sum']+agg['TCC_MISS_sum']):.2f}")
    print()
