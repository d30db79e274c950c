#!/bin/bash
# rocprofv3 --kernel-trace --stats for cfg3 and cfg5 -> gpurun_out/stats_cfg3 / stats_cfg5 and markdown tables
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/stats_cfg3 gpurun_out/stats_cfg5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_cfg3 -- python3 bench.py --no-cpu-baseline --dtype bf16 --cond --batch 4096 > /dev/null 2> gpurun_out/stats_cfg3.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_cfg5 -- python3 bench.py --no-cpu-baseline --workload decode > /dev/null 2> gpurun_out/stats_cfg5.log
python3 - <<'PY'
import csv, glob, re
for cfg in ("cfg3", "cfg5"):
    f = glob.glob(f"gpurun_out/stats_{cfg}/**/*_kernel_stats.csv", recursive=True)[0]
    print(f"\n## {cfg}: `rocprofv3 --kernel-trace --stats` per-kernel averages\n\n| kernel | calls | avg us | % |\n|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        if "psvae" in r["Name"]:
            n = re.sub(r"\(.*", "", r["Name"]).replace("void psvae::", "").replace("psvae::", "")
            print(f"| `{n}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.1f} |")
PY
