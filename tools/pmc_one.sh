#!/bin/bash
# PMC counters of isolated GEMM launches (tools/cold_bench.py specs).  usage: pmc_one.sh TAG spec...
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc1_$TAG
mkdir -p "$OUT"
for spec in "$@"; do
  n=$(echo $spec | tr ',' '_')
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/$n" -- python3 tools/cold_bench.py $spec > /dev/null 2> "$OUT/$n.log"
  python3 - "$OUT/$n" "$spec" <<'PY'
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
by = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(f)):
    if "gemm" in r["Kernel_Name"]:
        by[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
last = by[sorted(by)[-1]]
print(sys.argv[2], " ".join(f"{k}={v:.0f}" for k, v in sorted(last.items())))
PY
done
