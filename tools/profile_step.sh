#!/bin/bash
# Reproduce the committed profiles: the bench (autotuned) first, then kernel-trace stats and PMC passes of the
# same command with the same launch plan (--plan: no tuning launches inside the profiled processes).
# Run on the GPU box:  gpurun -- 'bash tools/profile_step.sh r01'
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"
PLAN=$(python3 -c "import json,sys; print(json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])['config']['launch_plan'])")
echo "$PLAN" > "$OUT/plan.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline --plan "$PLAN" > "$OUT/bench_under_trace.json" 2> "$OUT/trace.log"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py --mode eager --steps 20 --warmup 5 --no-cpu-baseline --plan "$PLAN" > /dev/null 2> "$OUT/pmc_sq.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --mode eager --steps 20 --warmup 5 --no-cpu-baseline --plan "$PLAN" > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --mode eager --steps 20 --warmup 5 --no-cpu-baseline --plan "$PLAN" > /dev/null 2> "$OUT/pmc_write.log"
python3 tools/summarize_profile.py "$OUT" > "$OUT/SUMMARY.md"
python3 tools/parse_trace.py "$OUT/trace" 100 > "$OUT/timeline.txt"
python3 tools/launch_roofline.py "$OUT/trace" > "$OUT/launch_roofline.md"
cat "$OUT/SUMMARY.md" "$OUT/timeline.txt" "$OUT/launch_roofline.md"
