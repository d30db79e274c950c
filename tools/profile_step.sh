#!/bin/bash
# Reproduce the committed profiles of the headline command: the bench first (the committed launch plan), then the
# kernel-trace stats and the PMC passes of the SAME command with the SAME plan (--plan), each counter group in its own
# run (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; no --pmc beside a trace domain).
# Raw rocprofv3 output stays in /tmp on the box; only the summaries land in gpurun_out/prof_TAG/ (copy into profiles/).
# Run on the GPU box:  gpurun -- 'bash tools/profile_step.sh r02 [bench args]'
set -e
TAG=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
RAW=/tmp/psvae_prof_$TAG
rm -rf "$OUT" "$RAW"; mkdir -p "$OUT" "$RAW"
python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.log"
PLAN=$(python3 -c "import json,sys; print(json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])['config']['launch_plan'])")
echo "$PLAN" > "$OUT/plan.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/trace" -- python3 bench.py --no-cpu-baseline --plan "$PLAN" "$@" > "$OUT/bench_under_trace.json" 2> "$OUT/trace.log"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$RAW/pmc_sq" -- python3 bench.py --mode eager --steps 20 --warmup 5 --no-cpu-baseline --plan "$PLAN" "$@" > /dev/null 2> "$OUT/pmc_sq.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$RAW/pmc_fetch" -- python3 bench.py --mode eager --steps 20 --warmup 5 --no-cpu-baseline --plan "$PLAN" "$@" > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$RAW/pmc_write" -- python3 bench.py --mode eager --steps 20 --warmup 5 --no-cpu-baseline --plan "$PLAN" "$@" > /dev/null 2> "$OUT/pmc_write.log"
find "$RAW/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
python3 tools/summarize_profile.py "$OUT" "$RAW" > "$OUT/SUMMARY.md"
python3 tools/parse_trace.py "$RAW/trace" 100 > "$OUT/timeline.txt"
python3 tools/launch_roofline.py "$RAW/trace" > "$OUT/launch_roofline.md" || true
cat "$OUT/SUMMARY.md" "$OUT/timeline.txt" "$OUT/launch_roofline.md"
