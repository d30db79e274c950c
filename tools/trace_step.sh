#!/bin/bash
# kernel-trace the bench step (graph mode) and print the per-launch timeline.  usage: trace_step.sh TAG [bench args]
# Raw rocprofv3 output goes to /tmp on the GPU box (it can be tens of MB); only the parsed timeline, the bench line and
# the per-kernel stats are kept under gpurun_out/trace_TAG/.
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAW=/tmp/psvae_trace_$TAG
OUT=gpurun_out/trace_$TAG
rm -rf "$RAW" "$OUT"; mkdir -p "$RAW" "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW" -- python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 "$@" > "$OUT/bench.json" 2> "$OUT/log.txt"
python3 tools/parse_trace.py "$RAW" 100 | tee "$OUT/timeline.txt"
find "$RAW" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
