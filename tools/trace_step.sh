#!/bin/bash
# kernel-trace the bench step (graph mode, default steps) and print the per-launch timeline.  usage: trace_step.sh TAG [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 "$@" > "$OUT/bench.json" 2> "$OUT/log.txt"
python3 tools/parse_trace.py "$OUT" 100
