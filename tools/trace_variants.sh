#!/bin/bash
# Per-launch timeline of the bench step for several library builds in ONE call (same box).
# usage: trace_variants.sh "bench args" lib1.so lib2.so ...
ARGS=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  n=$(basename $lib .so)
  OUT=gpurun_out/trace_$n
  rm -rf "$OUT"; mkdir -p "$OUT"
  PSVAE_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 $ARGS > "$OUT/bench.json" 2> "$OUT/log.txt"
  echo "== $n"
  python3 tools/parse_trace.py "$OUT" 100
  rm -rf "$OUT"/*/  # traces are large; keep only the printed summary
done
