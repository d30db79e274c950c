#!/bin/bash
# Reproduce the committed profiles: kernel-trace stats of the bench command, then PMC passes.
# Run on the GPU box:  gpurun -- 'bash tools/profile_step.sh r01'
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline > "$OUT/bench_under_trace.json" 2> "$OUT/trace.log"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py --mode eager --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_sq.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --mode eager --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --mode eager --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_write.log"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"
python3 tools/summarize_profile.py "$OUT" > "$OUT/SUMMARY.md"
cat "$OUT/SUMMARY.md"
