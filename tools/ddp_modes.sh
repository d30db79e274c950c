#!/bin/bash
# 1-rank RCCL / direct-IPC timing of the data-parallel forms of bench.py (one GPU: the exchange moves nothing, what shows
# is each form's fixed cost).  usage: tools/ddp_modes.sh
export PSVAE_BENCH_FORCE_DIST=1
run() { echo -n "$1: "; shift; env "$@" python bench.py --no-cpu-baseline --steps 300 --warmup 30 $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], '|', d['config']['launch'], '|', d['config']['allreduce'][:70], '|', d['config'].get('calibration'))"; }
EXTRA="" run "no process group          " PSVAE_BENCH_FORCE_DIST=0
EXTRA="" run "auto (calibrated), RCCL   " A=1
EXTRA="--allreduce direct" run "auto (calibrated), direct " A=1
EXTRA="--mode graph" run "graph + RCCL all-reduce   " A=1
EXTRA="--mode graph --allreduce direct" run "graph + direct exchange   " A=1
EXTRA="--mode graph" run "all-reduce inside graph   " PSVAE_DDP_IN_GRAPH=1
EXTRA="--mode eager" run "phase split, 2 buckets    " A=1
EXTRA="--mode overlapped" run "overlapped (side stream)  " A=1
