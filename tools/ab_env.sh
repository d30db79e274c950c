#!/bin/bash
# Same-box A/B of environment settings: tools/ab_env.sh ROUNDS "A=1 B=2" "A=0" ...  ("-" = no extra environment).
# Prints ms/step untuned and tuned per round (only numbers from ONE gpurun call are comparable).
R=$1; shift
get() { python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for r in $(seq 1 $R); do
  for e in "$@"; do
    ev=$e; [ "$e" = "-" ] && ev=""
    u=$(env $ev python bench.py --no-cpu-baseline --steps 400 --warmup 30 --no-tune | get)
    t=$(env $ev python bench.py --no-cpu-baseline --steps 400 --warmup 30 | get)
    echo "round $r [$e]: untuned $u tuned $t"
  done
done
