#!/bin/bash
# same-box: tune the launch plan (bench.py --tune) for the unfused (PSVAE_STAGE2=0) and the fused (3) step, R rounds each
R=${1:-2}
for r in $(seq 1 $R); do
  for m in 0 3; do
    PSVAE_STAGE2=$m python bench.py --no-cpu-baseline --steps 400 --warmup 40 --tune 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('round $r mask $m:', d['ms_per_step'], d['config']['launches_per_step'], d['config']['launch_plan'])"
  done
done
