#!/bin/bash
# same-box A/B of weight-gradient schedules (PSVAE_WSCHED) and stage-2 masks, heuristic launch plan, ROUNDS rounds
R=${1:-2}
run() { python bench.py --no-cpu-baseline --steps 400 --warmup 40 --plan untuned | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['launches_per_step'])"; }
for r in $(seq 1 $R); do
  echo "round $r unfused(mask0): $(PSVAE_STAGE2=0 run)"
  echo "round $r mask2: $(PSVAE_STAGE2=2 run)"
  echo "round $r S0 default 4,3,4,4,3,4,4,2,1: $(run)"
  echo "round $r S1 4,3,3,4,3,3,4,2,1: $(PSVAE_WSCHED=4,3,3,4,3,3,4,2,1 run)"
  echo "round $r S2 4,4,4,4,4,4,3,3,2: $(PSVAE_WSCHED=4,4,4,4,4,4,3,3,2 run)"
  echo "round $r S3 4,4,4,4,4,4,3,2,1: $(PSVAE_WSCHED=4,4,4,4,4,4,3,2,1 run)"
  echo "round $r S4 4,4,3,4,4,3,3,2,1: $(PSVAE_WSCHED=4,4,3,4,4,3,3,2,1 run)"
  echo "round $r S5 4,3,4,4,3,4,3,2,1: $(PSVAE_WSCHED=4,3,4,4,3,4,3,2,1 run)"
  echo "round $r S6 4,3,3,4,3,3,4,3,2: $(PSVAE_WSCHED=4,3,3,4,3,3,4,3,2 run)"
done
