#!/bin/bash
# fused (PSVAE_STAGE2=3) vs unfused (0) step at other shapes, tuned each (same box)
run() { python bench.py --no-cpu-baseline --steps 300 --warmup 30 --tune "$@" 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['launches_per_step'], d['config']['launch_plan'])"; }
for m in 0 3 2 0 3; do
  echo "cfg4 shape mask $m: $(PSVAE_STAGE2=$m run --input-dim 512 --latent-dim 128)"
done
for m in 0 3 2 0 3; do
  echo "B=4096 mask $m: $(PSVAE_STAGE2=$m run --batch 4096)"
done
for m in 0 3; do
  echo "B=2048 mask $m: $(PSVAE_STAGE2=$m run --batch 2048)"
done
