#!/bin/bash
# Same-box A/B of library builds: tools/ab.sh ROUNDS lib1.so lib2.so ...   (boxes differ by a few %, so only
# numbers from ONE call are comparable).  Prints ms/step untuned and tuned per round.
R=$1; shift
for r in $(seq 1 $R); do
  for lib in "$@"; do
    u=$(PSVAE_LIB=$lib python bench.py --no-cpu-baseline --steps 400 --warmup 30 --no-tune | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    t=$(PSVAE_LIB=$lib python bench.py --no-cpu-baseline --steps 400 --warmup 30 | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "round $r $(basename $lib): untuned $u tuned $t"
  done
done
