for rep in 1 2 3; do
for e in "" "HSA_ENABLE_INTERRUPT=0"; do
env $e python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$e]', round(d['ms_per_step']*1000,2), d['roofline']['device_ms_per_step'])"
done; done
