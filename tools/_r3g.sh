for rep in 1 2 3; do
for m in graph eager; do
  for st in "--steps 20 --warmup 5" "--steps 300 --warmup 30"; do
    echo -n "rep $rep mode $m $st: "
    python bench.py --mode $m --no-cpu-baseline $st 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['device_ms_per_step'])"
  done
done
done
