mkdir -p gpurun_out/r3h
python -m pytest tests/test_gpu_decode_fused.py -x -q 2>&1 | tail -12
for dt in f32 bf16; do
  for sep in 1 0; do
    echo -n "decode $dt separate_draw=$sep: "
    PSVAE_DECODE_SEPARATE_DRAW=$sep python bench.py --workload decode --dtype $dt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
  done
done
