#!/bin/bash
# One line of bench JSON per measured configuration (un-profiled) -> gpurun_out/all_configs.jsonl
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/all_configs.jsonl; : > $O
python3 bench.py --no-cpu-baseline >> $O
python3 bench.py --no-cpu-baseline --no-tune >> $O
python3 bench.py --no-cpu-baseline --with-adam >> $O
python3 bench.py --no-cpu-baseline --input-dim 512 --latent-dim 128 >> $O
python3 bench.py --no-cpu-baseline --batch 4096 >> $O
python3 bench.py --no-cpu-baseline --dtype bf16 --cond --batch 4096 >> $O
python3 bench.py --no-cpu-baseline --workload decode >> $O
python3 bench.py --no-cpu-baseline --workload decode --dtype bf16 >> $O
python3 - <<'PY'
import json
names = ["cfg2 train fp32 B=1024 d=256 z=64 (headline, autotuned plan)", "cfg2, heuristic plan (--no-tune)", "cfg2 + fused Adam",
         "cfg4 per-GPU shape: fp32 B=1024 d=512 z=128", "cfg2 model at B=4096 (fp32)", "cfg3: bf16, one-hot conditional, B=4096",
         "cfg5: decode-only, 1M rows", "cfg5 in bf16 (psvae_decode_bf16)"]
print("| configuration | embeddings/s | ms/step | TFLOP/s | peak | frac |\n|---|---|---|---|---|---|")
for n, l in zip(names, open("gpurun_out/all_configs.jsonl")):
    d = json.loads(l); r = d["roofline"]
    print(f"| {n} | {d['value']:,.0f} | {d['ms_per_step']:.4f} | {r['achieved']:.1f} | {r['peak']:.0f} | {r['frac']:.3f} |")
PY
