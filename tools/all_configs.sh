#!/bin/bash
# One line of bench JSON per measured configuration (un-profiled) -> gpurun_out/all_configs.jsonl + a markdown table.
# Shapes without a committed launch plan are tuned first (bench.py --tune with PSVAE_PLAN_CACHE) and the plans printed,
# so that they can be added to pseudo_speaker_vae_amd/launch_plans.json.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/all_configs.jsonl; : > $O
export PSVAE_PLAN_CACHE=gpurun_out/plans_tuned.json
j() { grep '"metric"'; }
python3 bench.py --no-cpu-baseline | j >> $O
python3 bench.py --no-cpu-baseline --plan untuned | j >> $O
python3 bench.py --no-cpu-baseline --with-adam | j >> $O
python3 bench.py --no-cpu-baseline --input-dim 512 --latent-dim 128 | j >> $O
python3 bench.py --no-cpu-baseline --batch 4096 | j >> $O
python3 bench.py --no-cpu-baseline --dtype bf16 --cond --batch 4096 | j >> $O
python3 bench.py --no-cpu-baseline --workload decode | j >> $O
python3 bench.py --no-cpu-baseline --workload decode --dtype bf16 | j >> $O
python3 bench.py --no-cpu-baseline --workload langevin | j >> $O
python3 - <<'PY'
import json
names = ["cfg2 train fp32 B=1024 d=256 z=64 (headline, committed plan)", "cfg2, heuristic plan (--plan untuned)", "cfg2 + Adam in the captured graph",
         "cfg4 per-GPU shape: fp32 B=1024 d=512 z=128", "cfg2 model at B=4096 (fp32)", "cfg3: bf16, one-hot conditional, B=4096",
         "cfg5: decode-only, 1M rows", "cfg5 in bf16 (psvae_decode_bf16)", "cfg5 conditional: 1M rows x 100 Langevin steps + decode"]
print("| configuration | embeddings/s | ms/step | launches | TFLOP/s | peak | frac |\n|---|---|---|---|---|---|---|")
for n, l in zip(names, open("gpurun_out/all_configs.jsonl")):
    d = json.loads(l); r = d["roofline"]
    print(f"| {n} | {d['value']:,.0f} | {d['ms_per_step']:.4f} | {d['config'].get('launches_per_step', '')} | {r['achieved']:.1f} | {r['peak']:.0f} | {r['frac']:.3f} |")
print()
for l in open("gpurun_out/all_configs.jsonl"):
    d = json.loads(l)
    if "launch_plan" in d["config"]:
        print(d["config"]["workload"][:60], "->", d["config"]["launch_plan"])
PY
cat gpurun_out/plans_tuned.json 2>/dev/null
