#!/bin/bash
# PMC passes of the cfg3 step (bf16, conditional, B = 4096; eager launches so that every kernel is a dispatch): SQ / TCC
# group, FETCH_SIZE, WRITE_SIZE -- each in its own run (MI355X_MICROARCH.md).  Summary -> gpurun_out/prof_cfg3/pmc.md
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_cfg3; RAW=/tmp/psvae_prof_cfg3
rm -rf "$OUT" "$RAW"; mkdir -p "$OUT" "$RAW"
ARGS="--dtype bf16 --cond --batch 4096 --mode eager --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$RAW/pmc_sq" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_sq.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$RAW/pmc_fetch" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$RAW/pmc_write" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_write.log"
python3 - "$RAW" > "$OUT/pmc.md" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
raw = sys.argv[1]
def short(n):
    return re.sub(r"\(.*", "", n).replace("void psvae::", "").replace("psvae::", "")[:48]
tab = defaultdict(dict)   # kernel position in the step -> counter -> value
order = []
for tag in ("pmc_sq", "pmc_fetch", "pmc_write"):
    f = glob.glob(raw + f"/{tag}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    by, names = defaultdict(lambda: defaultdict(float)), {}
    for r in csv.DictReader(open(f[0])):
        d = int(r["Dispatch_Id"]); by[d][r["Counter_Name"]] += float(r["Counter_Value"]); names[d] = r["Kernel_Name"]
    ids = [i for i in sorted(by) if "psvae" in names[i]]
    # the last complete step: from its first launch (the pack launch with PSVAE_BF16_PACK_IN_STEP=1, else the encoder chains)
    first = "chain_pack" if any("chain_pack" in names[i] for i in ids[-8:]) else "chain_fwd_bf16_kernel<8"
    pk = [k for k, i in enumerate(ids) if first in names[i]]
    step = ids[pk[-2]:pk[-1]]
    if not order:
        order = [short(names[i]) for i in step]
    for pos, i in enumerate(step):
        tab[pos].update(by[i])
print("# cfg3 step (bf16, conditional, B = 4096): rocprofv3 --pmc, one step, per launch\n")
print("| launch | FETCH_SIZE x2 (MB) | WRITE_SIZE (MB) | MFMA busy (us at 2.4 GHz, per SIMD) | L2 hit rate |\n|---|---|---|---|---|")
tf = tw = tm = 0.0
for pos, name in enumerate(order):
    c = tab[pos]
    f, w = 2 * c.get("FETCH_SIZE", 0) / 1024, c.get("WRITE_SIZE", 0) / 1024
    m = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / 2.4e3
    h = c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0))
    tf += f; tw += w; tm += m
    print(f"| `{name}` | {f:.1f} | {w:.1f} | {m:.1f} | {h:.2f} |")
print(f"| **step** | **{tf:.1f}** | **{tw:.1f}** | **{tm:.1f}** | |")
import json
json.dump({"hbm_bytes_per_step": int((tf + tw) * 1024 * 1024), "fetch_x2_mb": round(tf, 1), "write_mb": round(tw, 1),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `python bench.py --dtype bf16 --cond --batch 4096 --mode eager`, all launches of one step",
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md; Infinity-Cache hits are counted: L2-fill traffic, an upper bound on HBM bytes"},
          open(raw.replace("/tmp/psvae_prof_cfg3", "gpurun_out/prof_cfg3") + "/pmc.json", "w"), indent=1)
print("\n(FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 reports half of wide coalesced reads; Infinity-Cache hits are counted, so the"
      " figure is L2-fill traffic, an upper bound on HBM bytes.)")
PY
rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/trace" -- python3 bench.py --dtype bf16 --cond --batch 4096 --no-cpu-baseline > "$OUT/bench_under_trace.json" 2> "$OUT/trace.log"
find "$RAW/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
python3 tools/parse_trace.py "$RAW/trace" 100 > "$OUT/timeline.txt"
python3 bench.py --dtype bf16 --cond --batch 4096 > "$OUT/bench.json" 2> "$OUT/bench.log"
PSVAE_BF16_PACK_IN_STEP=1 python3 bench.py --dtype bf16 --cond --batch 4096 --no-cpu-baseline > "$OUT/bench_pack_in_step.json" 2>> "$OUT/bench.log"
cat "$OUT/pmc.md" "$OUT/timeline.txt"; tail -1 "$OUT/bench.json"
