#!/bin/bash
# SQ counters of the bf16 weight-gradient launch, round 2's kernel against this round's forms (each counter group in its own
# rocprofv3 --pmc pass; cfg3 step, eager launches).  usage: tools/wgrad_counters.sh  -> gpurun_out/wgrad_counters.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/wgrad_counters.txt; : > $OUT
ARGS="--dtype bf16 --cond --batch 4096 --mode eager --steps 10 --warmup 4 --no-cpu-baseline"
for form in "PSVAE_WGRAD_V1=1" "PSVAE_WGRAD_BK=64" "PSVAE_WGRAD_BK=32" "PSVAE_WGRAD_LW=1"; do
  echo "== $form" >> $OUT
  for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
    rm -rf /tmp/wc; export $form
    timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d /tmp/wc -- python3 bench.py $ARGS > /dev/null 2> gpurun_out/wc.err
    unset ${form%%=*}
    python3 tools/pmc_kernel.py /tmp/wc "gemm_bf16_kernel<4, 4" >> $OUT 2>/dev/null
    python3 tools/pmc_kernel.py /tmp/wc wgrad_bf16 >> $OUT 2>/dev/null
  done
done
cat $OUT
