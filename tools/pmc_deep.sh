#!/bin/bash
# Deeper PMC passes over the step's GEMM launches (one counter group per pass).  Run on the GPU box.
set -e
TAG=${1:-deep}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
run() {  # name counters...
    local name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --mode eager --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2> "$OUT/$name.log"
    python3 tools/parse_pmc.py "$OUT/$name" > "$OUT/$name.txt"
}
run sq1 SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32
run sq2 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum
run ta TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TCP_TCR_RDRET_STALL_sum SQ_VMEM_TA_ADDR_FIFO_FULL
cat "$OUT"/*.txt
