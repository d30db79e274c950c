cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3w
ARGS="--dtype bf16 --cond --batch 4096 --mode eager --steps 12 --warmup 4 --no-cpu-baseline"
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU --output-format csv -d /tmp/p1 -- python3 bench.py $ARGS > /dev/null 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/p2 -- python3 bench.py $ARGS > /dev/null 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/p3 -- python3 bench.py $ARGS > /dev/null 2>&1
python3 tools/pmc_step_table.py /tmp/p1 /tmp/p2 /tmp/p3 > gpurun_out/r3w/cfg3_inst_table.md
cat gpurun_out/r3w/cfg3_inst_table.md
