cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3w
timeout -k 10 300 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -2
for nb in 2 3; do
PSVAE_WGRAD_NBUF=$nb PSVAE_LIB=variants/libpsvae_wstamp.so timeout -k 10 120 python tools/stamp_run.py > gpurun_out/r3w/stamps_lw_nb$nb.txt 2>&1
grep "wgrad-lw stamp wg 203" gpurun_out/r3w/stamps_lw_nb$nb.txt | tail -1
done
ARGS="--dtype bf16 --cond --batch 4096 --steps 200 --warmup 20 --no-cpu-baseline"
for cfg in "1 2" "1 3" "0 2"; do set -- $cfg
PSVAE_WGRAD_LW=$1 PSVAE_WGRAD_NBUF=$2 python3 bench.py $ARGS > gpurun_out/r3w/lw.json 2> gpurun_out/r3w/e.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r3w/lw.json').read().strip().splitlines()[-1]); print('lw=$1 nbuf=$2', d['ms_per_step']*1000)
"
done
rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py $ARGS > /dev/null 2> gpurun_out/r3w/s.err
python3 tools/parse_trace.py /tmp/tr 50 | tail -4
