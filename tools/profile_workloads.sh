#!/bin/bash
# rocprofv3 --kernel-trace --stats of the non-headline bench workloads (cfg5 decode fp32 / bf16, cfg5 conditional);
# per-kernel stats go to gpurun_out/prof_workloads/<name>_kernel_stats.csv, the bench lines beside them.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_workloads; rm -rf $OUT; mkdir -p $OUT
run() {
    name=$1; shift
    RAW=/tmp/psvae_prof_$name; rm -rf $RAW; mkdir -p $RAW
    rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -- python3 bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.log
    find $RAW -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
    head -6 $OUT/${name}_kernel_stats.csv | cut -c1-160
}
run decode_f32 --workload decode
run decode_bf16 --workload decode --dtype bf16
run langevin --workload langevin
