#!/bin/bash
# 1-rank RCCL timing of the data-parallel modes of bench.py on one GPU (what the exchange costs beyond the step's graph)
run() { python bench.py --no-cpu-baseline --steps 300 --warmup 30 "$@" 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['launch'], '|', d['config']['allreduce'])"; }
echo "no group:            $(run)"
echo "no group + adam:     $(run --with-adam)"
export PSVAE_BENCH_FORCE_DIST=1
echo "graph + allreduce:   $(run)"
echo "graph + ar + adam:   $(run --with-adam)"
echo "in-graph allreduce:  $(PSVAE_DDP_IN_GRAPH=1 run)"
echo "in-graph, 2 buckets: $(PSVAE_DDP_IN_GRAPH=2 run)"
echo "two graphs, eager ar:$(PSVAE_DDP_OVERLAP=1 run)"
echo "eager split:         $(run --mode eager)"
