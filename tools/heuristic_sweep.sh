get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; }
for B in 0 9 11 14; do
 echo "bpc=$B: cfg2 $(PSVAE_COST_BPC=$B python bench.py --no-cpu-baseline --no-tune --steps 400 | get)  cfg4 $(PSVAE_COST_BPC=$B python bench.py --no-cpu-baseline --no-tune --input-dim 512 --latent-dim 128 --steps 300 | get)  B4096 $(PSVAE_COST_BPC=$B python bench.py --no-cpu-baseline --no-tune --batch 4096 --steps 200 | get)  B256 $(PSVAE_COST_BPC=$B python bench.py --no-cpu-baseline --no-tune --batch 256 --steps 400 | get)  decode $(PSVAE_COST_BPC=$B python bench.py --no-cpu-baseline --workload decode | get)"
done
