mkdir -p gpurun_out/r3full
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3full/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3full/pytest.log; grep -E "^FAILED|passed|failed" gpurun_out/r3full/pytest.log | cut -c1-200 | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --steps 300 --warmup 30 2>/dev/null > gpurun_out/r3full/bench300.json; python3 -c "
import json; d=json.loads(open('gpurun_out/r3full/bench300.json').read().strip().splitlines()[-1]); print('steady', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v.get('ms_per_step') for k,v in d['secondary'].items()})"
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style', d['value'], d['ms_per_step'], d['roofline']['frac'])"
