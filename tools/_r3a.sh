set -x
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
tail -5 gpurun_out/r3a/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench20.json 2> gpurun_out/r3a/bench20.err; tail -1 gpurun_out/r3a/bench20.json
python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; tail -1 gpurun_out/r3a/bench.json
