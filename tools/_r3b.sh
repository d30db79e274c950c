set -x
mkdir -p gpurun_out/r3b
python -m pytest tests/test_gpu_api.py tests/test_gpu_fused_step.py -x -q > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3b/pytest.log
tail -15 gpurun_out/r3b/pytest.log
python tools/api_step_profile.py > gpurun_out/r3b/profile_graph.txt 2>&1; cat gpurun_out/r3b/profile_graph.txt
PSVAE_API_GRAPH=0 python tools/api_step_profile.py > gpurun_out/r3b/profile_eager.txt 2>&1; cat gpurun_out/r3b/profile_eager.txt
python tools/api_step_profile.py --with-adam > gpurun_out/r3b/profile_graph_adam.txt 2>&1; cat gpurun_out/r3b/profile_graph_adam.txt
python bench.py --workload api-step > gpurun_out/r3b/bench_api.json 2> gpurun_out/r3b/bench_api.err; tail -1 gpurun_out/r3b/bench_api.json; tail -3 gpurun_out/r3b/bench_api.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r3b/bench20.json 2> gpurun_out/r3b/bench20.err; tail -1 gpurun_out/r3b/bench20.json
