mkdir -p gpurun_out/r3full
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3full/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3full/pytest.log; tail -6 gpurun_out/r3full/pytest.log | cut -c1-300
