bash tools/profile_step.sh r03_cfg2 > gpurun_out/prof_r03_cfg2.log 2>&1; tail -40 gpurun_out/prof_r03_cfg2.log
bash tools/profile_cfg3.sh > gpurun_out/prof_cfg3.log 2>&1; tail -30 gpurun_out/prof_cfg3.log
