# round-3 closing run on the GPU box: the whole GPU suite under `time`, smoke(), then the profile refresh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/suite
(time python -m pytest tests/ -x -q -m gpu --durations=25) > gpurun_out/suite/gpu_suite.log 2>&1; tail -4 gpurun_out/suite/gpu_suite.log
python __graft_entry__.py smoke > gpurun_out/suite/smoke.log 2>&1; tail -1 gpurun_out/suite/smoke.log
bash tools/refresh_profiles_r3.sh > gpurun_out/suite/refresh.log 2>&1; tail -3 gpurun_out/suite/refresh.log
