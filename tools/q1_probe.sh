set -x
mkdir -p gpurun_out/q1
timeout 900 python -m pytest tests/test_gpu_worker.py tests/test_gpu_icl.py -x -q 2>&1 | tail -8 > gpurun_out/q1/worker_tests.log
export VOX_LIB=tools/bin/libvoxhip_dev.so
for ab in 0 256 2304 4352 6400 2; do
  for B in 1 32; do
    echo "ABLATE=$ab B=$B" >> gpurun_out/q1/ablate.log
    VOX_ABLATE=$ab timeout 200 python tools/lm_timing.py $B 60 2>&1 | tail -2 >> gpurun_out/q1/ablate.log
  done
done
