#!/bin/bash
# round 5, GPU batch 2: fail-loud persistent kernels (new tests), then the whole GPU suite, a bench sanity line
cd "$(dirname "$0")/.."
O=gpurun_out/r5b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_qwen3.py -x -q -k "handoff or second_stream or persistent" > $O/persist_tests.log 2>&1
tail -15 $O/persist_tests.log
timeout 900 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1
tail -8 $O/gpu_suite.log
for B in 1 32; do LM_KV=200 timeout 300 python tools/lm_timing.py $B 60 2>&1 | grep -v amdgpu.ids >> $O/lm_timing.txt; done
cat $O/lm_timing.txt
