#!/bin/bash
# round 5, GPU batch 3: attention inside the persistent talker layer (A/B + parity), GLM full-depth tape replay
cd "$(dirname "$0")/.."
O=gpurun_out/r5c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_qwen3.py -x -q > $O/qwen3_tests.log 2>&1; tail -6 $O/qwen3_tests.log
timeout 600 python -m pytest tests/test_gpu_lm.py -x -q -k "full_depth" > $O/glm_depth.log 2>&1; tail -4 $O/glm_depth.log
for rep in 1 2; do
  for v in 0 1; do
    echo "VOX_TALKER_ATTN=$v" >> $O/ab_attn.txt
    VOX_TALKER_ATTN=$v LM_KV=200 timeout 300 python tools/lm_timing.py 1 80 2>&1 | grep -v amdgpu.ids >> $O/ab_attn.txt
    VOX_TALKER_ATTN=$v LM_KV=40 timeout 300 python tools/lm_timing.py 1 80 2>&1 | grep -v amdgpu.ids >> $O/ab_attn.txt
  done
done
cat $O/ab_attn.txt
