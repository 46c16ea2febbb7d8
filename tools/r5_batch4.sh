#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_qwen3.py -x -q -k "persistent" > $O/persist_tests.log 2>&1; tail -4 $O/persist_tests.log
for rep in 1 2; do
  for v in 0 1; do
    echo "VOX_TALKER_ATTN=$v" >> $O/ab_attn.txt
    VOX_TALKER_ATTN=$v LM_KV=200 timeout 300 python tools/lm_timing.py 1 80 2>&1 | grep -v amdgpu.ids >> $O/ab_attn.txt
  done
done
cat $O/ab_attn.txt
for v in 0 1; do VOX_TALKER_ATTN=$v VOX_LIB=tools/bin/libvoxhip_dev.so timeout 300 python tools/mlp_trace.py 200 2>&1 | grep -v amdgpu.ids | tee -a $O/mlp_trace.txt; done
