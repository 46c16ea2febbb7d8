#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_qwen3.py -x -q -k "persistent" > $O/persist_tests.log 2>&1; tail -3 $O/persist_tests.log
echo "VOX_TALKER_ATTN=0" >> $O/ab.txt
VOX_TALKER_ATTN=0 LM_KV=200 timeout 300 python tools/lm_timing.py 1 80 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
for d in 0 2 4 6 8 12; do
  echo "VOX_TALKER_ATTN=1 DELAY=$d" >> $O/ab.txt
  VOX_TALKER_ATTN_DELAY=$d LM_KV=200 timeout 300 python tools/lm_timing.py 1 80 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
done
echo "VOX_TALKER_ATTN=0" >> $O/ab.txt
VOX_TALKER_ATTN=0 LM_KV=200 timeout 300 python tools/lm_timing.py 1 80 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
cat $O/ab.txt
for d in 0 4 8; do echo "DELAY=$d"; VOX_TALKER_ATTN_DELAY=$d VOX_LIB=tools/bin/libvoxhip_dev.so timeout 300 python tools/mlp_trace.py 200 2>&1 | grep -v amdgpu.ids; done | tee $O/mlp_trace.txt
