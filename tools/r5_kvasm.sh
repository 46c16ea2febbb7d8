#!/bin/bash
# parity of the asm K/V fetch + same-box A/B against the previous build
cd "$(dirname "$0")/.."
O=gpurun_out/r5kv; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_qwen3.py tests/test_gpu_ops.py tests/test_gpu_lm.py tests/test_gpu_csm.py -m gpu -x -q 2>&1 | tail -5 > $O/tests.log
cat $O/tests.log
bash tools/r5_ab.sh tools/bin/libvoxhip_head.so vox_serve_amd/libvoxhip.so 1 8 32 > $O/ab.log 2>&1
grep -A1 "^lib=" gpurun_out/r5ab/ab.txt | grep -v "^--" | paste - - | awk '{print $1, $2, $8, $9, $10}' 
