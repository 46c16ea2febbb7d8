#!/bin/bash
# same-box A/B/C of library variants at B=1: bash tools/r5_ab3.sh <lib>...
cd "$(dirname "$0")/.."
O=gpurun_out/r5ab; mkdir -p $O; : > $O/ab3.txt
for rep in 1 2 3; do
  for lib in "$@"; do
    echo -n "lib=$lib " >> $O/ab3.txt
    VOX_LIB=$lib LM_KV=200 timeout 300 python tools/lm_timing.py ${AB_B:-1} 60 2>&1 | grep -v amdgpu.ids | awk '{print $5, $6, $7}' >> $O/ab3.txt
  done
done
cat $O/ab3.txt
