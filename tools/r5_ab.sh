#!/bin/bash
# same-box A/B of two library builds: bash tools/r5_ab.sh <libA> <libB> [batches...]
cd "$(dirname "$0")/.."
O=gpurun_out/r5ab; mkdir -p $O; : > $O/ab.txt
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for b in "$@"; do
    for lib in $A $B; do
      echo "lib=$lib B=$b" >> $O/ab.txt
      VOX_LIB=$lib LM_KV=200 timeout 300 python tools/lm_timing.py $b 60 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
    done
  done
done
cat $O/ab.txt
