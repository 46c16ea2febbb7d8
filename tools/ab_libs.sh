#!/bin/bash
# same-box comparison of several library builds at one batch size: bash tools/ab_libs.sh <batch> <reps> lib1 lib2 ...  (prints gpu ms/frame per lib per rep)
cd "$(dirname "$0")/.."
b=$1; reps=$2; shift 2
for rep in $(seq $reps); do
  for lib in "$@"; do
    echo -n "B=$b $(basename $lib): "
    VOX_LIB=$lib LM_KV=${LM_KV:-200} timeout 300 python tools/lm_timing.py $b 60 2>&1 | grep -o "gpu [0-9.]* ms/frame"
  done
done
