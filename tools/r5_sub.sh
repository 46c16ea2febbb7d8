#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5s; mkdir -p $O; : > $O/ab.txt
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
  for b in 8 16; do
    for v in 0 1 8; do
      echo "SUB=$v B=$b" >> $O/ab.txt
      VOX_ROWSPLIT_SUB=$v LM_KV=200 timeout 300 python tools/lm_timing.py $b 60 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
    done
  done
  for v in 0 1 8; do
    echo "SUB=$v CSM16" >> $O/ab.txt
    VOX_ROWSPLIT_SUB=$v timeout 300 python tools/bench_csm.py --batch 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['lm_frame_graph_ms'], d['ms_per_step'])" >> $O/ab.txt
  done
done
paste - - < $O/ab.txt
