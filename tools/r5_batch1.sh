#!/bin/bash
# round 5, GPU batch 1: kernarg-preload probe, chain traces (dev build), baseline frame timings
cd "$(dirname "$0")/.."
O=gpurun_out/r5a; mkdir -p $O
( tools/bin/kernarg_probe_off; echo ---- preload on; tools/bin/kernarg_probe_on ) > $O/kernarg_probe.txt 2>&1
for B in 1 8 32; do LM_KV=200 timeout 300 python tools/lm_timing.py $B 60 >> $O/lm_timing.txt 2>&1; done
VOX_LIB=tools/bin/libvoxhip_dev.so timeout 300 python tools/chain_trace.py 32 4 200 > $O/chain_trace_b32.txt 2>&1
VOX_LIB=tools/bin/libvoxhip_dev.so timeout 300 python tools/chain_trace.py 16 4 200 > $O/chain_trace_b16.txt 2>&1
VOX_LIB=tools/bin/libvoxhip_dev.so timeout 300 python tools/chain_trace.py 1 4 200 > $O/chain_trace_b1.txt 2>&1
tail -5 $O/kernarg_probe.txt $O/lm_timing.txt
