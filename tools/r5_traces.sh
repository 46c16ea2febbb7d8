#!/bin/bash
# round 5: in-kernel chain traces committed under profiles/ (development build)
cd "$(dirname "$0")/.."
O=gpurun_out/r5t; mkdir -p $O
export VOX_LIB=tools/bin/libvoxhip_dev.so
for B in 1 16 32; do timeout 300 python tools/chain_trace.py $B 4 200 2>&1 | grep -v amdgpu.ids > $O/chain_trace_b$B.txt; done
VOX_TRACE=1 timeout 400 python tools/bench_csm.py --batch 16 --warmup 12 2>&1 | grep -v amdgpu.ids > $O/chain_trace_csm_b16.txt
( for v in 0 1; do echo "VOX_TALKER_ATTN=$v"; VOX_TALKER_ATTN=$v timeout 300 python tools/mlp_trace.py 200 2>&1 | grep -v amdgpu.ids; done ) > $O/talker_layer_stamps.txt
( VOX_TALKER_ATTN=1 timeout 300 python tools/attn_in_layer_stamps.py 200 2>&1 | grep -v amdgpu.ids; echo "--- stand-alone launch"; VOX_TALKER_ATTN=0 timeout 300 python tools/attn_stamps.py 1 10 200 2>&1 | grep -v amdgpu.ids ) > $O/attn_in_layer.txt
unset VOX_LIB
( for cfg in "16 2" "8 4"; do set -- $cfg; echo "B per engine $1, engines $2"; timeout 300 python tools/dual_stream_timing.py $1 60 $2 2>&1 | grep -v amdgpu.ids | tail -1; done; for B in 16 32; do LM_KV=200 timeout 300 python tools/lm_timing.py $B 60 2>&1 | grep -v amdgpu.ids; done ) > $O/dual_stream.txt
( tools/bin/kernarg_probe_off; echo "---- -mllvm -amdgpu-kernarg-preload-count=16"; tools/bin/kernarg_probe_on ) > $O/kernarg_probe.txt 2>&1
tail -3 $O/*.txt | head -60
