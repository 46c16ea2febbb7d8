#!/bin/bash
# Same-box A/B of one environment switch (run through gpurun): bash tools/ab_knob.sh <KNOB> '<command>' [repeats]
#   e.g.  bash tools/ab_knob.sh VOX_DEPTH_PICK 'LM_KV=200 python tools/lm_timing.py' 3
#         bash tools/ab_knob.sh VOX_CODEC_ATTN2 'python tools/codec_timing.py 32 10'
# The command runs with KNOB=1 and KNOB=0 alternately (box-to-box differences are +-2..3 %: only same-box pairs mean anything);
# the last line of its output is printed.  VOX_LIB=<path> A/Bs two builds instead (tools/bin/ is git-ignored).
knob=$1; cmd=$2; n=${3:-3}
for i in $(seq $n); do
  for v in 1 0; do echo -n "$knob=$v  "; env $knob=$v bash -c "$cmd" 2>/dev/null | tail -1; done
done
