# frame with parts removed (dev-knob build; VOX_ABLATE bits: 1 attention, 2 depth loop, 4 talker layers, 8 samplers, 16 qkv, 32 o_proj, 64 gate/up,
# 128 down, 256 separate depth attention kernel, 2048 depth heads, 4096 depth step 1)
export VOX_LIB=$GRAFT_REPO_ROOT/tools/bin/libvoxhip_dev.so
O=$GRAFT_REPO_ROOT/gpurun_out/ablate; mkdir -p $O
for B in 1 32; do
for A in 0 2 4 6 12 2060 269 20 36 68 132 5 3; do
  echo -n "B=$B ABLATE=$A: "; VOX_ABLATE=$A timeout 120 python tools/lm_timing.py $B 100 2>&1 | tail -1
done
done > $O/ablate.txt 2>&1
cat $O/ablate.txt
