export VOX_LIB=$GRAFT_REPO_ROOT/tools/bin/libvoxhip_dev.so
mkdir -p gpurun_out/f3
for B in 32 1; do
for A in 0 2 18 34 66 130 3 6 4; do
  echo -n "B=$B ABLATE=$A: "; VOX_ABLATE=$A timeout 120 python tools/lm_timing.py $B 30 2>&1 | tail -1
done
done > gpurun_out/f3/ablate.txt 2>&1
cat gpurun_out/f3/ablate.txt
