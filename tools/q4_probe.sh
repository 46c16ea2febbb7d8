mkdir -p gpurun_out/q4
export VOX_LIB=tools/bin/libvoxhip_dev.so
for pn in 0 1; do for B in 32 17; do
  echo "PRENORM2048=$pn B=$B" >> gpurun_out/q4/prenorm.log
  VOX_PRENORM2048=$pn timeout 200 python tools/lm_timing.py $B 60 2>&1 | tail -1 >> gpurun_out/q4/prenorm.log
  VOX_PRENORM2048=$pn VOX_ABLATE=2 timeout 200 python tools/lm_timing.py $B 60 2>&1 | tail -1 >> gpurun_out/q4/prenorm.log
done; done
