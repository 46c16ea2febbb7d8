cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; mkdir -p $O
for B in 32 8 1; do
  rm -f /tmp/ru_$B.pt
  VOX_RES_UNIT=0 timeout 300 python tools/res_unit_check.py $B /tmp/ru_$B.pt 2>&1 | grep -v amdgpu.ids | tail -1
  VOX_RES_UNIT=1 timeout 300 python tools/res_unit_check.py $B /tmp/ru_$B.pt 2>&1 | grep -v amdgpu.ids | tail -1
done > $O/res_unit.txt 2>&1
cat $O/res_unit.txt
timeout 600 python -m pytest tests/test_gpu_codec.py -x -q -m gpu 2>&1 | tail -3
