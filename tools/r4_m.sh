cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m; mkdir -p $O
for mode in 2 3 1; do
  echo "== PERSIST_MODE $mode"; PERSIST_MODE=$mode timeout 600 python tools/depth_persist_check.py 25 2>&1 | grep -v amdgpu.ids | tail -7
done > $O/persist_modes.txt 2>&1
cat $O/persist_modes.txt
