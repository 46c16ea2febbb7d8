cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
timeout 600 python tools/depth_persist_check.py 30 > $O/persist_check.txt 2>&1
grep -v amdgpu.ids $O/persist_check.txt | tail -12
