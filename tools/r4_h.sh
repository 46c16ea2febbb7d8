cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
VOX_LIB=$PWD/tools/bin/libvoxhip_dev.so timeout 600 python tools/depth_persist_stamps.py > $O/stamps.txt 2>&1
grep -v amdgpu.ids $O/stamps.txt | tail -12
timeout 600 python tools/depth_persist_check.py 30 > $O/persist_check.txt 2>&1
grep -v amdgpu.ids $O/persist_check.txt | tail -8
