cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
VOX_LIB=$PWD/tools/bin/libvoxhip_dev.so timeout 600 python tools/depth_persist_stamps.py > $O/stamps.txt 2>&1
grep -v amdgpu.ids $O/stamps.txt | tail -14
