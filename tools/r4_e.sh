cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
timeout 600 python tools/depth_persist_check.py 40 > $O/persist_check.txt 2>&1
grep -v amdgpu.ids $O/persist_check.txt | tail -25
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1
tail -12 $O/gpu_suite.log
cat gpurun_out/parity_counts.json 2>/dev/null
