# round 4: (1) depth chain + LM frame, round-3 library vs the new one on the same box; (2) where a steady-state serving step goes;
# (3) the new bench sub-results (kv_sweep, 250-frame serving jobs with the steady-state window)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
for rep in 1 2; do
  for L in tools/bin/libvoxhip_r3.so vox_serve_amd/libvoxhip.so; do
    echo "== $L"; VOX_LIB=$PWD/$L timeout 300 python tools/depth_stack_chain.py 4 2>&1 | grep -v amdgpu.ids | tail -1
    for B in 1 32; do VOX_LIB=$PWD/$L timeout 300 python tools/lm_timing.py $B 200 2>&1 | tail -1; done
  done
done > $O/ab_r3_vs_new.txt 2>&1
cat $O/ab_r3_vs_new.txt
timeout 600 python tools/serve_profile.py 31 150 > $O/serve_sync.txt 2>&1
timeout 600 python tools/serve_profile.py 31 150 async > $O/serve_async.txt 2>&1
head -60 $O/serve_sync.txt | grep -v amdgpu.ids
head -50 $O/serve_async.txt | grep -v amdgpu.ids
timeout 900 python bench.py --no-other-configs --no-cpu-baseline --serving-modes throughput --serving-ttfa-requests 5 --sub-batches 32 > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4c/bench.json") if l.startswith("{")][-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "roofline", "kv_sweep", "serving_path_throughput") if k in d}, indent=1)[:6000])
print("batch32", d.get("batch32", {}).get("value"), d.get("batch32", {}).get("ms_per_step"))
PY
