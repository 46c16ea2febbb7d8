cd $GRAFT_REPO_ROOT
O=gpurun_out/r4o; mkdir -p $O
for rep in 1 2; do
 for L in tools/bin/libvoxhip_prev.so vox_serve_amd/libvoxhip.so; do
  echo "== $L"; VOX_LIB=$PWD/$L PERSIST_MODE=3 timeout 600 python tools/depth_persist_check.py 20 2>&1 | grep -E "launch chain|persistent  |final"
 done
done > $O/mlp_roles_ab.txt 2>&1
cat $O/mlp_roles_ab.txt
