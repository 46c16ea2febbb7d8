cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O
for v in base nobar sleep2 pw2 pw1 pw4s1; do
  echo "== variant $v"
  VOX_LIB=$PWD/tools/bin/ds_$v.so timeout 300 python tools/depth_persist_stamps.py 2>&1 | grep -v amdgpu.ids | tail -10
done > $O/variants.txt 2>&1
cat $O/variants.txt
