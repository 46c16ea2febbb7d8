# round 4, talker decode attention: phase stamps (dev build) + A/B of the page-id hoist and the per-q-head split
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
export VOX_LIB=$PWD/tools/bin/libvoxhip_dev.so
for B in 1 32; do
  timeout 300 python tools/attn_stamps.py $B 20 200 > $O/stamps_b$B.txt 2>&1
  VOX_ATTN_HS2_ROWS=64 timeout 300 python tools/attn_stamps.py $B 20 200 > $O/stamps_hs2_b$B.txt 2>&1
done
unset VOX_LIB
for B in 1 8 32; do
 for rep in 1 2; do
  echo "B=$B base"; timeout 300 python tools/lm_timing.py $B 200 | tail -1
  echo "B=$B nohoist"; VOX_ATTN_HOIST=0 timeout 300 python tools/lm_timing.py $B 200 | tail -1
  echo "B=$B hs2"; VOX_ATTN_HS2_ROWS=64 timeout 300 python tools/lm_timing.py $B 200 | tail -1
 done
done > $O/ab.txt 2>&1
cat $O/stamps_b1.txt $O/stamps_hs2_b1.txt $O/stamps_b32.txt $O/ab.txt
