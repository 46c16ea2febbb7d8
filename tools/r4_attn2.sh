# round 4: rewritten k_attn_decode8 / attn_short_wave — bit-exact suites, then phase stamps (v1 vs new), depth chain, frame times
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_qwen3.py tests/test_gpu_lm.py tests/test_gpu_csm.py -x -q -m gpu > $O/tests.txt 2>&1
tail -5 $O/tests.txt
export VOX_LIB=$PWD/tools/bin/libvoxhip_dev.so
for B in 1 32; do
  VOX_ATTN_V1=1 timeout 300 python tools/attn_stamps.py $B 20 200 > $O/stamps_v1_b$B.txt 2>&1
  timeout 300 python tools/attn_stamps.py $B 20 200 > $O/stamps_new_b$B.txt 2>&1
  VOX_ATTN_HS2_ROWS=64 timeout 300 python tools/attn_stamps.py $B 20 200 > $O/stamps_new_hs2_b$B.txt 2>&1
done
unset VOX_LIB
timeout 300 python tools/depth_stack_chain.py 4 > $O/depth_chain.txt 2>&1
for B in 1 8 32; do
 for rep in 1 2; do
  echo "B=$B new"; timeout 300 python tools/lm_timing.py $B 200 | tail -1
  echo "B=$B new hs2"; VOX_ATTN_HS2_ROWS=64 timeout 300 python tools/lm_timing.py $B 200 | tail -1
 done
done > $O/ab.txt 2>&1
grep -v amdgpu.ids $O/stamps_v1_b1.txt $O/stamps_new_b1.txt $O/stamps_new_hs2_b1.txt $O/stamps_v1_b32.txt $O/stamps_new_b32.txt $O/stamps_new_hs2_b32.txt $O/depth_chain.txt $O/ab.txt
