cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/q8f; mkdir -p $O
for B in 1 32; do
echo -n "default          " >> $O/times.txt; timeout 200 python $R/tools/lm_timing.py $B 200 2>&1 | tail -1 >> $O/times.txt
echo -n "LM_ONE_STREAM    " >> $O/times.txt; LM_ONE_STREAM=1 timeout 200 python $R/tools/lm_timing.py $B 200 2>&1 | tail -1 >> $O/times.txt
echo -n "NO_EXIT_FENCE    " >> $O/times.txt; VOX_NO_EXIT_FENCE=1 timeout 200 python $R/tools/lm_timing.py $B 200 2>&1 | tail -1 >> $O/times.txt
echo -n "HW_QUEUES=1      " >> $O/times.txt; GPU_MAX_HW_QUEUES=1 timeout 200 python $R/tools/lm_timing.py $B 200 2>&1 | tail -1 >> $O/times.txt
done
cat $O/times.txt
