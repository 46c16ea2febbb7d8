# kernel-level profile of the GLM-4-Voice B=8 development bench (LM graph + flow windows)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_glm -o glm -- python $GRAFT_REPO_ROOT/tools/bench_glm.py --steps 40 --greedy > $O/glm_prof.json 2> $O/glm_prof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_glm -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/glm_b8_kernel_stats.csv
rm -rf gpurun_out/prof_glm
head -12 gpurun_out/glm_b8_kernel_stats.csv | cut -c1-150
