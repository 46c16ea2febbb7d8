cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/q10; mkdir -p $O
cd $R
for b in 1 8; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cv$b -o cv -- python tools/bench_cosyvoice2.py --batch $b --steps 50 --warmup 0 > $O/cv${b}_prof.json 2> $O/cv${b}_prof.err
f=$(find $O/prof_cv$b -name "*kernel_stats.csv" | head -1); cp $f $O/cv${b}_kernel_stats.csv
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_glm1 -o glm -- python tools/bench_glm.py --batch 1 --greedy --steps 40 --warmup 10 > $O/glm1_prof.json 2> $O/glm1_prof.err
f=$(find $O/prof_glm1 -name "*kernel_stats.csv" | head -1); cp $f $O/glm1_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
for f in cv1 cv8 glm1; do echo "== $f"; head -14 $O/${f}_kernel_stats.csv | cut -c1-160; done
