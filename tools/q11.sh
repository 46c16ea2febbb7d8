# session-5 baseline: per-kernel stats of the detokenizer chunks and the LM frames on the current build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q11; mkdir -p $O
Q="--no-cpu-baseline --ttfa-requests 0 --serving-ttfa-requests 0 --no-other-configs"
for b in 1 8; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cv$b -o cv -- python tools/bench_cosyvoice2.py --batch $b --steps 50 --warmup 0 > $O/cv_b${b}_prof.json 2> $O/cv_b${b}_prof.err
cp $(find $O/prof_cv$b -name "*kernel_stats.csv" | head -1) $O/kernel_stats_cosyvoice2_b$b.csv; rm -rf $O/prof_cv$b
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_glm -o glm -- python tools/bench_glm.py --batch 8 --greedy --steps 80 --warmup 10 > $O/glm_b8_prof.json 2> $O/glm_b8_prof.err
cp $(find $O/prof_glm -name "*kernel_stats.csv" | head -1) $O/kernel_stats_glm_b8.csv; rm -rf $O/prof_glm
for b in 1 32; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$b -o b$b -- python bench.py --batch $b --steps 40 --warmup 10 $Q > $O/bench_b${b}_prof.json 2> $O/bench_b${b}_prof.err
  python tools/trace_summary.py $(find $O/prof_b$b -name "*kernel_trace.csv" | head -1) 60000 > $O/trace_summary_b$b.txt 2>&1
  rm -rf $O/prof_b$b
done
for b in 1 8; do timeout 300 python tools/bench_cosyvoice2.py --batch $b > $O/cv_b$b.json 2> $O/cv_b$b.err; done
tail -c 400 $O/cv_b1.json $O/cv_b8.json
