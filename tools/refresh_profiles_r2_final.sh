# Final (trimmed) round-2 refresh: the default bench line, the B=1 kernel summary, the detokenizer configs and the voice-clone prompt side.
# (The PMC traffic / MFMA-utilisation passes of tools/refresh_profiles_r2.sh were taken earlier in the round on the same LM kernels.)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/p2; mkdir -p $O
Q="--no-cpu-baseline --ttfa-requests 0 --serving-ttfa-requests 0 --no-other-configs"
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -o b1 -- python bench.py --batch 1 --steps 40 --warmup 10 $Q > $O/bench_b1_prof.json 2> $O/bench_b1_prof.err
cp $(find $O/prof_b1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_b1.csv
python tools/trace_summary.py $(find $O/prof_b1 -name "*kernel_trace.csv" | head -1) 60000 > $O/trace_summary_b1.txt 2>&1
rm -rf $O/prof_b1
for b in 1; do timeout 120 python tools/bench_cosyvoice2.py --batch $b > $O/cosyvoice2_b$b.json 2> $O/cosyvoice2_b$b.err; done
for b in 1; do timeout 120 python tools/bench_glm.py --batch $b --greedy --steps 100 > $O/glm_b$b.json 2> $O/glm_b$b.err; done
timeout 100 python tools/bench_clone.py > $O/clone.json 2> $O/clone.err
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cv -o cv -- python tools/bench_cosyvoice2.py --batch 1 --steps 50 --warmup 0 > $O/cv_prof.json 2> $O/cv_prof.err
cp $(find $O/prof_cv -name "*kernel_stats.csv" | head -1) $O/kernel_stats_cosyvoice2_b1.csv; rm -rf $O/prof_cv
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_clone -o clone -- python tools/bench_clone.py --reps 5 --seconds 5 > $O/clone_prof.json 2> $O/clone_prof.err
cp $(find $O/prof_clone -name "*kernel_stats.csv" | head -1) $O/kernel_stats_clone.csv; rm -rf $O/prof_clone
tail -c 400 $O/bench_default.json; cat $O/clone.json; cut -c1-400 $O/cosyvoice2_b1.json
