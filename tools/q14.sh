# per-kernel stats of the detokenizer chunks at B = 8 (CosyVoice2, GLM) and B = 1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q14; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_hift.py tests/test_gpu_glm_decoder.py -q -x 2>&1 | tail -3) > $O/parity.log; cat $O/parity.log
for b in 1 8; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cv$b -o cv -- python tools/bench_cosyvoice2.py --batch $b --steps 50 --warmup 0 > $O/cv_b${b}_prof.json 2> $O/cv_b${b}_prof.err
cp $(find $O/prof_cv$b -name "*kernel_stats.csv" | head -1) $O/kernel_stats_cosyvoice2_b$b.csv; rm -rf $O/prof_cv$b
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_glm -o glm -- python tools/bench_glm.py --batch 8 --greedy --steps 80 --warmup 10 > $O/glm_b8_prof.json 2> $O/glm_b8_prof.err
cp $(find $O/prof_glm -name "*kernel_stats.csv" | head -1) $O/kernel_stats_glm_b8.csv; rm -rf $O/prof_glm
for b in 1 8; do timeout 300 python tools/bench_cosyvoice2.py --batch $b > $O/cv_b$b.json 2> $O/cv_b$b.err; done
timeout 300 python tools/bench_glm.py --batch 8 --greedy --steps 150 > $O/glm_b8.json 2> $O/glm_b8.err
tail -c 300 $O/cv_b1.json $O/cv_b8.json $O/glm_b8.json
