# Round-6 profile refresh (run on the GPU box through gpurun; outputs under gpurun_out/p6, copied into profiles/ by hand).
# Counters are collected in their own rocprofv3 passes (--pmc with --kernel-trace only), one counter per pass.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/p6; mkdir -p $O
Q="--no-cpu-baseline --ttfa-requests 0 --serving-ttfa-requests 0 --no-other-configs"
for b in 1 8 32; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$b -o b$b -- python bench.py --batch $b --steps 40 --warmup 10 $Q > $O/bench_b${b}_prof.json 2> $O/bench_b${b}_prof.err
  cp $(find $O/prof_b$b -name "*kernel_stats.csv" | head -1) $O/kernel_stats_b$b.csv
  python tools/trace_summary.py $(find $O/prof_b$b -name "*kernel_trace.csv" | head -1) 60000 > $O/trace_summary_b$b.txt 2>&1
  rm -rf $O/prof_b$b
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${b}_$c -o p -- python bench.py --batch $b --steps 30 --warmup 5 $Q > $O/pmc_${b}_$c.log 2>&1
    python tools/pmc_summary.py $(find $O/pmc_${b}_$c -name "*counter_collection.csv" | head -1) $c > $O/pmc_${b}_$c.json 2>&1
    rm -rf $O/pmc_${b}_$c
  done
done
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- python bench.py --batch 32 --steps 20 --warmup 5 $Q > $O/pmc_mfma.log 2>&1
python tools/mfma_summary.py $(find $O/pmc_mfma -name "*counter_collection.csv") $(find $O/pmc_mfma -name "*kernel_trace.csv") > $O/mfma_b32.json 2>&1
rm -rf $O/pmc_mfma
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json
# BASELINE configs 3 and 4 (development benches; one JSON line each with a roofline block) + their kernel summaries
for b in 1 16; do timeout 600 python tools/bench_csm.py --batch $b > $O/csm_b$b.json 2> $O/csm_b$b.err; done
for b in 1 8; do timeout 600 python tools/bench_glm.py --batch $b --greedy --steps 150 > $O/glm_b$b.json 2> $O/glm_b$b.err; done
for b in 1 8; do timeout 600 python tools/bench_cosyvoice2.py --batch $b > $O/cosyvoice2_b$b.json 2> $O/cosyvoice2_b$b.err; done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cv -o cv -- python tools/bench_cosyvoice2.py --batch 1 --steps 50 --warmup 0 > $O/cv_prof.json 2> $O/cv_prof.err
cp $(find $O/prof_cv -name "*kernel_stats.csv" | head -1) $O/kernel_stats_cosyvoice2_b1.csv; rm -rf $O/prof_cv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cv8 -o cv -- python tools/bench_cosyvoice2.py --batch 8 --steps 50 --warmup 0 > $O/cv8_prof.json 2> $O/cv8_prof.err
cp $(find $O/prof_cv8 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_cosyvoice2_b8.csv; rm -rf $O/prof_cv8
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_csm -o csm -- python tools/bench_csm.py --batch 16 --steps 40 --warmup 10 > $O/csm_b16_prof.json 2> $O/csm_b16_prof.err
cp $(find $O/prof_csm -name "*kernel_stats.csv" | head -1) $O/kernel_stats_csm_b16.csv; rm -rf $O/prof_csm
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_glm -o glm -- python tools/bench_glm.py --batch 8 --greedy --steps 40 --warmup 10 > $O/glm_b8_prof.json 2> $O/glm_b8_prof.err
cp $(find $O/prof_glm -name "*kernel_stats.csv" | head -1) $O/kernel_stats_glm_b8.csv; rm -rf $O/prof_glm
# HBM traffic of the CSM-1B (B = 16) and GLM-4-Voice (B = 8) LM graphs: separate FETCH_SIZE / WRITE_SIZE passes
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_csm_$c -o p -- python tools/bench_csm.py --batch 16 --steps 20 --warmup 5 > $O/pmc_csm_$c.log 2>&1
  python tools/pmc_summary.py $(find $O/pmc_csm_$c -name "*counter_collection.csv" | head -1) $c k_csm_feedback > $O/pmc_csm16_$c.json 2>&1; rm -rf $O/pmc_csm_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_glm_$c -o p -- python tools/bench_glm.py --batch 8 --greedy --steps 20 --warmup 5 > $O/pmc_glm_$c.log 2>&1
  python tools/pmc_summary.py $(find $O/pmc_glm_$c -name "*counter_collection.csv" | head -1) $c k_lm_feedback > $O/pmc_glm8_$c.json 2>&1; rm -rf $O/pmc_glm_$c
done
# voice-clone prompt side: timings + kernel summary
timeout 300 python tools/bench_clone.py > $O/clone.json 2> $O/clone.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_clone -o clone -- python tools/bench_clone.py --reps 5 --seconds 5 > $O/clone_prof.json 2> $O/clone_prof.err
cp $(find $O/prof_clone -name "*kernel_stats.csv" | head -1) $O/kernel_stats_clone.csv; rm -rf $O/prof_clone
cat $O/csm_b16.json $O/glm_b8.json $O/clone.json
# the GPU suite on the same library
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_suite.log 2>&1
tail -5 $O/gpu_suite.log
