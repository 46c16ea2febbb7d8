cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v8
for b in 1 8 32; do
  timeout 600 python bench.py --batch $b > gpurun_out/v8/bench_b$b.json 2> gpurun_out/v8/bench_b$b.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/v8/pmc_${b}_$c
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/v8/pmc_${b}_$c -o p -- python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --ttfa-requests 0 > gpurun_out/v8/pmc_${b}_$c.log 2>&1
    f=$(find gpurun_out/v8/pmc_${b}_$c -name "*counter_collection.csv" | head -1)
    python tools/pmc_summary.py $f $c > gpurun_out/v8/pmc_${b}_$c.json 2>&1
    rm -rf gpurun_out/v8/pmc_${b}_$c
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v8/prof_b1 -o b1 -- python bench.py --batch 1 > gpurun_out/v8/bench_b1_prof.json 2> gpurun_out/v8/bench_b1_prof.err
python tools/trace_summary.py gpurun_out/v8/prof_b1/b1_kernel_trace.csv 60000 > gpurun_out/v8/prof_b1_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v8/prof_b32 -o b32 -- python bench.py --batch 32 --no-cpu-baseline --ttfa-requests 0 > gpurun_out/v8/bench_b32_prof.json 2> gpurun_out/v8/bench_b32_prof.err
python tools/trace_summary.py gpurun_out/v8/prof_b32/b32_kernel_trace.csv 120000 > gpurun_out/v8/prof_b32_summary.txt
rm -f gpurun_out/v8/prof_b1/b1_kernel_trace.csv gpurun_out/v8/prof_b32/b32_kernel_trace.csv
timeout 600 python tools/bench_csm.py > gpurun_out/v8/csm.json 2>&1
for b in 1 8 32; do python -c "
import json;d=json.load(open('gpurun_out/v8/bench_b$b.json'));print($b, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('ttfa_ms_p50'))"; done
cat gpurun_out/v8/pmc_*_*.json | head -40
timeout 600 python tools/bench_csm.py --batch 1 > gpurun_out/v8/csm_b1.json 2>&1
timeout 600 python tools/bench_glm.py > gpurun_out/v8/glm_b8.json 2>&1
timeout 600 python tools/bench_glm.py --batch 1 > gpurun_out/v8/glm_b1.json 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/v8/pmc_mfma -o p -- python bench.py --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --ttfa-requests 0 > gpurun_out/v8/pmc_mfma.log 2>&1
python tools/mfma_summary.py $(find gpurun_out/v8/pmc_mfma -name "*counter_collection.csv") $(find gpurun_out/v8/pmc_mfma -name "*kernel_trace.csv") > gpurun_out/v8/mfma_b32.json
rm -rf gpurun_out/v8/pmc_mfma
tail -n 1 gpurun_out/v8/csm.json gpurun_out/v8/csm_b1.json gpurun_out/v8/glm_b8.json gpurun_out/v8/glm_b1.json | cut -c1-400
timeout 600 python bench.py --batch 8 --exact-rows 2 --no-cpu-baseline --ttfa-requests 0 > gpurun_out/v8/bench_b8_fast.json 2> gpurun_out/v8/bench_b8_fast.err
timeout 600 python tools/bench_glm.py --exact-rows 2 > gpurun_out/v8/glm_b8_fast.json 2>&1
tail -n 1 gpurun_out/v8/bench_b8_fast.json gpurun_out/v8/glm_b8_fast.json | cut -c1-300
