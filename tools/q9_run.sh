cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/q9; mkdir -p $O
for B in 1 8 32; do timeout 200 python $R/tools/lm_timing.py $B 200 2>&1 | tail -1 >> $O/times.txt; done
cat $O/times.txt
cd $R
timeout 900 python -m pytest tests -m gpu -x -q --durations=15 > $O/gpu_suite.log 2>&1; tail -30 $O/gpu_suite.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["ttfa_ms_p50"], d["roofline"]["frac"], d["batch8"]["ms_per_step"], d["batch32"]["ms_per_step"], {k:round(v["value"]/1e6,2) for k,v in d["serving_path_throughput"].items()})
print({k:(v.get("audio_samples_per_s"), v.get("detokenizer_chunk_ms")) for k,v in d["other_configs"].items()})
PY
