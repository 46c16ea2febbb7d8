# round 4 closing run: GPU suite + smoke + default bench on the final library
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_suite.log 2>&1
tail -4 $O/gpu_suite.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 1200 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4final/bench_default.json") if l.startswith("{")][-1])
print("lines on stdout:", sum(1 for l in open("gpurun_out/r4final/bench_default.json")))
print(round(d["value"]), round(d["ms_per_step"], 4), round(d["roofline"]["frac"], 4), round(d["roofline"]["avg_launch_ms"], 4), d["roofline"]["traffic_source"], d["depth_persist"])
for b in ("batch8", "batch32"):
    print(b, round(d[b]["value"]), round(d[b]["ms_per_step"], 4), round(d[b]["roofline"]["frac"], 4), round(d[b]["roofline"]["avg_launch_ms"], 4))
print({k: round(v["frame_ms"], 3) for k, v in d["kv_sweep"]["batch1"].items()}, {k: round(v["frame_ms"], 3) for k, v in d["kv_sweep"]["batch32"].items()})
print({k: (round(v["value"] / 1e6, 2), round((v.get("steady_state") or {}).get("value", 0) / 1e6, 2)) for k, v in d["serving_path_throughput"].items()})
print(d["ttfa_ms_p50"], d["ttfa_ms_p50_detokenize_interval_2"], d.get("ttfa_ms_p50_under_32way_load"), d["serving_pool_dp"]["value"], d["cpu_baseline"]["value"])
print({k: (v.get("audio_samples_per_s") or v.get("value"), v.get("ms_per_step"), v.get("error")) for k, v in d["other_configs"].items()})
PY
