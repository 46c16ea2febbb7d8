# round 4: full GPU suite on the current library + the default bench line (timed as the driver runs it)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1
tail -25 $O/gpu_suite.log
cat gpurun_out/parity_counts.json 2>/dev/null
( time timeout 1200 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -30 $O/bench_default.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4d/bench_default.json") if l.startswith("{")][-1])
keep = {k: d.get(k) for k in ("value", "ms_per_step", "ttfa_ms_p50", "ttfa_ms_p50_detokenize_interval_2", "ttfa_ms_p50_under_32way_load", "serving_pool_dp", "cpu_baseline")}
keep["roofline_frac"] = d["roofline"]["frac"]; keep["graph_ms"] = d["roofline"]["avg_launch_ms"]
for b in ("batch8", "batch32"):
    keep[b] = {k: d[b][k] for k in ("value", "ms_per_step")}; keep[b]["frac"] = d[b]["roofline"]["frac"]; keep[b]["graph_ms"] = d[b]["roofline"]["avg_launch_ms"]
keep["serving"] = {k: {"value": v.get("value"), "steady": (v.get("steady_state") or {}).get("value")} for k, v in d.get("serving_path_throughput", {}).items()}
keep["other"] = {k: {kk: v.get(kk) for kk in ("value", "ms_per_step", "error") if kk in v} for k, v in d.get("other_configs", {}).items()}
print(json.dumps(keep, indent=1))
PY
