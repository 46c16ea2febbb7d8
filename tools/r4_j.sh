cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1
tail -6 $O/gpu_suite.log
timeout 600 python tools/depth_persist_check.py 30 2>&1 | grep -v amdgpu.ids | tail -4
( time timeout 1200 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -4 $O/bench_default.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4j/bench_default.json") if l.startswith("{")][-1])
keep = {k: d.get(k) for k in ("value", "ms_per_step", "ttfa_ms_p50", "ttfa_ms_p50_detokenize_interval_2", "ttfa_ms_p50_under_32way_load", "depth_persist")}
keep["roofline_frac"] = d["roofline"]["frac"]; keep["graph_ms"] = d["roofline"]["avg_launch_ms"]
for b in ("batch8", "batch32"):
    keep[b] = {k: d[b][k] for k in ("value", "ms_per_step")}; keep[b]["frac"] = d[b]["roofline"]["frac"]; keep[b]["graph_ms"] = d[b]["roofline"]["avg_launch_ms"]
keep["kv_sweep_b1"] = {k: round(v["frame_ms"], 3) for k, v in d["kv_sweep"]["batch1"].items()}
keep["serving"] = {k: {"value": v.get("value"), "steady": (v.get("steady_state") or {}).get("value")} for k, v in d.get("serving_path_throughput", {}).items()}
keep["pool"] = {k: d["serving_pool_dp"].get(k) for k in ("value", "ttfa_ms_p50_client", "error")}
keep["other"] = {k: {kk: v.get(kk) for kk in ("value", "ms_per_step", "error") if kk in v} for k, v in d.get("other_configs", {}).items()}
print(json.dumps(keep, indent=1))
PY
