"""Copy the summaries written by tools/refresh_profiles_r6.sh (gpurun_out/p6) into profiles/round6_* (the tracked copies)."""
import json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, D = os.path.join(R, "gpurun_out", "p6"), os.path.join(R, "profiles")
first = lambda p: json.loads(open(p).read().split("\n", 1)[0])
traffic = {}
for b in (1, 8, 32):
    shutil.copy(os.path.join(S, f"kernel_stats_b{b}.csv"), os.path.join(D, f"round6_bench_b{b}_kernel_stats.csv"))
    shutil.copy(os.path.join(S, f"trace_summary_b{b}.txt"), os.path.join(D, f"round6_bench_b{b}_trace_summary.txt"))
    shutil.copy(os.path.join(S, f"bench_b{b}_prof.json"), os.path.join(D, f"round6_bench_b{b}_profiled_run.json"))
    f, w = first(os.path.join(S, f"pmc_{b}_FETCH_SIZE.json")), first(os.path.join(S, f"pmc_{b}_WRITE_SIZE.json"))
    traffic[f"batch_{b}"] = {
        "frames_fetch_pass": f["frames"], "fetch_raw_bytes_per_launch": f["lm_raw_bytes_per_frame"],
        "fetch_corrected_bytes_per_launch": f["lm_corrected_bytes_per_frame"], "write_raw_bytes_per_launch": w["lm_raw_bytes_per_frame"],
        "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py --batch B; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                "(gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncalibrated"}
    shutil.copy(os.path.join(S, f"pmc_{b}_FETCH_SIZE.json"), os.path.join(D, f"round6_pmc_fetch_per_kernel_b{b}.txt"))
json.dump(traffic, open(os.path.join(D, "round6_pmc_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(S, "mfma_b32.json"), os.path.join(D, "round6_mfma_util_b32.json"))
shutil.copy(os.path.join(S, "bench_default.json"), os.path.join(D, "round6_bench_default.json"))
shutil.copy(os.path.join(S, "bench_default.err"), os.path.join(D, "round6_bench_default_phases.txt"))
for name in ("csm_b1", "csm_b16", "glm_b1", "glm_b8", "cosyvoice2_b1", "cosyvoice2_b8"):
    shutil.copy(os.path.join(S, name + ".json"), os.path.join(D, f"round6_{name}.json"))
shutil.copy(os.path.join(S, "kernel_stats_csm_b16.csv"), os.path.join(D, "round6_csm_b16_kernel_stats.csv"))
shutil.copy(os.path.join(S, "kernel_stats_glm_b8.csv"), os.path.join(D, "round6_glm_b8_kernel_stats.csv"))
shutil.copy(os.path.join(S, "kernel_stats_cosyvoice2_b1.csv"), os.path.join(D, "round6_cosyvoice2_b1_kernel_stats.csv"))
if os.path.exists(os.path.join(S, "kernel_stats_cosyvoice2_b8.csv")):
    shutil.copy(os.path.join(S, "kernel_stats_cosyvoice2_b8.csv"), os.path.join(D, "round6_cosyvoice2_b8_kernel_stats.csv"))
for src, dst in (("clone.json", "round6_voice_clone_prompt_side.json"), ("kernel_stats_clone.csv", "round6_voice_clone_kernel_stats.csv")):
    if os.path.exists(os.path.join(S, src)):
        shutil.copy(os.path.join(S, src), os.path.join(D, dst))
other = {}
for tag in ("csm16", "glm8"):
    try:
        f, w = first(os.path.join(S, f"pmc_{tag}_FETCH_SIZE.json")), first(os.path.join(S, f"pmc_{tag}_WRITE_SIZE.json"))
        other[tag] = {"frames_fetch_pass": f["frames"], "fetch_corrected_bytes_per_launch": f["lm_corrected_bytes_per_frame"],
                      "write_raw_bytes_per_launch": w["lm_raw_bytes_per_frame"],
                      "traffic_bytes_per_launch": f["lm_corrected_bytes_per_frame"] + w["lm_raw_bytes_per_frame"]}
        shutil.copy(os.path.join(S, f"pmc_{tag}_FETCH_SIZE.json"), os.path.join(D, f"round6_pmc_fetch_per_kernel_{tag}.txt"))
    except Exception as ex:
        other[tag] = {"error": repr(ex)}
json.dump(other, open(os.path.join(D, "round6_pmc_traffic_other_configs.json"), "w"), indent=1)
print({k: round(v["fetch_corrected_bytes_per_launch"] / 1e9, 3) for k, v in traffic.items()}, other)
if os.path.exists(os.path.join(S, "gpu_suite.log")):
    shutil.copy(os.path.join(S, "gpu_suite.log"), os.path.join(D, "round6_gpu_suite.log"))
