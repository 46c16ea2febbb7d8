"""MFMA utilisation per kernel from a rocprofv3 `--pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace` pass.
usage: mfma_summary.py <counter_collection.csv> <kernel_trace.csv>
SQ_VALU_MFMA_BUSY_CYCLES = cycles the MFMA pipes were busy, summed over every SIMD (checked here: the talker gate/up GEMM at
32 rows issues 384 blocks x 8 waves x 32 v_mfma_f32_16x16x32_bf16 = 98 304 MFMAs and reads 1.573 M = 16.0 cycles each, the
instruction's issue cost in MI355X_MICROARCH.md 'Per-instruction cycle constants').  Utilisation = busy cycles /
(kernel duration x 2.4 GHz x 1024 SIMDs): the fraction of the dense bf16 MFMA peak (2.5 PFLOP/s) the kernel sustains."""
import csv, json, sys
from collections import defaultdict
pmc, trace = sys.argv[1], sys.argv[2]
dur = {}
for r in csv.DictReader(open(trace)):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy, t_ns, n = defaultdict(float), defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(pmc)):
    if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
        continue
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    busy[k] += float(r["Counter_Value"])
    t_ns[k] += dur.get(r["Dispatch_Id"], 0)
    n[k] += 1
rows = sorted(((b, k) for k, b in busy.items() if b > 0 and t_ns[k] > 0), reverse=True)
out = [{"kernel": k, "dispatches": n[k], "avg_us": round(t_ns[k] / n[k] / 1e3, 2), "mfma_busy_cycles_per_dispatch": round(b / n[k]),
        "mfma_util_pct_of_bf16_peak": round(100.0 * b / (t_ns[k] * 2.4 * 1024), 2)} for b, k in rows[:24]]
print(json.dumps(out, indent=1))
