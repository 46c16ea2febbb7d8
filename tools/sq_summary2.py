"""Per-kernel SQ counter ratios from a rocprofv3 --pmc pass (counter_collection.csv): where do a kernel's wave-cycles go?
usage: sq_summary2.py <counter_collection.csv> [name filter ...]"""
import csv, sys
from collections import defaultdict
acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(int)
filt = sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    if filt and not any(f in k for f in filt):
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        cnt[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:24]:
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    n = max(cnt[k], 1)
    line = " ".join(f"{name[3:]}={v / wc:.2f}" for name, v in sorted(c.items()) if name not in ("SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_BUSY_CYCLES") and "INSTS" not in name)
    insts = " ".join(f"{name[3:]}/wave={v / max(c.get('SQ_WAVES', 1), 1):.0f}" for name, v in sorted(c.items()) if "INSTS" in name)
    print(f"{k:44s} n={n:5d} wave_cycles(quad)/dispatch={wc / n:9.0f} waves/dispatch={c.get('SQ_WAVES', 0) / n:6.0f} | {line} | {insts}")
