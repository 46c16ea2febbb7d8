"""Summarise a rocprofv3 kernel trace: per (kernel, grid) count / avg / total for the last N frames, plus idle gaps."""
import csv, sys
from collections import defaultdict
path, nlast = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6000
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-nlast:]
d = defaultdict(list)
gap = 0
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
    d[(name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Grid_Size_Y"], r["Grid_Size_Z"])].append(e - s)
    if prev is not None and 0 < s - prev < 50000:
        gap += s - prev
    prev = e
tot = sum(sum(v) for v in d.values())
print(f"kernels {len(rows)}  busy {tot/1e6:.3f} ms  small-gaps {gap/1e6:.3f} ms")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:28]:
    print(f"{k[0]:36s} grid {k[1]:5d}x{k[2]}x{k[3]:3s} n={len(v):5d} avg {sum(v)/len(v)/1e3:7.2f} us  tot {sum(v)/1e6:7.3f} ms  {100*sum(v)/tot:5.1f}%")
