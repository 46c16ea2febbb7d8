"""rocprofv3 kernel trace -> per kernel: median duration and median start-to-next-start pitch over the last N kernels."""
import csv, sys
from collections import defaultdict
import statistics as st
path, nlast = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-nlast:]
dur, pitch, gap = defaultdict(list), defaultdict(list), defaultdict(list)
for a, b in zip(rows, rows[1:]):
    name = a["Kernel_Name"].split("(")[0].replace("void ", "")[:40] + f" g{int(a['Grid_Size_X']) // max(1, int(a['Workgroup_Size_X']))}"
    s, e, s2 = int(a["Start_Timestamp"]), int(a["End_Timestamp"]), int(b["Start_Timestamp"])
    if s2 - s < 40000:
        dur[name].append(e - s); pitch[name].append(s2 - s); gap[name].append(s2 - e)
tot = sum(sum(v) for v in pitch.values())
print(f"kernels {len(rows)}, pitch total {tot/1e6:.3f} ms")
for k, v in sorted(pitch.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print(f"{k:48s} n={len(v):5d} dur {st.median(dur[k])/1e3:6.2f}  gap {st.median(gap[k])/1e3:6.2f}  pitch {st.median(v)/1e3:6.2f} us  tot {sum(v)/1e6:7.3f} ms")
