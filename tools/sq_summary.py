import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    if "conv_gemm" not in k and "fullk<2, 8, 1, 2" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print(k, {n: round(v / wc, 3) for n, v in c.items()})
