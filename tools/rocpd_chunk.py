"""Per-dispatch durations of the last codec chunk in a rocprofv3 results database (rocpd sqlite)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'rocpd_kernel_dispatch_' in t][0]
sym = [t for t in tabs if 'rocpd_info_kernel_symbol_' in t][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from {kd} d join {sym} s on d.kernel_id=s.id order by d.start"))
idx = [i for i, r in enumerate(rows) if 'k_rvq' in r[0]]
last = rows[idx[-1]:]
t0, tot = last[0][1], 0.0
for r in last:
    nm = r[0].replace('_Z11k_conv_gemmI', 'cg<').replace('EEv12ConvGemmArgs.kd', '>').replace('Li', '').replace('E', ',')
    d = (r[2] - r[1]) / 1e3
    if d > thr:
        print(f"{(r[1]-t0)/1e3:9.1f} us  {nm[:50]:50s} grid {r[3]//r[5]}x{r[4]}  {d:8.1f} us")
    tot += d
print('sum of kernels', round(tot, 1), 'us; span', round((last[-1][2] - t0) / 1e3, 1), 'us;', len(last), 'launches')
