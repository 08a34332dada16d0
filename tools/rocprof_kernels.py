"""Per-kernel duration table from a rocprofv3 --kernel-trace run that wrote the default rocpd (sqlite) output:
python tools/rocprof_kernels.py <results.db>."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"
agg = collections.defaultdict(list)
for n, st, en, gx, gy in cur.execute(q):
    agg[(n[:70], gx, gy)].append((en - st) / 1e3)
print(f"{'kernel':72s} {'grid':>16s} {'calls':>6s} {'median_us':>10s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s}")
for (n, gx, gy), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{n:72s} {str(gx) + 'x' + str(gy):>16s} {len(v):6d} {v2[len(v2) // 2]:10.1f} {v2[0]:9.1f} {v2[-1]:9.1f} {sum(v) / 1e3:9.2f}")
