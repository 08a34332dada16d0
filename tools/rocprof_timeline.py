"""Time-ordered kernel list (start, duration, stream/queue) of a window of a rocprofv3 --kernel-trace run (rocpd sqlite):
python tools/rocprof_timeline.py <results.db> [first] [count]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.queue_id, d.stream_id, d.tid from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
t0 = rows[0][1]
for n, st, en, q, sid, tid in rows[first:first + count]:
    short = n.split("(")[0].replace("_ZN12_GLOBAL__N_1", "")[:40]
    print(f"{(st - t0) / 1e3:12.1f} us  +{(en - st) / 1e3:8.1f} us  queue {q} stream {sid} tid {tid % 10000:5d}  {short}")
