"""Where do RCCL's kernels run relative to the interior sweeps?  Reads the rocpd (sqlite) output of `rocprofv3 --kernel-trace` of each rank of a
decomposed run (tools/rccl_overlap_timeline.sh: real RCCL ranks as processes) and reports, per rank:
  * every kernel of the exchange (shell slabs, pack, RCCL send/recv, unpack, all-reduce, test) with the share of its duration that lies
    inside an interior sweep (jacobi2p_k) of the same rank,
  * wall time per pass (distance between interior starts) against the interior kernel's own duration = what the exchange leaves exposed.
usage: python tools/rccl_overlap.py <rank0.db> [<rank1.db> ...]"""
import collections
import sqlite3
import sys


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    return list(cur.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))


def kind(name):
    n = name.lower()
    if "jacobi2p_k" in n:
        return "interior"
    if "nccl" in n or "rccl" in n:
        return "rccl"
    for k in ("pair_shell_k", "box_copy_k", "shell_fold_k", "check2_k", "check_k", "copy_shell_k"):
        if k in n:
            return k
    return None


for path in sys.argv[1:]:
    rows = load(path)
    inter = [(s, e) for n, s, e, q in rows if kind(n) == "interior"]
    if len(inter) < 6:
        print(path, ": too few interior launches")
        continue
    inter = inter[2:]  # the first passes include first-launch effects
    t_lo, t_hi = inter[0][0], inter[-1][1]
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])  # count, duration, inside
    names = {}
    for n, s, e, q in rows:
        k = kind(n)
        if k in (None, "interior") or s < t_lo or e > t_hi:
            continue
        ins = sum(max(0, min(e, ie) - max(s, is_)) for is_, ie in inter)
        a = agg[k]
        a[0] += 1
        a[1] += (e - s) / 1e3
        a[2] += ins / 1e3
        names.setdefault(k, n.split("(")[0][:60])
    dur = sorted((e - s) / 1e3 for s, e in inter)
    gaps = sorted((inter[i + 1][0] - inter[i][0]) / 1e3 for i in range(len(inter) - 1))
    print(f"== {path}")
    print(f"interior jacobi2p_k: {len(inter)} launches, median {dur[len(dur) // 2]:.1f} us; start-to-start median {gaps[len(gaps) // 2]:.1f} us "
          f"-> exposed per pass {gaps[len(gaps) // 2] - dur[len(dur) // 2]:.1f} us")
    print(f"{'kernel':14s} {'calls':>6s} {'mean us':>9s} {'inside an interior sweep':>26s}   name")
    for k, (c, d, ins) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:14s} {c:6d} {d / c:9.1f} {100.0 * ins / d if d else 0.0:25.1f}%   {names[k]}")
    # one pass in detail
    mid = inter[len(inter) // 2]
    print(f"one pass (interior starts at 0, lasts {(mid[1] - mid[0]) / 1e3:.1f} us):")
    for n, s, e, q in rows:
        k = kind(n)
        if k and k != "interior" and s >= mid[0] - 20e3 and s < mid[1] + 60e3:
            print(f"   {(s - mid[0]) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  queue {q}  {k}")
