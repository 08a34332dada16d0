#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes (one counter per pass, as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes)
into HBM bytes per launch of the dominant kernel.

    FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction: FETCH_SIZE reports exactly half of the bytes of a wide
    (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.

usage: summarize_pmc.py <key> <kernel substring> <fetch_dir> <write_dir> [out.json]
       summarize_pmc.py --all[=name1;name2;...] <key> <units> <fetch_dir> <write_dir> [out.json]
           every kernel of the run by (shortened) name: launches per unit (units = e.g. the BiCGSTAB iterations of the profiled command),
           bytes per launch, and the bytes one unit moves in all -- the physical traffic of an iteration"""
import csv
import glob
import hashlib
import json
import os
import socket
import sys
import time

ALL = sys.argv[1].startswith("--all")
ONLY = None
if ALL:
    a = sys.argv.pop(1)
    if "=" in a:  # kernels of the unit by name prefix; what the set-up of the profiled command launched (fills, copies, boundary faces) is left out
        ONLY = a.split("=", 1)[1].split(";")  # (";": kernel names contain commas)
key, kname, fdir, wdir = sys.argv[1:5]
out = sys.argv[5] if len(sys.argv) > 5 else None


def short_name(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):  # up to the argument list: the "(" at template depth 0
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return n[:cut].strip()


def per_kernel(d, counter):
    acc = {}
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc.setdefault(short_name(r["Kernel_Name"]), [0.0, 0])
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return acc


def build_id():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("cz_k_common.h", "cz_k_fastdiv.h", "cz_k_stencil.h", "cz_k_pair.h", "cz_k_pair2.h", "cz_k_rb4.h", "cz_k_blas.h", "cz_h_launch.h"):
        h.update(open(os.path.join(root, "cubez_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


if ALL:
    units = float(kname)
    fe, wr = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    kernels, total = {}, 0.0
    for n in sorted(set(fe) | set(wr)):
        fs, fn = fe.get(n, [0.0, 0])
        ws, wn = wr.get(n, [0.0, 0])
        cnt = max(fn, wn)
        rb, wb = (fs / fn * 2048.0 if fn else 0.0), (ws / wn * 1024.0 if wn else 0.0)
        if (rb + wb) * cnt / units < 1e6:  # one-thread launches: not a line of the table
            continue
        if ONLY is not None and not any(n.startswith(o) for o in ONLY):
            continue
        kernels[n] = {"launches_per_unit": cnt / units, "read_bytes": rb, "write_bytes": wb, "bytes_per_launch": rb + wb}
        total += (rb + wb) * cnt / units
    rec = {"units": units, "kernels": kernels, "bytes_per_unit": total, "kernel_source_sha": build_id(), "box": socket.gethostname(),
           "recorded": time.strftime("%Y-%m-%d %H:%M:%S"),
           "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads count 64 B per 128-B request), WRITE_SIZE x1; KiB -> bytes"}
    print(json.dumps({key: rec}, indent=1))
    if out:
        try:
            allrec = json.load(open(out))
        except Exception:
            allrec = {}
        allrec[key] = rec
        json.dump(allrec, open(out, "w"), indent=1)
    sys.exit(0)


def mean_counter(d, counter):
    vals = []
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kname in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


fetch_kib, nf = mean_counter(fdir, "FETCH_SIZE")
write_kib, nw = mean_counter(wdir, "WRITE_SIZE")
rec = {"kernel": kname, "launches": [nf, nw], "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
       "read_bytes": fetch_kib * 1024 * 2, "write_bytes": write_kib * 1024,
       "bytes_per_launch": fetch_kib * 1024 * 2 + write_kib * 1024,
       "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads count 64 B per 128-B request), WRITE_SIZE x1"}
# which build and box the figure belongs to (bench.py prints it beside `traffic` and says whether it is the build it is running)
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha256()
for _f in ("cz_k_common.h", "cz_k_fastdiv.h", "cz_k_stencil.h", "cz_k_pair.h", "cz_k_pair2.h", "cz_k_rb4.h", "cz_k_blas.h", "cz_h_launch.h"):
    _h.update(open(os.path.join(_root, "cubez_amd", "csrc", _f), "rb").read())
rec["kernel_source_sha"] = _h.hexdigest()[:16]
rec["box"] = socket.gethostname()
rec["recorded"] = time.strftime("%Y-%m-%d %H:%M:%S")
print(json.dumps({key: rec}, indent=1))
if out:
    try:
        allrec = json.load(open(out))
    except Exception:
        allrec = {}
    allrec[key] = rec
    json.dump(allrec, open(out, "w"), indent=1)
