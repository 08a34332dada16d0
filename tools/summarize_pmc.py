#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes (one counter per pass, as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes)
into HBM bytes per launch of the dominant kernel.

    FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction: FETCH_SIZE reports exactly half of the bytes of a wide
    (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.

usage: summarize_pmc.py <key> <kernel substring> <fetch_dir> <write_dir> [out.json]"""
import csv
import glob
import hashlib
import json
import os
import socket
import sys
import time

key, kname, fdir, wdir = sys.argv[1:5]
out = sys.argv[5] if len(sys.argv) > 5 else None


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
for _f in ("cz_k_common.h", "cz_k_fastdiv.h", "cz_k_stencil.h", "cz_k_pair.h", "cz_k_pair2.h", "cz_h_launch.h"):
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
