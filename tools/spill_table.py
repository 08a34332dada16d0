#!/usr/bin/env python3
"""Register use of every kernel the two libraries ship, from the code objects' metadata:
    python3 tools/spill_table.py [--all]
prints the kernels with a non-zero .vgpr_spill_count / .sgpr_spill_count (or all of them) with VGPRs, scratch bytes and LDS."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
show_all = "--all" in sys.argv
for prec in ("f32", "f64"):
    so = os.path.join(ROOT, "cubez_amd", f"libczhip_{prec}.so")
    with tempfile.TemporaryDirectory() as td:
        tmp_so = os.path.join(td, os.path.basename(so))
        os.symlink(so, tmp_so)  # llvm-objdump --offloading writes the bundles next to its input
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp_so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=td)
        rows = []
        for f in sorted(os.listdir(td)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(td, f)], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                def field(name, blk=blk):
                    m = re.search(r"\.%s:\s*(\S+)" % name, blk)
                    return m.group(1) if m else "?"
                name = field("name")
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = dem.replace("(anonymous namespace)::", "").replace("void ", "")
                dem = dem[:dem.index("(")] if "(" in dem and not dem.startswith("(") else dem
                rows.append((dem, int(field("vgpr_count")), int(field("vgpr_spill_count")), int(field("sgpr_spill_count")),
                             int(field("private_segment_fixed_size")), int(field("group_segment_fixed_size"))))
    bad = [r for r in rows if r[2]]  # (SGPR spills go to VGPR lanes: no memory traffic)
    print(f"== libczhip_{prec}.so: {len(rows)} kernels, {len(bad)} with VGPR spills (scratch memory)")
    for r in sorted(rows if show_all else bad, key=lambda r: (-r[2], r[0])):
        print(f"  vgpr {r[1]:3d}  vgpr_spill {r[2]:3d}  sgpr_spill {r[3]:3d}  scratch {r[4]:5d} B  lds {r[5]:6d} B  {r[0]}")
