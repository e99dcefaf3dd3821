#!/usr/bin/env python3
"""VGPR / SGPR / LDS / scratch of the gfx950 kernels in an object file or shared library (code-object metadata).
usage: tools/kernel_resources.py <file.o|.so> [kernel substring]"""
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, __import__("os").path.dirname(__file__))
from check_dpp_hazards import code_objects

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for co in code_objects(path):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co)
        f.flush()
        out = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    cur = {}
    rows = []
    for line in out.splitlines():
        m = re.search(r"\.(name|vgpr_count|sgpr_count|group_segment_fixed_size|vgpr_spill_count|private_segment_fixed_size|"
                      r"max_flat_workgroup_size):\s+(\S+)", line)
        if not m:
            continue
        if m.group(1) == "name":
            if cur.get("name") and "vgpr_count" in cur:
                rows.append(cur)
            cur = {"name": m.group(2)} if not m.group(2).endswith(".kd") else cur
        else:
            cur[m.group(1)] = m.group(2)
    if cur.get("name") and "vgpr_count" in cur:
        rows.append(cur)
    for r in rows:
        if sub in r["name"]:
            v = int(r["vgpr_count"])
            alloc = (v + 7) // 8 * 8
            print(f"{r['name'][:70]:70s} vgpr {v:3d} (waves/SIMD {min(8, 512 // max(alloc, 1))}) sgpr {r.get('sgpr_count')} "
                  f"lds {r.get('group_segment_fixed_size')} scratch {r.get('private_segment_fixed_size')} "
                  f"spills {r.get('vgpr_spill_count')}")
