#!/usr/bin/env python3
"""Disassembles the gfx950 code objects of an object file / shared library and prints, per kernel whose name contains
the given substring, the instruction count by mnemonic class (VALU FP64, integer VALU, LDS, VMEM, SALU ...).
usage: tools/disasm.py <file.o|.so> <kernel substring> [--dump out.s]"""
import collections
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, __import__("os").path.dirname(__file__))
from check_dpp_hazards import OBJDUMP, code_objects

path, sub = sys.argv[1], sys.argv[2]
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
dumped = []
for co in code_objects(path):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co)
        f.flush()
        text = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True).stdout
    func, keep = None, False
    counts = {}
    out = []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            func = m.group(1)
            keep = sub in func
            if keep:
                counts[func] = collections.Counter()
            continue
        if not keep:
            continue
        out.append(line)
        m = re.match(r"^\s+(\S+)", line)
        if m:
            counts[func][m.group(1)] += 1
    for fn, c in counts.items():
        tot = sum(c.values())
        print(f"== {fn}: {tot} instructions")
        cls = collections.Counter()
        for k, v in c.items():
            if k.startswith("v_mad_u64") or k.startswith("v_mul_hi") or k.startswith("v_mul_lo"):
                cls["valu int mul"] += v
            elif k.startswith("v_") and "f64" in k:
                cls["valu f64"] += v
            elif k.startswith("v_"):
                cls["valu other"] += v
            elif k.startswith("ds_"):
                cls["lds"] += v
            elif k.startswith(("global_", "buffer_", "flat_")):
                cls["vmem"] += v
            elif k.startswith("s_"):
                cls["salu/ctrl"] += v
            else:
                cls["other"] += v
        print("  ", dict(cls))
        print("  ", c.most_common(28))
    dumped += out
if dump:
    open(dump, "w").write("\n".join(dumped))
