#!/usr/bin/env python3
"""Static instruction mix of one kernel of a hipcc -S listing, split at its workgroup barriers (s_barrier) or at
`; MARK name` comments: VALU / MFMA / transcendental-class FP64 / LDS / VMEM / SALU counts per region.  The
moving-neighbourhood kernels are straight-line code, so the static count of a region is what a wave executes there.
usage: tools/isa_regions.py <file.s> <kernel symbol substring> [--marks]"""
import re
import sys
from collections import Counter, OrderedDict

path, sub = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l and l.rstrip().split(":")[0].endswith("Ph") is not None and re.match(r"^_Z\S*:", l) and sub in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
# the kernel may have several s_endpgm (early exits): take the .Lfunc_end
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
def cls(m):
    if m.startswith("v_mfma"): return "mfma"
    if m.startswith("ds_"): return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if m.startswith("v_"):
        if "_f64" in m or "f64" in m: return "valu_f64"
        return "valu_other"
    if m.startswith("s_waitcnt"): return "waitcnt"
    if m.startswith("s_barrier"): return "barrier"
    if m.startswith("s_"): return "salu"
    return "other"
regions = OrderedDict()
name = "r0"
regions[name] = Counter()
nb = 0
for l in lines[start + 1:end]:
    t = l.strip()
    m = re.match(r"; MARK (\S+)", t)
    if m:
        name = m.group(1)
        regions.setdefault(name, Counter())
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    mn = t.split()[0]
    c = cls(mn)
    if c == "barrier" and "--marks" not in sys.argv:
        nb += 1
        regions[name]["barrier"] += 1
        name = "r%d" % nb
        regions[name] = Counter()
        continue
    regions[name][c] += 1
cols = ["valu_f64", "valu_other", "mfma", "lds", "vmem", "salu", "waitcnt", "barrier", "other"]
print("%-28s" % "region" + "".join("%11s" % c for c in cols) + "%9s" % "total")
tot = Counter()
for n, c in regions.items():
    print("%-28s" % n + "".join("%11d" % c[k] for k in cols) + "%9d" % sum(c.values()))
    tot.update(c)
print("%-28s" % "TOTAL" + "".join("%11d" % tot[k] for k in cols) + "%9d" % sum(tot.values()))
