#!/usr/bin/env python3
"""Static check of the built library for the one hazard the compiler cannot see: the v_fmac_f64_dpp instructions of
csrc/tile16.h are inline assembly, so the hazard recogniser does not know that they read their first source through
DPP.  gfx9 rule: a VGPR written by a VALU instruction may be read through DPP only two wait states later (an s_nop N
gives N + 1).  The script extracts every gfx950 code object from libgss_hip.so, disassembles it and reports any DPP
read whose source register was written by a VALU instruction fewer than two wait states earlier.

    python tools/check_dpp_hazards.py [path/to/libgss_hip.so]      exit code 1 if a hazard is found
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def code_objects(path):
    blob = open(path, "rb").read()
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def regs(op):
    """'v[2:3]' -> {2, 3}; 'v7' -> {7}; anything else -> empty."""
    op = op.strip().lstrip("-|").rstrip("|")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", op)
    return {int(m.group(1))} if m else set()


DPP_MARK = re.compile(r"row_newbcast|quad_perm|row_shl|row_shr|row_ror|row_mirror|row_half_mirror|row_bcast|wave_sh|wave_ro")


def check(text):
    hazards, ndpp = [], 0
    func = "?"
    window = []          # (wait states this instruction provides, set of VGPRs it writes as a VALU op, text)
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            func, window = m.group(1), []
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if not m:
            continue
        mn, ops = m.group(1), m.group(2)
        operands = [o for o in re.split(r",\s*", ops.split(" row_")[0].split(" quad_perm")[0]) if o]
        if DPP_MARK.search(ops):
            ndpp += 1
            src = regs(operands[1]) if len(operands) > 1 else set()
            ws = 0
            for w, written, txt in reversed(window):
                if ws >= 2:
                    break
                if written & src:
                    hazards.append((func, txt.strip(), line.strip()))
                    break
                ws += w
        written = set()
        if mn.startswith("v_") and not mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and operands:
            written = regs(operands[0])
        w = 1
        if mn == "s_nop":
            w = int(ops.split()[0], 0) + 1
        window.append((w, written, line))
        window = window[-4:]
    return hazards, ndpp


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "geostatssolvers.jl_amd", "lib", "libgss_hip.so")
    total, bad = 0, []
    for k, obj in enumerate(code_objects(lib)):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(obj)
            f.flush()
            text = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True, check=True).stdout
        hz, n = check(text)
        total += n
        bad += hz
    print("%d DPP instructions checked, %d hazards" % (total, len(bad)))
    for func, w, r in bad[:20]:
        print("  in %s:\n    write: %s\n    read:  %s" % (func[:100], w, r))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
