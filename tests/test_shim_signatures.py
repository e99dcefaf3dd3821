"""The Julia shim cannot be executed here (no Julia toolchain), so its `ccall`s are checked mechanically:

every `ccall((:gss_x, libgss), Int32, (T1, T2, ...), ...)` type tuple in geostatssolvers.jl_amd/julia/GeoStatsSolversHIP.jl
must be the Julia spelling of the parameter list `include/gss.h` declares for `gss_x`, every call must pass exactly
as many arguments as the tuple has types, and the two `struct`s that cross the ABI by value/pointer must list the
fields of `gss_variogram_t` in order with matching types.  CPU only."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gss.h")
SHIM = os.path.join(ROOT, "geostatssolvers.jl_amd", "julia", "GeoStatsSolversHIP.jl")

SCALARS = {"int32_t": "Int32", "int64_t": "Int64", "uint64_t": "UInt64", "double": "Float64", "uint8_t": "UInt8"}


def _strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def _julia_type(ctype):
    """C parameter type (without the name) -> the Julia type a `ccall` tuple must name."""
    t = ctype.replace("const", " ").strip()
    t = re.sub(r"\s+", " ", t)
    stars = t.count("*")
    base = t.replace("*", "").strip()
    if base == "gss_variogram_t":
        assert stars == 1
        return "Ptr{GssVariogram}"
    if re.fullmatch(r"gss_\w+_t", base) or base == "void":      # opaque handles and void* are Ptr{Cvoid}
        assert stars in (1, 2), ctype
        return "Ptr{Cvoid}" if stars == 1 else "Ptr{Ptr{Cvoid}}"
    if base == "char":
        return "Ptr{UInt8}"
    j = SCALARS[base]
    for _ in range(stars):
        j = "Ptr{%s}" % j
    return j


def header_signatures():
    text = _strip_comments(open(HEADER).read())
    sigs = {}
    for m in re.finditer(r"int32_t\s+(gss_\w+)\s*\(([^)]*)\)\s*;", text):
        name, params = m.group(1), m.group(2).strip()
        types = []
        if params and params != "void":
            for p in params.split(","):
                p = p.strip()
                pm = re.fullmatch(r"(.*?)(\w+)(\[\d*\])?", p)          # type, name, optional array suffix
                ctype = pm.group(1).strip() + ("*" if pm.group(3) else "")
                types.append(_julia_type(ctype))
        sigs[name] = types
    return sigs


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def shim_ccalls():
    """[(name, [type, ...], number of arguments passed)] for every ccall into libgss."""
    text = open(SHIM).read()
    text = re.sub(r"#[^\n]*", "", text)                           # comments
    calls = []
    for m in re.finditer(r"ccall\(\(:(gss_\w+), libgss\)", text):
        i = m.end()
        depth, j = 1, i                                           # find the matching ')' of ccall(
        while depth:
            ch = text[j]
            depth += ch in "([{"
            depth -= ch in ")]}"
            j += 1
        args = _split_top(text[i:j - 1].lstrip(","))              # [rettype, (types...), a1, a2, ...]
        assert args[0] == "Int32", (m.group(1), args[0])
        tup = args[1].strip()
        assert tup.startswith("(") and tup.endswith(")"), tup
        types = [t for t in _split_top(tup[1:-1]) if t]
        calls.append((m.group(1), types, len(args) - 2))
    return calls


def test_every_ccall_tuple_matches_the_header():
    sigs = header_signatures()
    calls = shim_ccalls()
    assert len(calls) >= 18
    for name, types, nargs in calls:
        assert name in sigs, f"{name} is not declared in include/gss.h"
        assert types == sigs[name], f"{name}: shim passes {types}, header declares {sigs[name]}"
        assert nargs == len(types), f"{name}: {nargs} arguments for {len(types)} types"


def test_header_parser_sees_every_export():
    """Guards the parser itself: the same declarations the ctypes table binds (tests/test_abi.py keeps that table in
    step with the built library)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "geostatssolvers.jl_amd"))
    from gss import _lib
    sigs = header_signatures()
    assert set(sigs) | {"gss_version"} >= set(_lib.SIGNATURES), set(_lib.SIGNATURES) - set(sigs)
    for name, args in _lib.SIGNATURES.items():
        if name in sigs:
            assert len(args) == len(sigs[name]), name


def _c_struct_fields(body):
    fields = []
    body = re.sub(r"struct\s*\{(.*?)\}\s*(\w+)\[(\d+)\]\s*;", lambda m: f"@INNER@ {m.group(2)}[{m.group(3)}];", body, flags=re.S)
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.fullmatch(r"(@INNER@|\w+)\s+(\w+)(?:\[(\d+)\])?", decl)
        assert m, decl
        ctype, name, arr = m.groups()
        j = "GssVgExtra" if ctype == "@INNER@" else SCALARS[ctype]
        fields.append((name, f"NTuple{{{arr},{j}}}" if arr else j))
    return fields


def _julia_struct_fields(name):
    text = open(SHIM).read()
    m = re.search(r"struct %s\b(.*?)\nend" % name, text, flags=re.S)
    fields = []
    for line in m.group(1).splitlines():
        line = re.sub(r"#.*", "", line).strip()
        if "::" in line:
            f, t = line.split("::")
            fields.append((f.strip(), t.strip()))
    return fields


def test_variogram_structs_have_the_header_layout():
    text = _strip_comments(open(HEADER).read())
    m = re.search(r"typedef struct gss_variogram \{(.*)\} gss_variogram_t;", text, flags=re.S)
    body = m.group(1)
    inner = re.search(r"struct\s*\{(.*?)\}\s*extra\[3\]", body, flags=re.S).group(1)
    c_outer, c_inner = _c_struct_fields(body), _c_struct_fields(inner)
    j_outer, j_inner = _julia_struct_fields("GssVariogram"), _julia_struct_fields("GssVgExtra")
    assert [t for _, t in c_outer] == [t for _, t in j_outer], (c_outer, j_outer)
    assert [n for n, _ in c_outer] == [n for n, _ in j_outer]
    assert c_inner == j_inner
    # and the ctypes mirror the twin uses
    import sys
    sys.path.insert(0, os.path.join(ROOT, "geostatssolvers.jl_amd"))
    from gss import _lib
    assert [f[0] for f in _lib.Variogram._fields_] == [n for n, _ in c_outer]
