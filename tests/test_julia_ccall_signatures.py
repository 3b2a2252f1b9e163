"""Static cross-check of the (unexecuted) Julia extension against the C ABI it binds.

There is no Julia in this image or on the GPU box, so `julia/ext/*.jl` cannot be run.  What CAN be checked
without running it: every `ccall((:dpr_..., libdpr), Ret, (ArgTypes...), args...)` in the Julia sources names
a function `include/dpr.h` declares, with the declared return type, the declared number of arguments and an
argument-type tuple that is ABI-compatible with the prototype, slot by slot -- and passes as many values as
it declares types.  A header change that is not carried into the extension (or the reverse) fails here.

The reference side of these bindings: ext/DiffPointRasterisationCUDAExt.jl:231-333 (the CUDA methods the
AMDGPU extension mirrors); the ABI: include/dpr.h.
"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dpr.h")
JULIA_DIRS = [os.path.join(ROOT, "julia", "ext"), os.path.join(ROOT, "julia", "test")]


# ---------------------------------------------------------------- the header
def _strip_c_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#[^\n]*(\\\n[^\n]*)*", ";", src, flags=re.M)  # preprocessor lines
    return src.replace('extern "C" {', ";")


def _canon_c(t):
    """C parameter / return type -> an ABI class."""
    t = re.sub(r"\bconst\b", " ", t)
    t = re.sub(r"\s+", " ", t).strip()
    t = t.replace(" *", "*").replace("* ", "*")
    table = {
        "int": "i32", "unsigned": "u32", "unsigned int": "u32", "int64_t": "i64", "size_t": "usize",
        "double": "f64", "float": "f32", "void": "void",
        "int64_t*": "ptr:i64", "float*": "ptr:f32", "double*": "ptr:f64", "uint32_t*": "ptr:u32",
        "void*": "ptr:any", "char*": "ptr:char", "void**": "ptr:ptr", "dpr_comm_t*": "ptr:any",
        "dpr_comm_t**": "ptr:ptr",
    }
    assert t in table, f"dpr.h uses a type this test does not know: {t!r}"
    return table[t]


def header_prototypes():
    src = _strip_c_comments(open(HEADER).read())
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(dpr_\w+)\s*\(([^()]*)\)\s*;", src):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        if "typedef" in ret or "#define" in ret:
            continue
        args = []
        if params.strip() not in ("", "void"):
            for p in params.split(","):
                p = p.strip()
                # drop the parameter name (last identifier), keep the type
                tm = re.match(r"(.*?)(\b[A-Za-z_]\w*)$", p, flags=re.S)
                assert tm, p
                args.append(_canon_c(tm.group(1)))
        protos[name] = (_canon_c(ret), args)
    return protos


# ---------------------------------------------------------------- the Julia sources
def _balanced(src, i):
    """src[i] == '(' -> index one past its matching ')' (strings and comments do not occur inside ccalls here)."""
    depth = 0
    for j in range(i, len(src)):
        c = src[j]
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                return j + 1
    raise AssertionError("unbalanced ccall")


def _split_top(s):
    out, depth, cur = [], 0, []
    for c in s:
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
        if c == "," and depth == 0:
            out.append("".join(cur).strip())
            cur = []
        else:
            cur.append(c)
    tail = "".join(cur).strip()
    if tail:
        out.append(tail)
    return out


def _strip_jl_comments(src):
    return re.sub(r"#[^\n]*", "", src)


_JL_TYPES = {
    "Cint": {"i32"}, "Int32": {"i32"}, "Cuint": {"u32"}, "UInt32": {"u32"}, "Int64": {"i64"},
    "Csize_t": {"usize"}, "Cdouble": {"f64"}, "Float64": {"f64"}, "Cfloat": {"f32"}, "Float32": {"f32"},
    "Cvoid": {"void"}, "Cstring": {"ptr:char"},
    "Ptr{Int64}": {"ptr:i64"}, "Ptr{UInt32}": {"ptr:u32"}, "Ptr{Float32}": {"ptr:f32"},
    "Ptr{Float64}": {"ptr:f64"}, "Ptr{Cvoid}": {"ptr:any"}, "Ptr{UInt8}": {"ptr:any"},
    "Ptr{Ptr{Cvoid}}": {"ptr:ptr"},
}


def _jl_type_classes(tok, elem):
    tok = tok.replace(" ", "")
    if tok == "Ptr{T}":
        return {f"ptr:{elem}"} if elem else {"ptr:f32", "ptr:f64"}
    assert tok in _JL_TYPES, f"Julia ccall uses a type this test does not know: {tok!r}"
    return _JL_TYPES[tok]


def _resolve_symbols(src, pos, first):
    """names the first ccall argument `(sym_or_literal, libdpr)` can take"""
    inner = first.strip()
    assert inner.startswith("(") and inner.endswith(")"), first
    sym, lib = [x.strip() for x in _split_top(inner[1:-1])]
    assert lib == "libdpr", f"ccall into {lib!r}, not libdpr"
    if sym.startswith(":"):
        return [sym[1:]]
    # a variable: the nearest assignment above, `sym = cond ? :a : :b` or `sym = :a`
    best = None
    for m in re.finditer(r"\b" + re.escape(sym) + r"\s*=\s*([^\n]*)", src[:pos]):
        best = m
    assert best, f"cannot resolve ccall symbol variable {sym!r}"
    names = re.findall(r":(dpr_\w+)", best.group(1))
    assert names, f"no symbol literal in the assignment of {sym!r}: {best.group(1)!r}"
    return names


def julia_ccalls():
    calls = []
    for d in JULIA_DIRS:
        for fn in sorted(os.listdir(d)):
            if not fn.endswith(".jl"):
                continue
            src = _strip_jl_comments(open(os.path.join(d, fn)).read())
            for m in re.finditer(r"\bccall\s*\(", src):
                end = _balanced(src, m.end() - 1)
                parts = _split_top(src[m.end():end - 1])
                assert len(parts) >= 3, (fn, parts)
                line = src.count("\n", 0, m.start()) + 1
                for name in _resolve_symbols(src, m.start(), parts[0]):
                    calls.append((f"{fn}:{line}", name, parts[1], parts[2], parts[3:]))
    return calls


# ---------------------------------------------------------------- the checks
def test_header_parses_and_holds_the_entry_points():
    protos = header_prototypes()
    for name in ("dpr_raster_f32", "dpr_raster_pullback_f64", "dpr_raster_pullback_ex_f32",
                 "dpr_raster_residual_pullback_f32", "dpr_sort_points_f64", "dpr_comm_init",
                 "dpr_raster_pullback_sharded_f32", "dpr_workspace_bytes_ex_f64", "dpr_last_error"):
        assert name in protos, name
    assert protos["dpr_raster_f32"][0] == "i32" and len(protos["dpr_raster_f32"][1]) == 15
    assert protos["dpr_raster_pullback_f32"][1][6] == "ptr:f32"
    assert protos["dpr_last_error"] == ("ptr:char", [])


def test_every_julia_ccall_matches_its_prototype():
    protos = header_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 25, "the extension's ccalls were not found"
    seen = set()
    for where, name, ret, argtypes, args in calls:
        assert name in protos, f"{where}: {name} is not declared in include/dpr.h"
        seen.add(name)
        c_ret, c_args = protos[name]
        elem = "f32" if name.endswith("_f32") else ("f64" if name.endswith("_f64") else None)
        assert c_ret in _jl_type_classes(ret, elem), f"{where}: {name} returns {c_ret}, ccall says {ret}"
        at = argtypes.strip()
        assert at.startswith("(") and at.endswith(")"), f"{where}: argument types are not a tuple: {at}"
        jl_types = _split_top(at[1:-1])
        assert len(jl_types) == len(c_args), \
            f"{where}: {name} takes {len(c_args)} arguments, the ccall declares {len(jl_types)}"
        for k, (jt, ct) in enumerate(zip(jl_types, c_args)):
            ok = ct in _jl_type_classes(jt, elem)
            # a typed pointer may be handed over as an untyped one (device addresses: Ptr{Cvoid})
            ok = ok or (ct.startswith("ptr:") and ct != "ptr:ptr" and jt.replace(" ", "") == "Ptr{Cvoid}")
            assert ok, f"{where}: {name} argument {k}: header {ct}, ccall {jt}"
        if not (len(args) == 1 and args[0].endswith("...")):
            assert len(args) == len(c_args), \
                f"{where}: {name}: {len(args)} values for {len(c_args)} declared argument types"
    # the extension binds both element types of every operation it offers
    for stem in ("dpr_raster", "dpr_raster_pullback", "dpr_raster_ex", "dpr_raster_pullback_ex",
                 "dpr_raster_residual_pullback", "dpr_sort_points", "dpr_raster_pullback_sharded",
                 "dpr_workspace_bytes", "dpr_workspace_bytes_ex"):
        for suffix in ("_f32", "_f64"):
            assert stem + suffix in seen, f"the Julia extension never calls {stem + suffix}"


def test_splatted_argument_tuples_have_the_declared_length():
    """`ccall(..., (types...), args...)`: the tuple `args` is built a few lines above the call; its length
    must equal the prototype's (a wrong length is a run-time MethodError in Julia, not a build error)."""
    protos = header_prototypes()
    checked = 0
    for d in JULIA_DIRS:
        for fn in sorted(os.listdir(d)):
            if not fn.endswith(".jl"):
                continue
            src = _strip_jl_comments(open(os.path.join(d, fn)).read())
            for m in re.finditer(r"\bccall\s*\(", src):
                end = _balanced(src, m.end() - 1)
                parts = _split_top(src[m.end():end - 1])
                if not (len(parts) == 4 and parts[3].endswith("...")):
                    continue
                var = parts[3][:-3].strip()
                best = None
                for a in re.finditer(r"\b" + re.escape(var) + r"\s*=\s*\(", src[:m.start()]):
                    best = a
                assert best, f"{fn}: cannot find the tuple {var!r}"
                tend = _balanced(src, best.end() - 1)
                n = len(_split_top(src[best.end():tend - 1]))
                for name in _resolve_symbols(src, m.start(), parts[0]):
                    assert n == len(protos[name][1]), f"{fn}: {name}: tuple {var} has {n} values, prototype {len(protos[name][1])}"
                    checked += 1
    assert checked >= 4
