"""julia/GPSLCHip.jl is the reference-side binding a CausalGPSLC.jl maintainer would add.  Julia is not available in
this pipeline, so the file cannot be executed; what CAN be checked mechanically is checked here: every `ccall` in it is
parsed and its return type, arity and argument type order are compared with the C prototypes of include/gpslc_hip.h
(and those with the ctypes table the GPU tests call through), the two C structs it mirrors have the header's field
order and types, every header symbol is bound, the helper and method names the shim relies on are defined, and the
block structure (function / if / for / ... / end) balances."""
import ctypes as C
import os
import re

from causalgpslc_jl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "julia", "GPSLCHip.jl")

# canonical type classes
C_TYPES = {
    "gpslc_ctx**": "ctxpp", "gpslc_ctx*": "ctx", "int": "int", "int64_t": "i64", "int32_t": "i32",
    "uint32_t": "u32", "uint64_t": "u64", "double": "f64", "double*": "pf64", "int32_t*": "pi32",
    "int64_t*": "pi64", "char*": "cstr", "gpslc_node*": "pnode", "gpslc_pack_header*": "phdr", "void": "void",
}
JL_TYPES = {
    "Ref{Ptr{Cvoid}}": "ctxpp", "Ptr{Cvoid}": "ctx", "Cint": "int", "Int64": "i64", "Int32": "i32", "UInt32": "u32",
    "UInt64": "u64", "Float64": "f64", "Ptr{Float64}": "pf64", "Ref{Float64}": "pf64", "Ptr{Int32}": "pi32",
    "Ref{Int64}": "pi64", "Ptr{Int64}": "pi64", "Cstring": "cstr", "Ptr{GPSLCNode}": "pnode",
    "Ref{GPSLCNode}": "pnode", "Ref{PackHeader}": "phdr", "Ptr{PackHeader}": "phdr",
}


def _ctype_class(t):
    if t is C.c_int:                    # ctypes aliases c_int32 to c_int on this ABI
        return {"int", "i32"}
    return {C.c_int64: {"i64"}, C.c_int32: {"i32"}, C.c_uint32: {"u32"}, C.c_uint64: {"u64"}, C.c_double: {"f64"},
            C.c_char_p: {"cstr"}, C.c_void_p: {"ctx", "pf64", "pnode", "phdr"}}.get(t) or (
        {"ctxpp"} if t == C.POINTER(C.c_void_p) else {"pf64"} if t == _lib.c_double_p else
        {"pi32"} if t == _lib.c_int32_p else {"pi64"} if t == _lib.c_int64_p else {"?"})


def header_prototypes():
    txt = open(_lib.HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"((?:const\s+)?\w+\s*\**)\s*\b(gpslc_\w+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3)

        def canon(decl):
            decl = decl.replace("const", " ").strip()
            stars = decl.count("*")
            base = decl.replace("*", " ").split()[0]
            return C_TYPES[base + "*" * stars]
        argl = [] if args.strip() in ("", "void") else [canon(a) for a in args.split(",")]
        protos[name] = (canon(ret), argl)
    return protos


def julia_source():
    return open(JL).read()


def julia_ccalls():
    src = julia_source()
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*lib\),\s*(\w+),\s*\(([^()]*)\)", src, flags=re.S):
        name, ret, tup = m.group(1), m.group(2), m.group(3)
        types = [t.strip() for t in re.split(r",(?![^{]*\})", tup) if t.strip()]
        calls.append((name, JL_TYPES[ret], [JL_TYPES[t] for t in types]))
    return calls


def test_every_ccall_matches_the_header_prototype():
    protos = header_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= len(protos)
    for name, ret, args in calls:
        assert name in protos, f"{name}: not declared in include/gpslc_hip.h"
        pret, pargs = protos[name]
        assert ret == pret, (name, ret, pret)
        assert args == pargs, f"{name}: ccall types {args} != header {pargs}"


def test_every_header_symbol_is_bound_by_the_julia_file():
    bound = {c[0] for c in julia_ccalls()}
    assert bound == set(header_prototypes()), set(header_prototypes()) ^ bound
    assert bound == set(_lib.header_symbols())


def test_ctypes_table_matches_the_header_prototype():
    """The binding the GPU tests call through (_lib.SIGNATURES) and the header agree in arity and type order too, so
    the Julia tuples are checked against what is actually exercised on hardware."""
    protos = header_prototypes()
    assert set(protos) == set(_lib.SIGNATURES)
    for name, (res, args) in _lib.SIGNATURES.items():
        pret, pargs = protos[name]
        assert pret in _ctype_class(res), (name, res, pret)
        assert len(args) == len(pargs), name
        for i, (a, p) in enumerate(zip(args, pargs)):
            assert p in _ctype_class(a), (name, i, a, p)


def _c_struct_fields(name):
    txt = open(_lib.HEADER_PATH).read()
    body = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", txt, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.replace("const", " ").strip()
        if not decl:
            continue
        m = re.match(r"(\w+)\s*(\**)\s*(.*)", decl)
        base, stars, names = m.group(1), m.group(2), m.group(3)
        for nm in names.split(","):
            nm = nm.strip()
            arr = re.match(r"(\w+)\[(\d+)\]", nm)
            if arr:
                out.append((arr.group(1), C_TYPES[base] + "x" + arr.group(2)))
            else:
                st = stars + "*" * nm.count("*")
                out.append((nm.replace("*", "").strip(), C_TYPES[base + st]))
    return out


def _jl_struct_fields(name):
    body = re.search(r"^struct " + name + r"\n(.*?)^end", julia_source(), flags=re.S | re.M).group(1)
    out = []
    for line in body.strip().splitlines():
        nm, ty = line.strip().split("::")
        ty = ty.strip()
        m = re.match(r"NTuple\{(\d+),Float64\}", ty)
        out.append((nm, "f64x" + m.group(1) if m else JL_TYPES[ty]))
    return out


def test_struct_layouts_mirror_the_header():
    assert _jl_struct_fields("GPSLCNode") == _c_struct_fields("gpslc_node")
    assert _jl_struct_fields("PackHeader") == _c_struct_fields("gpslc_pack_header")
    # and the ctypes mirrors have the same field order
    assert [f[0] for f in _lib.Node._fields_] == [f[0] for f in _c_struct_fields("gpslc_node")]
    assert [f[0] for f in _lib.PackHeader._fields_] == [f[0] for f in _c_struct_fields("gpslc_pack_header")]


def test_the_shim_defines_what_its_method_bodies_use():
    src = julia_source()
    for needle in (
        "const KCTX", "function kctx()", "function ctx(g::GPSLCObject)", "function posterior_pack(g::GPSLCObject)",
        "function rbfKernelLogScalar(Xi::SupportedRBFVector",
        "function rbfKernelLog(X1::SupportedRBFMatrix, X2::SupportedRBFMatrix",
        "function rbfKernelLog(X1::SupportedRBFData, X2::SupportedRBFData",          # src/kernel.jl:34-42
        "function processCov(", "function conditionalITE(g::GPSLCObject", "function ITEDistributions(g::GPSLCObject",
        "function SATEDistributions(g::GPSLCObject", "function sampleITE(g::GPSLCObject",
        "function sampleSATE(g::GPSLCObject",                                        # src/driver.jl:108-111
        "function predictCounterfactualEffects(g::GPSLCObject", "function summarizeEstimates(samples",
        "struct HipYNormal <: Gen.Distribution{Vector{Float64}}", "function Gen.logpdf(::HipYNormal",
    ):
        assert needle in src, needle
    assert src.count("function likelihoodDistribution(") == 4                         # src/likelihood.jl: four methods
    # every helper called as GPSLCHip.<name>( in part 2 is defined in the module
    module = src[src.index("module GPSLCHip"):src.index("end # module GPSLCHip")]
    part2 = src[src.index("end # module GPSLCHip"):]
    for nm in set(re.findall(r"GPSLCHip\.(\w+)\(", part2)):
        assert re.search(r"^(?:function\s+)?" + nm + r"[!]?\(|^(?:mutable\s+)?struct " + nm + r"\b|^\s*function " + nm + r"\(",
                         module, flags=re.M), f"GPSLCHip.{nm} is used but not defined"


def _strip(src):
    src = re.sub(r'"""(?:.|\n)*?"""', '""', src)            # docstrings
    src = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', src)          # strings
    src = re.sub(r"#=.*?=#", "", src, flags=re.S)
    return re.sub(r"#[^\n]*", "", src)


def test_block_structure_balances():
    """function / if / for / while / begin / do / struct / module / let / try ... end: openers at bracket depth 0 (a `for` or
    `if` inside (...) or [...] is a generator / comprehension, not a block) must equal the `end`s."""
    src = _strip(julia_source())
    depth, opens, ends = 0, 0, 0
    for m in re.finditer(r"[()\[\]{}]|\b(function|if|for|while|begin|do|struct|module|let|try|end)\b", src):
        t = m.group(0)
        if t in "([{":
            depth += 1
        elif t in ")]}":
            depth -= 1
            assert depth >= 0
        elif depth == 0:
            if t == "end":
                ends += 1
            else:
                opens += 1
    assert depth == 0
    assert opens == ends, (opens, ends)
