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
    "Ref{Ptr{Cvoid}}": "ctxpp", "Ptr{Ptr{Cvoid}}": "ctxpp", "Ptr{Cvoid}": "ctx", "Cint": "int", "Int64": "i64", "Int32": "i32", "UInt32": "u32",
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
        "const KCTX", "function kctx()", "ctx(g::GPSLCObject) = _device_side(g).ctx",
        "function posterior_pack(g::GPSLCObject)", "function _device_side(g::GPSLCObject)",
        "function release!(g::GPSLCObject)", "function dctx(n::Integer, nX::Integer, nU::Integer)",
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


# ---- round 4: the defects a reader found that the type-tuple check cannot see --------
def _part2():
    src = julia_source()
    return src[src.index("end # module GPSLCHip"):]


def _function_blocks(src):
    """{signature line: body} of every top-level `function ... end` block (column-0 `function`, column-0 `end`)."""
    out, lines, i = [], src.splitlines(), 0
    while i < len(lines):
        if lines[i].startswith("function "):
            j = i
            while not lines[j].startswith("end"):
                j += 1
            blk = "\n".join(lines[i:j + 1])
            # the signature runs to the first line whose parentheses balance
            depth, k = 0, i
            while True:
                depth += lines[k].count("(") - lines[k].count(")")
                if depth == 0:
                    break
                k += 1
            out.append(("\n".join(lines[i:k + 1]), "\n".join(lines[k + 1:j])))
            i = j
        i += 1
    return out


def _split_top(argstr):
    parts, depth, cur = [], 0, ""
    for ch in argstr:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return parts


def _params(sig):
    inner = sig[sig.index("(") + 1:sig.rindex(")")]
    pos = inner.split(";")[0] if ";" in inner else inner
    return _split_top(pos)


DISTRIBUTIONS = ("HipYNormal", "HipGPNormal", "HipMvNormal")


def test_gen_distribution_arities_follow_the_logpdf_signature():
    """Gen's interface: logpdf(dist, value, args...), random(dist, args...), has_argument_grads -> one Bool per arg,
    logpdf_grad -> (value grad, one per arg).  Round 3 shipped 9 / 10 for a distribution with 8 arguments."""
    src = _part2()
    blocks = _function_blocks(src)
    for D in DISTRIBUTIONS:
        lp = [sig for sig, _ in blocks if sig.startswith(f"function Gen.logpdf(::{D},")]
        rd = [sig for sig, _ in blocks if sig.startswith(f"function Gen.random(::{D},")]
        assert len(lp) == 1 and len(rd) == 1, D
        nargs = len(_params(lp[0])) - 2                      # minus the distribution and the value
        assert len(_params(rd[0])) - 1 == nargs, (D, "random takes the same arguments as logpdf, without the value")
        m = re.search(r"Gen\.has_argument_grads\(::" + D + r"\) = ntuple\(_ -> false, (\d+)\)", src)
        g = re.search(r"Gen\.logpdf_grad\(::" + D + r", \w+, args\.\.\.\) = ntuple\(_ -> nothing, (\d+)\)", src)
        assert m and g, D
        assert int(m.group(1)) == nargs, (D, m.group(1), nargs)
        assert int(g.group(1)) == nargs + 1, (D, g.group(1), nargs + 1)
        for needle in (f"struct {D} <: Gen.Distribution{{Vector{{Float64}}}} end", f"Gen.has_output_grad(::{D}) = false",
                       f"(d::{D})(args...) = Gen.random(d, args...)", f"_register_continuous({D})"):
            assert needle in src, needle


def test_one_distribution_per_node_kind_with_its_trace_replacement_text():
    """next-1's reference-side half: :U => u => :U (src/model_likelihood.jl:4-10), :X => k => :X (:13-22), :T / :logitT
    (:25-80), :Y (:83-120) — each has a distribution and the line that replaces the @gen body's @trace."""
    src = _part2()
    for needle in ("@trace(hip_mv_normal(SigmaU, uNoise), :U)",
                   "@trace(hip_gp_normal(F, ls, xScale[k], xNoise[k]), :X => k => :X)",
                   "@trace(hip_gp_normal(F, ls, tScale, tNoise), :T)", ":logitT",
                   "@trace(hip_gp_normal(F, ls, yScale, yNoise), :Y)",
                   "@trace(hip_y_normal(c, U, X, uyLS, xyLS, tyLS, yScale, yNoise), :Y)",
                   "_features(U => utLS, X => xtLS)", "_features(U => uyLS, X => xyLS, T => tyLS)"):
        assert needle in src, needle
    module = julia_source()
    for helper in ("function gp_score(c::Ctx", "function draw(c::Ctx", "function mvn_score(c::Ctx"):
        assert helper in module, helper


def test_every_U_goes_through_umat():
    """`Confounders` includes nested vectors — the reference's own estimation tests pass U = [[1.0]]
    (test/test_data.jl:42) — so no function of part 2 may convert a U itself or read nU off size(U, 2)."""
    src = _strip(_part2())                      # code only: the comments explain what not to do
    assert "size(U, 2)" not in src
    outside = "\n".join(l for l in src.splitlines() if not l.startswith("_umat("))     # _umat itself converts
    assert not re.search(r"f64\(U\)", outside)
    for form in ("_umat(::Nothing)", "_umat(U::AbstractMatrix{<:Real})", "_umat(U::AbstractVector{<:Real})",
                 "_umat(U::AbstractVector{<:AbstractVector})"):
        assert form in src, form
    forwards = ("_likelihood_blocks(", "conditionalITE(", "hip_y_normal(")
    checked = 0
    for sig, body in _function_blocks(src):
        if sig.startswith("function _umat"):
            continue
        if not any(re.match(r"U(::|$)", prm) for prm in _params(sig)):
            continue
        checked += 1
        if "_umat(U)" in body:
            continue
        # otherwise U may only be asserted on (size(U, 1): the same for every form) or forwarded whole
        for line in body.splitlines():
            if re.search(r"\bU\b", line):
                assert "size(U, 1)" in line or any(f in line for f in forwards), (sig.splitlines()[0], line)
    assert checked >= 8                       # 4 likelihoodDistribution methods, _likelihood_blocks, conditionalITE, 2 x HipYNormal


def test_device_side_cache_is_weak_and_releasable():
    src = _part2()
    assert "IdDict" not in src                                     # round 3: strong IdDicts keyed by g, never emptied
    blk = dict((sig.splitlines()[0], body) for sig, body in _function_blocks(src))
    body = blk["function _device_side(g::GPSLCObject)"]
    assert "WeakRef(ps)" in body and "finalizer(" in body and "destroy!" in body and "filter!(" in body
    assert "GPSLCHip.destroy!(d.ctx)" in blk["function release!(g::GPSLCObject)"]
    # no context is created and destroyed per call inside the parameter-level functions
    for sig, body in _function_blocks(src):
        if sig.startswith(("function conditionalITE(uyLS", "function _likelihood_blocks(")):
            assert "GPSLCHip.dctx(" in body and "GPSLCHip.Ctx(" not in body and "destroy!" not in body, sig


def test_draw_entry_points_offer_the_seeded_path():
    """predictCounterfactualEffects at BASELINE config 4 would allocate randn(4096, 10, 8192, 64) = 172 GB on the host."""
    src = _part2()
    for sig, body in _function_blocks(src):
        if sig.startswith(("function sampleITE(g", "function predictCounterfactualEffects(g", "function sampleSATE(g")):
            assert "seed::Union{Nothing,Integer}=nothing" in sig, sig
        if sig.startswith(("function sampleITE(g", "function predictCounterfactualEffects(g")):
            assert "_normals(" in body and "seed=sd" in body and "randn(" not in body, sig
    body = dict((s_.splitlines()[0], b) for s_, b in _function_blocks(src))["function _normals(n, spp, S, L, seed)"]
    assert "_HOST_NORMALS_MAX" in body and "rand(Random.default_rng(), UInt64)" in body


def header_param_names():
    txt = open(_lib.HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(gpslc_\w+)\s*\(([^)]*)\)\s*;", txt):
        args = m.group(2).strip()
        out[m.group(1)] = [] if args in ("", "void") else [re.findall(r"\w+", a)[-1] for a in args.split(",")]
    return out


# what the wrappers call their marshalled copies of the header's parameters
JL_ALIASES = {"uf": "u", "xf": "x", "yf": "y", "tf": "t", "uy": "uyls", "xy": "xyls", "ty": "tyls", "yn": "ynoise",
              "ys": "yscale", "ms": "meansate", "vs": "varsate", "mi": "meanite", "dr": "ite_draws", "zf": "z",
              "m": "meanites", "cv": "covites", "cf": "cov", "sf": "covscale", "x1": "x1", "x2": "x2", "lp": "logpdf",
              "out": None, "h": None, "r": None, "l": None, "fl": None, "st": None}
SAME_TYPED = {"u", "uyls", "xyls", "tyls", "yscale", "ynoise", "x", "t", "y", "dot", "meansate", "varsate", "meanite",
              "ite_draws", "meanites", "covites", "mean", "lower", "upper", "ls", "f", "target", "scale", "noise",
              "cov", "covscale", "z", "x1", "x2", "logcov", "samples", "draws"}


def _canon_expr(e):
    e = re.sub(r"\b(?:pointer|ptr|Ref)\(", "", e).rstrip(")")
    e = re.sub(r"^(?:p|c)\.", "", e)
    e = re.sub(r"\[\d+\]$", "", e)
    e = e.split("?")[0].strip() if "?" in e else e
    e = e.lower()
    return JL_ALIASES.get(e, e)


def test_ccall_arguments_follow_the_header_parameter_order_by_name():
    """Same-typed pointer arguments (yNoise / yScale, mean / lower / upper, ...) cannot be told apart by the type tuple:
    compare the NAME of the expression passed at each position with the header's parameter name."""
    names = header_param_names()
    src = julia_source()
    seen = 0
    for m in re.finditer(r"ccall\(\(:(\w+),\s*lib\),\s*\w+,\s*\(", src):
        name = m.group(1)
        i, depth = m.end(), 1                    # skip the type tuple
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        j, depth = i, 1                          # the ccall's own closing parenthesis
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        exprs = _split_top(src[i:j - 1].lstrip(", \n"))
        assert len(exprs) == len(names[name]), (name, exprs, names[name])
        for e, pname in zip(exprs, names[name]):
            want = re.sub(r"_or_null$", "", pname).lower()
            if want not in SAME_TYPED:
                continue
            got = _canon_expr(e)
            if got is None or got.startswith("blocks") or got == "c_null":
                continue
            assert got == want, f"{name}: `{e}` is passed where the header has `{pname}`"
            seen += 1
    assert seen >= 100


def test_shim_methods_have_the_reference_signatures():
    """Every method of part 2 that is meant to REPLACE a reference method must have that method's positional signature — same
    parameter types in the same order, otherwise Julia adds a new method and the reference's body keeps running — and must accept
    the reference's keywords (it may add its own, e.g. `seed`).  The reference's signatures are interface facts read from its
    source text into tests/golden/reference_signatures.json (tests/golden/make_reference_signatures.py)."""
    import json
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_signatures.json")))
    src = _part2()
    types = lambda plist: [p.split("::", 1)[1].strip() if "::" in p else "Any" for p in plist]   # noqa: E731
    for name, sigs in ref.items():
        mine = []
        for m in re.finditer(r"^function " + name + r"\(", src, flags=re.M):
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(src[i], 0)
                i += 1
            inner = re.sub(r"\s+", " ", src[m.end():i - 1]).strip()
            pos, _, kw = inner.partition(";")
            mine.append(([re.sub(r"\s*=.*$", "", p).strip() for p in _split_top(pos)],
                         [re.split(r"[:=]", k)[0].strip() for k in _split_top(kw)] if kw.strip() else []))
        assert len(mine) == len(sigs), (name, len(mine), len(sigs))
        for sig in sigs:
            match = [kws for pos, kws in mine if types(pos) == types(sig["positional"])]
            assert len(match) == 1, f"{name}{sig['positional']} ({sig['file']}) has no method with the same positional types in the shim"
            assert set(sig["keywords"]) <= set(match[0]), (name, sig["keywords"], match[0])


# ---- round 5: VERDICT r04 "next" items 2 and 6, ADVICE r04 (Julia shim residuals) ---------------------------------------
def _blocks_by_first_line(src):
    return dict((sig.splitlines()[0], (sig, body)) for sig, body in _function_blocks(src))


def test_likelihood_distribution_returns_symmetric_wrappers_where_the_reference_does():
    """src/likelihood.jl:31-32, 51: CovWW and CovWWp are `Symmetric` in the tuple the reference returns (positions 2 and 4)."""
    sig, body = _blocks_by_first_line(_part2())["function _likelihood_blocks(uyLS, xyLS, tyLS, yNoise, yScale, U, X, T, Y, doT)"]
    ret = [l for l in _strip(body).splitlines() if l.strip().startswith("Y, ")]
    assert len(ret) == 1
    parts = _split_top(ret[0].strip())
    assert len(parts) == 8
    assert parts[1] == "LinearAlgebra.Symmetric(b[1])" and parts[3] == "LinearAlgebra.Symmetric(b[3])"
    assert [p for i, p in enumerate(parts) if i not in (1, 3)] == ["Y", "b[2]", "b[4]", "b[5]", "b[6]", "b[7]"]
    assert "import LinearAlgebra" in _part2()


def test_scalar_kernel_stays_on_the_host():
    """rbfKernelLogScalar is five flops (src/kernel.jl:17): no ccall, no context, no upload may be reachable from it."""
    blocks = _blocks_by_first_line(_part2())
    key = [k for k in blocks if k.startswith("function rbfKernelLogScalar(")]
    assert len(key) == 1
    body = _strip(blocks[key[0]][1])
    assert "GPSLCHip." not in body and "ccall" not in body and "kctx" not in body
    assert "-sum((Xi .- Xiprime) .^ 2 ./ LS .^ 2)" in body


def test_dctx_evicts_one_least_recently_used_context():
    src = julia_source()
    module = src[src.index("module GPSLCHip"):src.index("end # module GPSLCHip")]
    body = dict((sig.splitlines()[0], b) for sig, b in _function_blocks(module))["function dctx(n::Integer, nX::Integer, nU::Integer)"]
    assert "empty!(DCTX)" not in body and "foreach(destroy!" not in body        # round 4 destroyed all eight at once
    assert "argmin(DCTX_USED)" in body and "delete!(DCTX, lru)" in body and "DCTX_USED[key] =" in body
    assert "destroy!" not in body                                               # a caller may still hold the evicted context


def test_ensemble_entry_points_reach_several_gpus_with_one_ccall():
    """VERDICT r04 missing #1: predictCounterfactualEffects / sampleITE / sampleSATE / SATEDistributions take `devices` and go
    through ONE gpslc_predict_multi call; the contexts per device hold the data and are freed with the object."""
    src = _part2()
    blocks = _blocks_by_first_line(src)
    for start in ("function predictCounterfactualEffects(g::GPSLCObject", "function sampleITE(g::GPSLCObject",
                  "function sampleSATE(g::GPSLCObject", "function SATEDistributions(g::GPSLCObject"):
        key = [k for k in blocks if k.startswith(start)]
        assert len(key) == 1, start
        sig, body = blocks[key[0]]
        assert "devices=nothing" in sig, sig
        assert "_predict(g, devices," in body or "devices=devices" in body, start
        assert "GPSLCHip.predict(" not in body, start
    sig, body = blocks["function _predict(g::GPSLCObject, devices, doT::Vector{Float64}; kw...)"]
    assert "GPSLCHip.predict_multi(ctxs(g, devices), posterior_pack(g), doT, pn; kw...)" in body
    assert "devices === nothing && return GPSLCHip.predict(ctx(g), posterior_pack(g), doT, pn; kw...)" in body
    sig, body = blocks["function ctxs(g::GPSLCObject, devices)"]
    assert "GPSLCHip.Ctx(d.ctx.n, d.ctx.nX, d.ctx.nU; device=dev)" in body and "GPSLCHip.set_data!(c, g.X, g.T, g.Y)" in body
    module = julia_source()
    assert module.count("ccall((:gpslc_predict_multi, lib)") == 1
    assert "ctxs = Ptr{Cvoid}[c.h for c in cs]" in module and "GC.@preserve cs ctxs" in module


def test_device_side_key_covers_data_and_hyperparameters():
    """two GPSLCObjects sharing one posteriorSamples vector but differing in data or hyper-parameters must not get
    each other's context / pack; dead entries drop their pack."""
    src = _part2()
    blocks = _blocks_by_first_line(src)
    assert "_ids(g::GPSLCObject) = (objectid(g.hyperparams), objectid(g.X), objectid(g.T), objectid(g.Y))" in src
    body = blocks["function _device_side(g::GPSLCObject)"][1]
    assert "d.ids != _ids(g)" in body and "_drop!(v)" in body
    drop = blocks["function _drop!(d::_DeviceSide)"][1]
    assert "d.pack = nothing" in drop and "GPSLCHip.destroy!(d.ctx)" in drop
