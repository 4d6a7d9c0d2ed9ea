"""Writes tests/golden/reference_signatures.json: for every function of the reference that julia/GPSLCHip.jl re-defines, the
positional parameter list (name::Type, defaults dropped) and the keyword names of EVERY method the reference declares —
interface facts read from the reference's source text (/root/reference/src, study only; Julia cannot run here).
tests/test_julia_binding.py compares the shim's signatures with them: a method whose signature drifted would ADD a method
instead of replacing the reference's body.

    python tests/golden/make_reference_signatures.py
"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"
NAMES = ["rbfKernelLogScalar", "rbfKernelLog", "processCov", "likelihoodDistribution", "conditionalITE", "ITEDistributions",
         "SATEDistributions", "sampleITE", "sampleSATE", "summarizeEstimates", "predictCounterfactualEffects"]


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def signatures(text, name):
    res = []
    for m in re.finditer(r"^function " + name + r"\(", text, flags=re.M):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        inner = re.sub(r"\s+", " ", text[m.end():i - 1]).strip()
        pos, _, kw = inner.partition(";")
        res.append({"positional": [re.sub(r"\s*=.*$", "", p).strip() for p in split_top(pos)],
                    "keywords": [re.split(r"[:=]", k)[0].strip() for k in split_top(kw)] if kw.strip() else []})
    return res


out = {}
for f in sorted(os.listdir(REF)):
    if not f.endswith(".jl"):
        continue
    text = open(os.path.join(REF, f)).read()
    for n in NAMES:
        for sig in signatures(text, n):
            out.setdefault(n, []).append(dict(sig, file="src/" + f))
json.dump(out, open(os.path.join(HERE, "reference_signatures.json"), "w"), indent=1, sort_keys=True)
print({k: len(v) for k, v in out.items()})
