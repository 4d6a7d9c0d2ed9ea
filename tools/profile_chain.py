#!/usr/bin/env python3
"""cProfile of one gpslc() chain (host-side Markov chain + GPU node scores): where the wall time goes.
usage: python tools/profile_chain.py [NEEC|IHDP]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import causalgpslc_jl_amd as gp   # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "IHDP"
path = os.path.join(ROOT, "tests", "golden", "neec", f"{which}_sampled.csv")
gp.gpslc(path, seed=1)
pr = cProfile.Profile()
pr.enable()
gp.gpslc(path, seed=1234)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
st.sort_stats("tottime").print_stats(18)
