#!/bin/bash
# round 5, call 22: W-solve column 0 through the strip kernel too — parity suite, unit B, then the final kernel traces + config-2 PMC
# passes (tools/profile_r05.sh part a)
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c22.log
: > $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee -a $O
timeout -k 10 300 python tools/bench_unit_b.py 4096 8 16 10 2>&1 | tail -1 | tee -a $O
bash tools/profile_r05.sh r05p a 2>&1 | tail -25 | tee -a $O
