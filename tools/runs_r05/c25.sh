#!/bin/bash
# round 5, call 25: MeanITE kernel instantiated for exactly 5 features (config 2: nU + nX = 1 + 4; it ran the 6-feature form) — A/B + tests
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c25.log
: > $O
PREV=causalgpslc.jl_amd/csrc/libgpslc_hip_var_prev.so
run() { timeout -k 10 300 python bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for rep in 1 2 3; do
  echo "== prev / cur (run $rep): N=1024 D=4 nU=1" | tee -a $O
  run --lib $PREV --n 1024 --d 4 --nu 1 --samples-per-step 8192 | tee -a $O
  run --n 1024 --d 4 --nu 1 --samples-per-step 8192 | tee -a $O
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "estimation or fuzz or golden or config2 or fullsize" 2>&1 | tail -3 | tee -a $O
