#!/bin/bash
# round 5, call 21: the first-column panel product as the strip kernel's second phase alone (GPSLC_PANEL0_STRIP=0|1, measurement
# build) — parity suite on the production build (on), same-box A/B at N = 1024 / 512 / 4096
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c21.log
: > $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee -a $O
run() { timeout -k 10 300 python bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for rep in 1 2 3; do
for on in 0 1; do
  echo "== GPSLC_PANEL0_STRIP=$on (run $rep): N=1024 / 512 / 4096" | tee -a $O
  GPSLC_PANEL0_STRIP=$on run --n 1024 --d 4 --nu 1 --samples-per-step 8192 | tee -a $O
  GPSLC_PANEL0_STRIP=$on run --n 512 --d 4 --nu 1 --samples-per-step 16384 | tee -a $O
  GPSLC_PANEL0_STRIP=$on run | tee -a $O
done
done
