#!/bin/bash
# round 5, call 11: diagonal-block factorisation with a one-block lookahead (GPSLC_POTRF_LA=0|1, measurement build) — parity suite on the
# production build (lookahead on), then same-box A/B at N = 512 / 1024 / 2048 / 4096
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c11.log
: > $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee -a $O
run() { timeout -k 10 300 python bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for rep in 1 2; do
for la in 0 1; do
  echo "== GPSLC_POTRF_LA=$la (run $rep): N=1024 / 512 / 4096" | tee -a $O
  GPSLC_POTRF_LA=$la run --n 1024 --d 4 --nu 1 --samples-per-step 8192 | tee -a $O
  GPSLC_POTRF_LA=$la run --n 512 --d 4 --nu 1 --samples-per-step 16384 | tee -a $O
  GPSLC_POTRF_LA=$la run | tee -a $O
done
done
echo "== N=2048, LA 0 / 1" | tee -a $O
GPSLC_POTRF_LA=0 run --n 2048 --d 8 --nu 2 --samples-per-step 4096 | tee -a $O
GPSLC_POTRF_LA=1 run --n 2048 --d 8 --nu 2 --samples-per-step 4096 | tee -a $O
