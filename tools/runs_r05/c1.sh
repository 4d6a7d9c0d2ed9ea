#!/bin/bash
# round 5, call 1: streaming draws kernel — parity tests, bit-identity digest vs the LDS kernel, register-set variants
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c1.log
: > $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "draw or philox or sharded or multi or plain_c" 2>&1 | tail -5 | tee -a $O
echo "== old LDS kernel (diag lib, GPSLC_DRAWS_STREAM=0)" | tee -a $O
GPSLC_DRAWS_STREAM=0 timeout -k 10 300 python tools/bench_draws.py --diag-lib 4096 8 8 10 3 2>&1 | tail -1 | tee -a $O
echo "== production lib" | tee -a $O
timeout -k 10 300 python tools/bench_draws.py 4096 8 8 10 3 2>&1 | tail -1 | tee -a $O
for v in 6 3 1 2 4 5; do
  echo "== GPSLC_DRAWS_VAR=$v" | tee -a $O
  GPSLC_DRAWS_VAR=$v timeout -k 10 300 python tools/bench_draws.py --diag-lib 4096 8 8 10 3 2>&1 | tail -1 | tee -a $O
done
echo "== var 6 (<32,2,2>) with 2 WG/CU (LDS pad 72 KiB)" | tee -a $O
GPSLC_DRAWS_VAR=6 GPSLC_DRAWS_LDS=72 timeout -k 10 300 python tools/bench_draws.py --diag-lib 4096 8 8 10 3 2>&1 | tail -1 | tee -a $O
echo "== var 6 with 3 WG/CU (LDS pad 48 KiB)" | tee -a $O
GPSLC_DRAWS_VAR=6 GPSLC_DRAWS_LDS=48 timeout -k 10 300 python tools/bench_draws.py --diag-lib 4096 8 8 10 3 2>&1 | tail -1 | tee -a $O
echo "== N=2048 / 1024 production" | tee -a $O
timeout -k 10 300 python tools/bench_draws.py 2048 16 8 10 3 2>&1 | tail -1 | tee -a $O
GPSLC_DRAWS_STREAM=0 timeout -k 10 300 python tools/bench_draws.py --diag-lib 2048 16 8 10 3 2>&1 | tail -1 | tee -a $O
