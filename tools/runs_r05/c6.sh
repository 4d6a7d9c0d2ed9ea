#!/bin/bash
# round 5, call 6: streaming draw kernel for 33..128 draws per unit (MFMA-bound) — parity tests + digests against the LDS-staged kernel
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c6.log
: > $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "draw or philox or sharded or multi or neec" 2>&1 | tail -4 | tee -a $O
for spp in 33 64 65 100 128 129; do
  echo "== spp=$spp  LDS-staged (GPSLC_DRAWS_STREAM=0) / stream" | tee -a $O
  GPSLC_DRAWS_STREAM=0 timeout -k 10 300 python tools/bench_draws.py --diag-lib 4096 8 4 $spp 2 2>&1 | tail -1 | tee -a $O
  timeout -k 10 300 python tools/bench_draws.py 4096 8 4 $spp 2 2>&1 | tail -1 | tee -a $O
done
echo "== n=300 spp=100, 3 levels" | tee -a $O
GPSLC_DRAWS_STREAM=0 timeout -k 10 300 python tools/bench_draws.py --diag-lib 300 6 3 100 1 2>&1 | tail -1 | tee -a $O
timeout -k 10 300 python tools/bench_draws.py 300 6 3 100 1 2>&1 | tail -1 | tee -a $O
