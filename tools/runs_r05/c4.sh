#!/bin/bash
# round 5, call 4: streaming draw kernel for 17..32 draws per unit — parity tests + digests against the LDS-staged kernel
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c4.log
: > $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "draw or philox or sharded or multi or unit_b or neec" 2>&1 | tail -4 | tee -a $O
for spp in 17 20 32 16 33; do
  echo "== spp=$spp  LDS-staged (GPSLC_DRAWS_STREAM=0) / stream" | tee -a $O
  GPSLC_DRAWS_STREAM=0 timeout -k 10 300 python tools/bench_draws.py --diag-lib 4096 8 8 $spp 2 2>&1 | tail -1 | tee -a $O
  timeout -k 10 300 python tools/bench_draws.py 4096 8 8 $spp 2 2>&1 | tail -1 | tee -a $O
done
echo "== n=300 spp=20, 3 levels (level sweep staging)" | tee -a $O
GPSLC_DRAWS_STREAM=0 timeout -k 10 300 python tools/bench_draws.py --diag-lib 300 6 3 20 1 2>&1 | tail -1 | tee -a $O
timeout -k 10 300 python tools/bench_draws.py 300 6 3 20 1 2>&1 | tail -1 | tee -a $O
echo "== unit B default (sub-batch 128)" | tee -a $O
timeout -k 10 300 python tools/bench_unit_b.py 4096 8 16 10 2>&1 | tail -1 | tee -a $O
