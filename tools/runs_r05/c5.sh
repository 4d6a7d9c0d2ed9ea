#!/bin/bash
# round 5, call 5: unit B at the new default sub-batch (128), shapes 8x16 / 8x32 / 16x16 / 64x1, second call timed separately
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c5.log
: > $O
for shape in "8 16" "8 32" "16 16" "64 1" "128 1"; do
  echo "== $shape" | tee -a $O
  timeout -k 10 300 python tools/bench_unit_b.py 4096 $shape 10 2>&1 | tail -2 | tee -a $O
done
echo "== 8 16 at sub-batch 64 (round 4)" | tee -a $O
GPSLC_UNITB_BATCH=64 timeout -k 10 300 python tools/bench_unit_b.py --diag-lib 4096 8 16 10 2>&1 | tail -2 | tee -a $O
