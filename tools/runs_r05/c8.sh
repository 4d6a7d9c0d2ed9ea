#!/bin/bash
# round 5, call 8: staging kernel shares one Philox / Box-Muller between partner lanes — digests vs the LDS-staged path, parity tests,
# then the kernel traces + config-2 PMC passes (tools/profile_r05.sh part a)
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c8.log
: > $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "draw or philox or sharded or multi or neec or plain_c" 2>&1 | tail -3 | tee -a $O
for args in "4096 8 8 10 3" "4096 8 4 20 2" "4096 8 4 100 2" "301 6 3 10 1" "300 6 3 10 1"; do
  echo "== $args  LDS-staged / stream" | tee -a $O
  GPSLC_DRAWS_STREAM=0 timeout -k 10 300 python tools/bench_draws.py --diag-lib $args 2>&1 | tail -1 | tee -a $O
  timeout -k 10 300 python tools/bench_draws.py $args 2>&1 | tail -1 | tee -a $O
done
bash tools/profile_r05.sh r05p a 2>&1 | tail -30 | tee -a $O
