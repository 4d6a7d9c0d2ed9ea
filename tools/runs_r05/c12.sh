#!/bin/bash
# round 5, call 12: kernel-level durations of the diagonal-block kernels with / without the lookahead (N = 1024, measurement build)
set -e
mkdir -p gpurun_out/r05
O=$GRAFT_REPO_ROOT/gpurun_out/r05/c12.log
: > $O
cd /tmp && export TMPDIR=/tmp
C2="python3 $GRAFT_REPO_ROOT/bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --n 1024 --d 4 --nu 1 --samples-per-step 8192"
for la in 0 1; do
  export GPSLC_POTRF_LA=$la
  rm -rf /tmp/tr$la
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr$la -- $C2 > /tmp/tr$la.log 2>&1
  echo "== GPSLC_POTRF_LA=$la" | tee -a $O
  python3 $GRAFT_REPO_ROOT/tools/kernel_stats_md.py /tmp/tr$la "LA=$la" 32768 | grep -E "diag|sum of" | tee -a $O
done
