#!/bin/bash
# Gram build: persistent workgroups again, on top of the batched staging (measurement build switch), with stamps
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_29
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for P in 0 4 5; do
  GPSLC_GRAM_PERSIST=$P GPSLC_GRAM_DBG=1 timeout -k 10 200 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-units --diag-lib --samples-per-step 256 > /dev/null 2>&1
  echo "== persist=$P"; python3 tools/gram_stamps.py | head -1
done
cd /tmp && export TMPDIR=/tmp
for P in 0 4 5; do
  export GPSLC_GRAM_PERSIST=$P
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_p$P -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-units --diag-lib > $OUT/trace_p$P.log 2>&1
  (cd $GRAFT_REPO_ROOT && python3 tools/kernel_stats_md.py $OUT/trace_p$P "persist=$P" 3072 | grep -E "gram|sum of")
done
find $OUT -name "*.csv" -size +2M -delete
