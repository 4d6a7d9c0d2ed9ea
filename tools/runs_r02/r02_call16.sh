#!/bin/bash
# timing experiment: Gram build without its tile stores (measurement build; results are garbage, only the kernel time is read)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPSLC_GRAM_NOSTORE=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-units --diag-lib > $OUT/trace.log 2>&1
echo "rc=$?"; tail -2 $OUT/trace.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python3 tools/kernel_stats_md.py $OUT/trace "nostore" 2048 > $OUT/stats.md; grep -E "gram|ite_mean" $OUT/stats.md
find $OUT -name "*.csv" -size +2M -delete
