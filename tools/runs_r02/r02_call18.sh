#!/bin/bash
# kernel time split at config 2 (N = 1024, D 4, nU 1)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_18
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --n 1024 --d 4 --nu 1 --samples-per-step 8192 > $OUT/trace.log 2>&1
echo "rc=$?"; grep '"metric"' $OUT/trace.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python3 tools/kernel_stats_md.py $OUT/trace "N=1024" 32768 > $OUT/stats.md; cat $OUT/stats.md
find $OUT -name "*.csv" -size +2M -delete
