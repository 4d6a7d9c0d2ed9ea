#!/bin/bash
# kernel time split of unit A with a 64-level sweep (BASELINE config 4 shape per GPU)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_23
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-units --levels 64 > $OUT/trace.log 2>&1
echo "rc=$?"; grep '"metric"' $OUT/trace.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python3 tools/kernel_stats_md.py $OUT/trace "L=64" 3072 > $OUT/stats.md; head -16 $OUT/stats.md
find $OUT -name "*.csv" -size +2M -delete
