#!/bin/bash
# A/B: Gram build one workgroup per (tile, sample) vs persistent workgroups (measurement build switch)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_15
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --diag-lib"
for P in 0 4 0 4 3 5 8; do
  GPSLC_GRAM_PERSIST=$P timeout -k 10 200 $B > $OUT/bench_p$P.json 2> $OUT/bench_p$P.err || { echo "P=$P failed"; tail -5 $OUT/bench_p$P.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_p$P.json').read().strip().splitlines()[-1])
print('persist=$P', d['value'], d['unit'], 'sate_rel_err', d.get('sate_rel_err'))"
done
cd /tmp && export TMPDIR=/tmp
export GPSLC_GRAM_PERSIST=4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_p4 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --diag-lib > $OUT/trace_p4.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kernel_stats_md.py $OUT/trace_p4 "persist=4" 4096 | head -12
find $OUT -name "*.csv" -size +2M -delete
