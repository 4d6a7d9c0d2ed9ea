#!/bin/bash
# round 4, call 21: timing-only TRAIL_NO_CLOAD — the trailing kernel without its C-tile loads (accumulators zeroed): the upper
# bound of what cross-item prefetch of the C tile could buy there (last sentence)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_21
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
L=causalgpslc.jl_amd/csrc
run() {
  label=$1; shift
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --no-configs --no-config4 "$@" > $OUT/c.json 2> $OUT/c.err || tail -3 $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$label:', round(d['value'],1), r['kernel'][:24], round(r['achieved'],2), round(r.get('second_kernel',{}).get('achieved',0),2))" | tee -a $OUT/log.txt
}
for rep in 1 2 3; do
run "N=4096 prod"
run "N=4096 trailing kernel without C loads (timing only)" --lib $L/libgpslc_hip_var_trnocl.so --timing-only
done
