#!/bin/bash
# round 4, call 23: panel width re-swept with the round-4 kernels (production library), N = 4096 and N = 8192
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_23
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() {
  label=$1; shift
  timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --no-configs --no-config4 "$@" > $OUT/c.json 2> $OUT/c.err || tail -3 $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$label:', round(d['value'],1), r['kernel'][:24], round(r['achieved'],2), round(r.get('second_kernel',{}).get('achieved',0),2))" | tee -a $OUT/log.txt
}
for rep in 1 2; do
for p in 8 12 16 20; do
run "N=4096 panel $p" --panel $p
done
done
for p in 8 16; do
run "N=8192 panel $p" --n 8192 --samples-per-step 256 --panel $p
done
