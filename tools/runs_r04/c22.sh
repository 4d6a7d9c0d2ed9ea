#!/bin/bash
# round 4, call 22: chunk size at small N (the automatic choice caps a chunk at 4,096 posterior samples)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_22
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() {
  label=$1; shift
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --no-configs --no-config4 "$@" > $OUT/c.json 2> $OUT/c.err || tail -3 $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); print('$label:', round(d['value'],1))" | tee -a $OUT/log.txt
}
for rep in 1 2; do
for mb in 0 8192 16384; do
run "N=1024 S=16384 max-batch=$mb" --n 1024 --d 4 --nu 1 --samples-per-step 16384 --max-batch $mb
done
done
for mb in 0 8192 16384 32768; do
run "N=512 S=32768 max-batch=$mb" --n 512 --d 4 --nu 1 --samples-per-step 32768 --max-batch $mb
done
for mb in 0 2048; do
run "N=2048 S=4096 max-batch=$mb" --n 2048 --samples-per-step 4096 --max-batch $mb
done
