#!/bin/bash
# round 4, call 4: chained in-panel launch WITHOUT the factorisation (GPSLC_CHAIN=1: the diagonal-tile update rides with tile
# (k+1, k), the diagonal-block kernel stays a launch of its own) against 0 (three launches) and 2 (factorisation chained too)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_04
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 -m pytest tests/test_gpu_estimation.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
run() {
  label=$1; shift
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --no-configs --no-config4 --diag-lib "$@" > $OUT/c.json 2> $OUT/c.err || tail -3 $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$label:', round(d['value'],1), r['kernel'][:24], round(r['achieved'],2), r.get('second_kernel',{}).get('achieved'))" | tee -a $OUT/log.txt
}
N1="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
for rep in 1 2; do
for m in 0 1 2; do
GPSLC_CHAIN=$m run "N=1024 chain=$m" $N1
done
for m in 0 1 2; do
GPSLC_CHAIN=$m run "N=4096 chain=$m"
done
done
for m in 0 1; do
GPSLC_CHAIN=$m run "N=2048 chain=$m" --n 2048 --samples-per-step 4096
GPSLC_CHAIN=$m run "N=4096 panel16 chain=$m" --panel 16
done
