#!/bin/bash
# round 4, call 3: the chained in-panel launch (tile_fused_chain_kernel): parity suite, then A/B against the three-launch
# column (measurement build, GPSLC_CHAIN=0) at N = 1024 and N = 4096
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_03
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 -m pytest tests/test_gpu_estimation.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_model_nodes.py -m gpu -x -q > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
run() {
  label=$1; shift
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --no-configs --no-config4 --diag-lib "$@" > $OUT/c.json 2> $OUT/c.err || tail -3 $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$label:', round(d['value'],1), r['kernel'][:24], round(r['achieved'],2), r.get('second_kernel',{}).get('achieved'))" | tee -a $OUT/log.txt
}
N1="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
for rep in 1 2; do
GPSLC_CHAIN=1 run "N=1024 chain" $N1
GPSLC_CHAIN=0 run "N=1024 3-launch" $N1
GPSLC_CHAIN=1 run "N=4096 chain"
GPSLC_CHAIN=0 run "N=4096 3-launch"
done
GPSLC_CHAIN=1 run "N=2048 chain" --n 2048 --samples-per-step 4096
GPSLC_CHAIN=0 run "N=2048 3-launch" --n 2048 --samples-per-step 4096
