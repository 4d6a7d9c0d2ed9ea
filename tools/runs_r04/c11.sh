#!/bin/bash
# round 4, call 11: diagonal tile of an in-panel column in one launch (update + Cholesky + inverse: diag_update_potrf_kernel)
# against the two launches it replaces (GPSLC_DIAG_FOLD=0, measurement build): parity, then N = 1024 / 2048 / 4096
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_11
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 -m pytest tests/test_gpu_estimation.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_model_nodes.py tests/test_gpu_abi_edges.py -m gpu -x -q > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
run() {
  label=$1; shift
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --no-configs --no-config4 --diag-lib "$@" > $OUT/c.json 2> $OUT/c.err || tail -3 $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$label:', round(d['value'],1), r['kernel'][:24], round(r['achieved'],2))" | tee -a $OUT/log.txt
}
N1="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
for rep in 1 2; do
for m in 0 1; do
GPSLC_DIAG_FOLD=$m run "N=1024 fold=$m" $N1
done
for m in 0 1; do
GPSLC_DIAG_FOLD=$m run "N=4096 fold=$m"
done
done
for m in 0 1; do
GPSLC_DIAG_FOLD=$m run "N=2048 fold=$m" --n 2048 --samples-per-step 4096
GPSLC_DIAG_FOLD=$m run "N=512 fold=$m" --n 512 --d 4 --nu 1 --samples-per-step 16384
done
