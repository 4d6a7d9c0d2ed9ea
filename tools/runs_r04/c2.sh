#!/bin/bash
# round 4, call 2: can the latency-bound kernels of one chunk (diag_potrf_inv, backsolve, syrk_diag) hide behind the
# MFMA kernels of ANOTHER chunk?  Two HIP streams, with the persistent tile kernels leaving some workgroup slots free
# (measurement build: GPSLC_GEMM_SLOTS), N = 1024 (two 4,096-sample chunks per step) and N = 4096 (batch 512 -> two chunks)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_02
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() {  # label n-args... 
  label=$1; shift
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --no-configs --no-config4 --diag-lib "$@" > $OUT/c.json 2> $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); print('$label:', round(d['value'],1))" | tee -a $OUT/log.txt
}
N1="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
for slots in 512 480 448 384; do
  for st in 1 2; do
    GPSLC_GEMM_SLOTS=$slots run "N=1024 slots=$slots streams=$st" $N1 --streams $st
  done
done
GPSLC_GEMM_SLOTS=448 run "N=1024 slots=448 streams=2 batch=2048" $N1 --streams 2 --max-batch 2048
GPSLC_GEMM_SLOTS=448 run "N=1024 slots=448 streams=4 batch=2048" $N1 --streams 4 --max-batch 2048
for slots in 512 448; do
  for st in 1 2; do
    GPSLC_GEMM_SLOTS=$slots run "N=4096 slots=$slots streams=$st batch=512" --streams $st --max-batch 512
  done
done
