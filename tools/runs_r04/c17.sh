#!/bin/bash
# round 4, call 16: Gram kernel tail barriers LDS-only (production) against __syncthreads (variant)
# them: the register staging then really runs two slabs ahead), A/B at N = 4096 and N = 1024
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_17
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
L=causalgpslc.jl_amd/csrc
run() {
  label=$1; shift
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --no-configs "$@" > $OUT/c.json 2> $OUT/c.err || tail -3 $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$label:', round(d['value'],1), r['kernel'][:24], round(r['achieved'],2), round(r.get('second_kernel',{}).get('achieved',0),2), 'parity', d.get('config4',{}).get('parity',{}).get('ok'))" | tee -a $OUT/log.txt
}
for rep in 1 2 3; do
run "N=4096 prod"
run "N=4096 round-3 barriers" --lib $L/libgpslc_hip_var_gramsync.so
done
N1="--n 1024 --d 4 --nu 1 --samples-per-step 8192 --no-config4"
for rep in 1 2; do
run "N=1024 prod" $N1
run "N=1024 round-3 barriers" --lib $L/libgpslc_hip_var_gramsync.so $N1
done
