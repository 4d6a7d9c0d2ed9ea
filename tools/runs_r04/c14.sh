#!/bin/bash
# round 4, call 13: __builtin_amdgcn_s_setprio variants of the strip kernel (second phase; K loop + second phase), A/B at N = 4096
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_14
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
run "N=4096 strip p2=1" --lib $L/libgpslc_hip_var_p2.so
run "N=4096 strip k=1 p2=1" --lib $L/libgpslc_hip_var_kp2.so
run "N=4096 strip k=2 p2=1" --lib $L/libgpslc_hip_var_k2p1.so
done
