#!/bin/bash
# round 4, call 1: A/B of two strip-kernel knobs (make variant): STRIP_PIPE (LDS fragments one k-step ahead) and the
# timing-only STRIP_NO_CLOAD (prices the C-strip load of the per-item prologue), at N = 1024 and N = 4096
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_01
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
L=causalgpslc.jl_amd/csrc
run() {  # name lib extra...
  name=$1; lib=$2; shift 2
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --no-configs --no-config4 --n 1024 --d 4 --nu 1 --samples-per-step 8192 $lib "$@" > $OUT/c.json 2> $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); r=d['roofline']; print('N=1024 $name:', round(d['value'],1), 'fused', r.get('second_kernel',{}).get('achieved'), r.get('achieved'))" | tee -a $OUT/log.txt
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --no-configs --no-config4 $lib "$@" > $OUT/c.json 2> $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); r=d['roofline']; print('N=4096 $name:', round(d['value'],1), 'trail', round(r['achieved'],2), 'fused', round(r['second_kernel']['achieved'],2))" | tee -a $OUT/log.txt
}
for rep in 1 2; do
run base ""
run pipe "--lib $L/libgpslc_hip_var_pipe.so"
run nocl "--lib $L/libgpslc_hip_var_nocl.so" --timing-only
done
