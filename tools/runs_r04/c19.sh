#!/bin/bash
# round 4, call 19: trailing kernel as a separate FULLONLY instantiation (no general path at all) against the single loop of round 3
# MFMA groups) against the single loop of round 3: parity, then N = 4096 (L = 1 and the 64-level region) and N = 2048
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_19
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 -m pytest tests/test_gpu_estimation.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
L=causalgpslc.jl_amd/csrc
run() {
  label=$1; shift
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --no-configs "$@" > $OUT/c.json 2> $OUT/c.err || tail -3 $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$label:', round(d['value'],1), r['kernel'][:24], round(r['achieved'],2), round(r.get('second_kernel',{}).get('achieved',0),2), 'c4', round(d.get('config4',{}).get('value',0),1), d.get('config4',{}).get('parity',{}).get('ok'))" | tee -a $OUT/log.txt
}
for rep in 1 2 3; do
run "N=4096 base" --lib $L/libgpslc_hip_var_base.so
run "N=4096 FULLONLY kernel"
done
run "N=2048 base" --lib $L/libgpslc_hip_var_base.so --n 2048 --samples-per-step 4096 --no-config4
run "N=2048 FULLONLY kernel" --n 2048 --samples-per-step 4096 --no-config4
