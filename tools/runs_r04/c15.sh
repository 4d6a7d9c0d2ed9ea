#!/bin/bash
# round 4, call 15: diagonal-tile update loops with LDS-only barriers and SYRK_PF register sets in flight (2 / 4 / 8) against
# the round-3 loop (two sets, __syncthreads): parity, then N = 1024 / 4096 / 512
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_15
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 -m pytest tests/test_gpu_estimation.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_model_nodes.py -m gpu -x -q > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
L=causalgpslc.jl_amd/csrc
run() {
  label=$1; shift
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --no-configs --no-config4 "$@" > $OUT/c.json 2> $OUT/c.err || tail -3 $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$label:', round(d['value'],1), r['kernel'][:24], round(r['achieved'],2))" | tee -a $OUT/log.txt
}
N1="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
for rep in 1 2; do
run "N=1024 base" --lib $L/libgpslc_hip_var_base.so $N1
run "N=1024 pf2" --lib $L/libgpslc_hip_var_pf2.so $N1
run "N=1024 pf4 (prod)" $N1
run "N=1024 pf8" --lib $L/libgpslc_hip_var_pf8.so $N1
run "N=4096 base" --lib $L/libgpslc_hip_var_base.so
run "N=4096 pf2" --lib $L/libgpslc_hip_var_pf2.so
run "N=4096 pf4 (prod)"
run "N=4096 pf8" --lib $L/libgpslc_hip_var_pf8.so
done
run "N=512 base" --lib $L/libgpslc_hip_var_base.so --n 512 --d 4 --nu 1 --samples-per-step 16384
run "N=512 pf4 (prod)" --n 512 --d 4 --nu 1 --samples-per-step 16384
run "N=2048 base" --lib $L/libgpslc_hip_var_base.so --n 2048 --samples-per-step 4096
run "N=2048 pf4 (prod)" --n 2048 --samples-per-step 4096
