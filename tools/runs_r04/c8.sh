#!/bin/bash
# round 4, call 8: parity of the live-rows-only augmented tiles, then timing-only STRIP_W0 at N = 1024: what do the second
# phase's inv(L_kk) fragment loads cost when only ~4 items share an inverse block (N = 4096: 24)?
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_08
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
L=causalgpslc.jl_amd/csrc
run() {
  label=$1; shift
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --no-configs --no-config4 "$@" > $OUT/c.json 2> $OUT/c.err || tail -3 $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$label:', round(d['value'],1), r['kernel'][:24], round(r['achieved'],2), r.get('second_kernel',{}).get('achieved'))" | tee -a $OUT/log.txt
}
N1="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
for rep in 1 2; do
run "N=1024 prod" $N1
run "N=1024 w0 (timing only)" --lib $L/libgpslc_hip_var_w0.so --timing-only $N1
run "N=1024 w0+slab0 (timing only)" --lib $L/libgpslc_hip_var_w0slab0.so --timing-only $N1
run "N=4096 prod"
run "N=4096 w0 (timing only)" --lib $L/libgpslc_hip_var_w0.so --timing-only
done
