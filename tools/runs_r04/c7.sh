#!/bin/bash
# round 4, call 7: full GPU suite (epilogue sums from the rows of R; bench rehearsal test), then
# (a) GPSLC_EPI_ROWS=0|1 (measurement build): the augmented diagonal tile update against the epilogue's own sums;
# (b) timing-only STRIP_SLAB0 at N = 1024: what the operand HBM traffic of the strip kernel costs there
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_07
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
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
GPSLC_EPI_ROWS=0 run "N=1024 epi_rows=0" --diag-lib $N1
GPSLC_EPI_ROWS=1 run "N=1024 epi_rows=1" --diag-lib $N1
GPSLC_EPI_ROWS=0 run "N=4096 epi_rows=0" --diag-lib
GPSLC_EPI_ROWS=1 run "N=4096 epi_rows=1" --diag-lib
run "N=1024 prod" $N1
run "N=1024 slab0 (timing only)" --lib $L/libgpslc_hip_var_slab0.so --timing-only $N1
done
