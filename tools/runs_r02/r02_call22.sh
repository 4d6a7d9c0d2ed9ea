#!/bin/bash
# A/B: the two workgroups of a CU swap the heavy / light wave columns of the fused in-panel kernel's second phase
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_22
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_estimation.py tests/test_gpu_kernels.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --diag-lib"
for P in 0 1 2 0 1 2; do
  GPSLC_WC_FLIP=$P timeout -k 10 200 $B > $OUT/bench_f$P.json 2> $OUT/bench_f$P.err || { echo "P=$P failed"; tail -5 $OUT/bench_f$P.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$OUT/bench_f$P.json').read().strip().splitlines()[-1])
r=d['roofline']
print('flip=$P', round(d['value'],1), 'trailing', round(r['achieved'],2), 'fused', round(r['second_kernel']['achieved'],2))"
done
