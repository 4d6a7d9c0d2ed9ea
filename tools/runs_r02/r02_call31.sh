#!/bin/bash
# trailing-update kernel with the LDS fragment reads one k-step ahead: parity, stamps, throughput
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_31
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_estimation.py tests/test_gpu_kernel.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
GPSLC_GEMM_DBG=25 timeout -k 10 200 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-units --diag-lib > /dev/null 2>&1; python3 tools/gemm_stamps.py | head -2
for i in 1 2; do timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units > $OUT/bench_$i.json 2> $OUT/bench_$i.err; python3 -c "
import json
d=json.loads(open('$OUT/bench_$i.json').read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value'],1), 'trailing', round(r['achieved'],2), 'fused', round(r['second_kernel']['achieved'],2))"; done
