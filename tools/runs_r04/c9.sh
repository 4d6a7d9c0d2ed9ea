#!/bin/bash
# round 4, call 9: software-pipelined draws kernel (NQ = 1): parity, then units B / C against the un-pipelined build
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_09
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 -m pytest tests/test_gpu_estimation.py tests/test_gpu_fullsize.py tests/test_sharded_gloo.py -m gpu -x -q > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
L=causalgpslc.jl_amd/csrc
run() {
  label=$1; shift
  timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs --no-config4 "$@" > $OUT/c.json 2> $OUT/c.err || tail -3 $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); u=d['units']; print('$label: B', round(u['B']['value'],1), round(u['B']['frac'],3), 'C draws/s', round(u['C']['value']), 'GB/s', round(u['C']['achieved']), 'frac', round(u['C']['frac'],3), 'avg ms', round(u['C']['avg_launch_ms'],3))" | tee -a $OUT/log.txt
}
for rep in 1 2 3; do
run "pipelined"
run "un-pipelined" --lib $L/libgpslc_hip_var_drawsnopipe.so
done
