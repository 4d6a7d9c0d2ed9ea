#!/bin/bash
# round 4, call 10: unit B with the W solve blocked in panels (GPSLC_W_PANEL: 0 = left-looking over the whole width, as in
# round 3; 4 / 8 / 16 = panel width) — parity first, then units B / C through bench.py (measurement build for the switch)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_10
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 -m pytest tests/test_gpu_estimation.py tests/test_gpu_fullsize.py tests/test_gpu_model_nodes.py tests/test_gpu_neec.py -m gpu -x -q > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
run() {
  label=$1; shift
  timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs --no-config4 --diag-lib "$@" > $OUT/c.json 2> $OUT/c.err || tail -3 $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); u=d['units']; print('$label: B', round(u['B']['value'],1), round(u['B']['frac'],3), 'single-level', round(u['B']['single_level']['value'],1), round(u['B']['single_level']['frac'],3), 'parity', u['B']['parity']['ok'], u['B']['parity']['draw_err'])" | tee -a $OUT/log.txt
}
for rep in 1 2; do
for w in 0 4 8 16; do
GPSLC_W_PANEL=$w run "W panel $w"
done
done
