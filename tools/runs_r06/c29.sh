#!/bin/bash
# round 6, call 29: task launch up to nt = 32 by default (N = 4096 as one left-looking panel): tests, suite, smoke, rows per strip task
# at N = 4096 (measurement build), the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c29; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_tasks.py -m gpu -x -q > $O/tasks.log 2>&1; echo "tasks rc=$?"; tail -2 $O/tasks.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -2 $O/suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
B="python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for r in 1 2 3 4; do
GPSLC_TASK_ROWS=$r timeout -k 10 300 $B --diag-lib > $O/rows$r.json 2> $O/err.txt; val $O/rows$r.json "N=4096 rows per strip task=$r (measurement build)"
done
S=$(date +%s); timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - S )) s"; cut -c1-400 $O/bench_default.json
