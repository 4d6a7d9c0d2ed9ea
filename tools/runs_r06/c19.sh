#!/bin/bash
# round 6, call 19: back-substitution task with three register sets in rotation
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c19; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_tasks.py tests/test_gpu_fuzz.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile"
C2="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value']))"; }
for rep in 1 2 3; do
timeout -k 10 200 $B $C2 > $O/a_$rep.json 2> $O/a.err; val $O/a_$rep.json "tasks"
GPSLC_TASKS=0 timeout -k 10 200 $B $C2 > $O/b_$rep.json 2> $O/b.err; val $O/b_$rep.json "per-column"
done
GPSLC_TASK_DBG=2 timeout -k 10 200 $B $C2 > $O/dbg.json 2> $O/dbg.err
python3 tools/task_stamps.py gpurun_out/task_dbg.bin > $O/stamps.md; head -9 $O/stamps.md; tail -1 $O/stamps.md
timeout -k 10 200 $B $C2 --samples-per-step 1000 --steps 5 > $O/l_a.json 2> $O/a.err; val $O/l_a.json "S=1000 tasks"
GPSLC_TASKS=0 timeout -k 10 200 $B $C2 --samples-per-step 1000 --steps 5 > $O/l_b.json 2> $O/a.err; val $O/l_b.json "S=1000 per-column"
