#!/bin/bash
# round 6, call 14: diag(k) and strip(k+1, k) + augmented tile as ONE task (one fetch / acquire / release fewer per column)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c14; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_tasks.py tests/test_gpu_fuzz.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile"
C2="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value']))"; }
for rep in 1 2 3; do
timeout -k 10 200 $B $C2 > $O/a_$rep.json 2> $O/a.err; val $O/a_$rep.json "merged"
GPSLC_TASK_MERGE=0 timeout -k 10 200 $B $C2 > $O/b_$rep.json 2> $O/b.err; val $O/b_$rep.json "separate"
done
timeout -k 10 200 $B --n 640 --d 4 --nu 1 --samples-per-step 8192 > $O/n640_a.json 2> $O/a.err; val $O/n640_a.json "n640 merged"
GPSLC_TASK_MERGE=0 timeout -k 10 200 $B --n 640 --d 4 --nu 1 --samples-per-step 8192 > $O/n640_b.json 2> $O/b.err; val $O/n640_b.json "n640 separate"
GPSLC_TASKS=0 timeout -k 10 200 $B --n 640 --d 4 --nu 1 --samples-per-step 8192 > $O/n640_c.json 2> $O/b.err; val $O/n640_c.json "n640 per-column"
