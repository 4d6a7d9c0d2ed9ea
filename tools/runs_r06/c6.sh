#!/bin/bash
# round 6, call 6: several tile rows per strip task; N = 2048 as one left-looking panel of tasks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c6; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_tasks.py tests/test_gpu_fuzz.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile"
C2="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value']))"; }
for rep in 1 2; do
for r in 1 2 3; do
GPSLC_TASK_ROWS=$r timeout -k 10 200 $B $C2 > $O/r${r}_$rep.json 2> $O/r.err; val $O/r${r}_$rep.json "tasks rows=$r"
done
GPSLC_TASKS=0 timeout -k 10 200 $B $C2 > $O/off_$rep.json 2> $O/off.err; val $O/off_$rep.json "per-column launches"
done
for r in 1 2 3; do
GPSLC_TASK_ROWS=$r timeout -k 10 200 $B --n 512 --d 4 --nu 1 --samples-per-step 16384 > $O/n512_r$r.json 2> $O/n512.err; val $O/n512_r$r.json "n512 rows=$r"
done
GPSLC_TASKS=0 timeout -k 10 200 $B --n 512 --d 4 --nu 1 --samples-per-step 16384 > $O/n512_off.json 2> $O/n512.err; val $O/n512_off.json "n512 per-column"
N2="--n 2048 --d 8 --nu 2 --samples-per-step 4096"
timeout -k 10 300 $B $N2 > $O/n2048_base.json 2> $O/n2048.err; val $O/n2048_base.json "n2048 panel 8 (production)"
timeout -k 10 300 $B $N2 --panel 16 --task-tiles 0 > $O/n2048_p16.json 2> $O/n2048.err; val $O/n2048_p16.json "n2048 panel 16 per-column"
timeout -k 10 300 $B $N2 --panel 16 --task-tiles 16 > $O/n2048_t16.json 2> $O/n2048.err; val $O/n2048_t16.json "n2048 panel 16 tasks"
GPSLC_TASK_ROWS=3 timeout -k 10 300 $B $N2 --panel 16 --task-tiles 16 > $O/n2048_t16r3.json 2> $O/n2048.err; val $O/n2048_t16r3.json "n2048 panel 16 tasks rows 3"
