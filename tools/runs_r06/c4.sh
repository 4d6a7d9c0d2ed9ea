#!/bin/bash
# round 6, call 4: task kernel with the augmented tile merged into strip(k+1,k), prefetched tickets, diagonal-task priority
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c4; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_tasks.py tests/test_gpu_estimation.py tests/test_gpu_fuzz.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile"
C2="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value']))"; }
for rep in 1 2; do
timeout -k 10 200 $B $C2 > $O/p3_$rep.json 2> $O/p3.err; val $O/p3_$rep.json "tasks, prio 3"
GPSLC_TASK_FENCE=0 timeout -k 10 200 $B $C2 > $O/p0_$rep.json 2> $O/p0.err; val $O/p0_$rep.json "tasks, prio 0"
GPSLC_TASK_FENCE=16 timeout -k 10 200 $B $C2 > $O/p1_$rep.json 2> $O/p1.err; val $O/p1_$rep.json "tasks, prio 1"
GPSLC_TASKS=0 timeout -k 10 200 $B $C2 > $O/off_$rep.json 2> $O/off.err; val $O/off_$rep.json "per-column launches"
done
GPSLC_TASK_G=4 timeout -k 10 200 $B $C2 > $O/g4.json 2> $O/g4.err; val $O/g4.json "tasks, group 4"
GPSLC_TASK_DBG=2 timeout -k 10 200 $B $C2 > $O/dbg.json 2> $O/dbg.err; val $O/dbg.json "stamped run"
python3 tools/task_stamps.py gpurun_out/task_dbg.bin > $O/stamps_n1024.md; cat $O/stamps_n1024.md
for tt in 8 0; do
timeout -k 10 200 $B --n 512 --d 4 --nu 1 --samples-per-step 16384 --task-tiles $tt > $O/n512_$tt.json 2> $O/n512.err; val $O/n512_$tt.json "n512 tiles $tt"
timeout -k 10 200 $B --n 768 --d 4 --nu 1 --samples-per-step 8192 --task-tiles $tt > $O/n768_$tt.json 2> $O/n768.err; val $O/n768_$tt.json "n768 tiles $tt"
timeout -k 10 200 $B $C2 --samples-per-step 1000 --steps 5 --task-tiles $tt > $O/c2l_$tt.json 2> $O/c2l.err; val $O/c2l_$tt.json "n1024 S=1000 tiles $tt"
done
