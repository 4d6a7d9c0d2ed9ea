#!/bin/bash
# round 6, call 3: where a task's time goes (measurement build, per-task stamps), release-fence cost, group size
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c3; mkdir -p $O
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile"
C2="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value']))"; }
GPSLC_TASK_DBG=2 timeout -k 10 200 $B $C2 > $O/dbg.json 2> $O/dbg.err; val $O/dbg.json "stamped run"
python3 tools/task_stamps.py gpurun_out/task_dbg.bin > $O/stamps_n1024.md; cat $O/stamps_n1024.md
for rep in 1 2; do
timeout -k 10 200 $B $C2 > $O/base_$rep.json 2> $O/base.err; val $O/base_$rep.json "tasks, release fence"
GPSLC_TASK_FENCE=1 timeout -k 10 200 $B $C2 > $O/nofence_$rep.json 2> $O/nofence.err; val $O/nofence_$rep.json "tasks, no release fence (timing)"
GPSLC_TASKS=0 timeout -k 10 200 $B $C2 > $O/off_$rep.json 2> $O/off.err; val $O/off_$rep.json "per-column launches"
done
for g in 2 4 16 32; do
GPSLC_TASK_G=$g timeout -k 10 200 $B $C2 > $O/g$g.json 2> $O/g$g.err; val $O/g$g.json "tasks, group $g"
done
GPSLC_TASK_DBG=2 timeout -k 10 200 $B --n 512 --d 4 --nu 1 --samples-per-step 16384 > $O/dbg512.json 2> $O/dbg512.err; val $O/dbg512.json "stamped run n512"
python3 tools/task_stamps.py gpurun_out/task_dbg.bin > $O/stamps_n512.md; cat $O/stamps_n512.md
