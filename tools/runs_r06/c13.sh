#!/bin/bash
# round 6, call 13: release fence per wave before the barrier (measurement build) against one lane after it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c13; mkdir -p $O
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile"
C2="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value']))"; }
for rep in 1 2 3; do
timeout -k 10 200 $B $C2 > $O/a_$rep.json 2> $O/a.err; val $O/a_$rep.json "one lane after the barrier"
GPSLC_TASK_FENCE=50 timeout -k 10 200 $B $C2 > $O/b_$rep.json 2> $O/b.err; val $O/b_$rep.json "every wave before the barrier"
done
