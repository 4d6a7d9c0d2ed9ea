#!/bin/bash
# round 6, call 15: write-through (sc1) payload stores + drained flag instead of plain stores + release fence (measurement build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c15; mkdir -p $O
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile"
C2="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value']))"; }
for rep in 1 2 3; do
timeout -k 10 200 $B $C2 > $O/a_$rep.json 2> $O/a.err; val $O/a_$rep.json "plain stores + release fence"
GPSLC_TASK_FENCE=50 timeout -k 10 200 $B $C2 > $O/b_$rep.json 2> $O/b.err; val $O/b_$rep.json "write-through stores"
done
GPSLC_TASK_FENCE=50 GPSLC_TASK_DBG=2 timeout -k 10 200 $B $C2 > $O/dbg.json 2> $O/dbg.err
python3 tools/task_stamps.py gpurun_out/task_dbg.bin > $O/stamps_wt.md; head -12 $O/stamps_wt.md; tail -1 $O/stamps_wt.md
for n in 640 768; do
timeout -k 10 200 $B --n $n --d 4 --nu 1 --samples-per-step 8192 > $O/n${n}_a.json 2> $O/a.err; val $O/n${n}_a.json "n$n fence"
GPSLC_TASK_FENCE=50 timeout -k 10 200 $B --n $n --d 4 --nu 1 --samples-per-step 8192 > $O/n${n}_b.json 2> $O/a.err; val $O/n${n}_b.json "n$n write-through"
done
