#!/bin/bash
# round 6, call 7: task launch with 2 tile rows per strip task — N = 896 / 768, S = 1000 single chunk, group size, 4 rows
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c7; mkdir -p $O
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile"
C2="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value']))"; }
for n in 896 768 640; do
timeout -k 10 200 $B --n $n --d 4 --nu 1 --samples-per-step 8192 > $O/n${n}_t.json 2> $O/n.err; val $O/n${n}_t.json "n$n tasks"
GPSLC_TASKS=0 timeout -k 10 200 $B --n $n --d 4 --nu 1 --samples-per-step 8192 > $O/n${n}_off.json 2> $O/n.err; val $O/n${n}_off.json "n$n per-column"
done
for rep in 1 2; do
timeout -k 10 200 $B $C2 --samples-per-step 1000 --steps 5 > $O/c2l_t_$rep.json 2> $O/c2l.err; val $O/c2l_t_$rep.json "S=1000 tasks"
GPSLC_TASKS=0 timeout -k 10 200 $B $C2 --samples-per-step 1000 --steps 5 > $O/c2l_off_$rep.json 2> $O/c2l.err; val $O/c2l_off_$rep.json "S=1000 per-column"
done
for g in 4 6 12 16; do
GPSLC_TASK_G=$g timeout -k 10 200 $B $C2 > $O/g$g.json 2> $O/g.err; val $O/g$g.json "tasks group $g"
done
GPSLC_TASK_ROWS=4 timeout -k 10 200 $B $C2 > $O/r4.json 2> $O/r.err; val $O/r4.json "tasks rows=4"
timeout -k 10 200 $B $C2 > $O/base.json 2> $O/r.err; val $O/base.json "tasks (G 8, rows 2)"
GPSLC_TASK_DBG=2 timeout -k 10 200 $B $C2 > $O/dbg.json 2> $O/dbg.err
python3 tools/task_stamps.py gpurun_out/task_dbg.bin > $O/stamps_n1024.md; cat $O/stamps_n1024.md
