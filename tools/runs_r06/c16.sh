#!/bin/bash
# round 6, call 16: write-through hand-off as the production path — task tests, the tile-count range again (N = 512 .. 1024)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c16; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_tasks.py tests/test_gpu_fuzz.py tests/test_gpu_estimation.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
B="python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value']))"; }
for n in 384 512 640 1024; do
S=8192; [ $n -le 512 ] && S=16384
for rep in 1 2; do
timeout -k 10 200 $B --n $n --d 4 --nu 1 --samples-per-step $S --task-tiles 8 --task-min-tiles 2 > $O/n${n}_t_$rep.json 2> $O/t.err; val $O/n${n}_t_$rep.json "n$n tasks"
timeout -k 10 200 $B --n $n --d 4 --nu 1 --samples-per-step $S --task-tiles 0 > $O/n${n}_c_$rep.json 2> $O/c.err; val $O/n${n}_c_$rep.json "n$n per-column"
done; done
