#!/bin/bash
# round 6, call 38: the task launch with more than 32 right-hand sides (augmented row as ordinary strips): bit-identity tests,
# then 64 levels at N = 4096 / 1024 against the panel / per-column schedule
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c38; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_tasks.py -m gpu -x -q > $O/tasks.log 2>&1; echo "tasks rc=$?"; tail -3 $O/tasks.log
B="python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for rep in 1 2; do
timeout -k 10 300 $B --levels 64 > $O/l64_t$rep.json 2> $O/err.txt; val $O/l64_t$rep.json "N=4096 L=64 tasks"
timeout -k 10 300 $B --levels 64 --task-tiles 0 > $O/l64_p$rep.json 2> $O/err.txt; val $O/l64_p$rep.json "N=4096 L=64 panel schedule"
done
timeout -k 10 300 $B --levels 64 --no-mean-ite > $O/l64s_t.json 2> $O/err.txt; val $O/l64s_t.json "N=4096 L=64 SATE only, tasks"
timeout -k 10 300 $B --levels 64 --no-mean-ite --task-tiles 0 > $O/l64s_p.json 2> $O/err.txt; val $O/l64s_p.json "N=4096 L=64 SATE only, panel schedule"
timeout -k 10 300 $B --n 1024 --d 4 --nu 1 --samples-per-step 4096 --levels 101 > $O/n1024_l101_t.json 2> $O/err.txt; val $O/n1024_l101_t.json "N=1024 L=101 tasks"
timeout -k 10 300 $B --n 1024 --d 4 --nu 1 --samples-per-step 4096 --levels 101 --task-tiles 0 > $O/n1024_l101_p.json 2> $O/err.txt; val $O/n1024_l101_p.json "N=1024 L=101 per column"
timeout -k 10 300 $B --n 512 --d 4 --nu 1 --samples-per-step 8192 --levels 101 > $O/n512_l101_t.json 2> $O/err.txt; val $O/n512_l101_t.json "N=512 L=101 tasks"
timeout -k 10 300 $B --n 512 --d 4 --nu 1 --samples-per-step 8192 --levels 101 --task-tiles 0 > $O/n512_l101_p.json 2> $O/err.txt; val $O/n512_l101_p.json "N=512 L=101 per column"
