#!/bin/bash
# round 6, call 30: tile rows per strip task against the tile count (measurement build: GPSLC_TASK_ROWS), same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c30; mkdir -p $O
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --no-profile"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for n in 1280 1536 2048 3072 4096; do
S=$(( 4096 * 2048 * 2048 / n / n )); [ $n = 4096 ] && S=1024
for r in 1 2; do
GPSLC_TASK_ROWS=$r timeout -k 10 300 $B --n $n --d 8 --nu 2 --samples-per-step $S > $O/n${n}_r$r.json 2> $O/err.txt; val $O/n${n}_r$r.json "N=$n rows=$r"
done; done
for n in 512 768 1024; do
S=$(( 8192 * 1024 * 1024 / n / n )); [ $S -gt 16384 ] && S=16384
for r in 1 2; do
GPSLC_TASK_ROWS=$r timeout -k 10 300 $B --n $n --d 4 --nu 1 --samples-per-step $S > $O/n${n}_r$r.json 2> $O/err.txt; val $O/n${n}_r$r.json "N=$n rows=$r"
done; done
for g in 16 32 64; do
GPSLC_TASK_ROWS=1 timeout -k 10 300 $B --n 4096 --d 8 --nu 2 --samples-per-step 1024 --task-group $g > $O/n4096_g$g.json 2> $O/err.txt; val $O/n4096_g$g.json "N=4096 rows=1 group=$g"
done
