#!/bin/bash
# round 6, call 23: group size 32 against 8 at the other sizes and batch sizes the task launch serves (and against the
# per-column launches below the 256-matrix threshold)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c23; mkdir -p $O
B="python3 bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile --d 4 --nu 1"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value']), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for n in 512 640 768 896; do
for G in 8 32 64; do
S=$(( 8192 * 1024 * 1024 / n / n )); [ $S -gt 16384 ] && S=16384
timeout -k 10 200 $B --n $n --samples-per-step $S --steps 5 --task-group $G > $O/n${n}_g$G.json 2> $O/err.txt; val $O/n${n}_g$G.json "N=$n S=$S group=$G"
done; done
for S in 64 128 256 512 2000 4000; do
for G in 8 32; do
timeout -k 10 200 $B --n 1024 --samples-per-step $S --task-group $G --task-min-matrices 1 > $O/s${S}_g$G.json 2> $O/err.txt; val $O/s${S}_g$G.json "N=1024 S=$S group=$G"
done
timeout -k 10 200 $B --n 1024 --samples-per-step $S --task-tiles 0 > $O/s${S}_pc.json 2> $O/err.txt; val $O/s${S}_pc.json "N=1024 S=$S per-column"
done
