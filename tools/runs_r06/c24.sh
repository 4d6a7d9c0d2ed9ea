#!/bin/bash
# round 6, call 24: default group 32 — the tile-count range re-swept (N = 384 below, N = 1280 .. 2048 above as one left-looking
# panel of tasks against the production schedule: panel 8 + trailing updates)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c24; mkdir -p $O
B="python3 bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value']), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
timeout -k 10 200 $B --n 384 --d 4 --nu 1 --samples-per-step 16384 > $O/n384_pc.json 2> $O/err.txt; val $O/n384_pc.json "N=384 default (per column)"
timeout -k 10 200 $B --n 384 --d 4 --nu 1 --samples-per-step 16384 --task-min-tiles 3 > $O/n384_t.json 2> $O/err.txt; val $O/n384_t.json "N=384 tasks"
timeout -k 10 200 $B --n 256 --d 4 --nu 1 --samples-per-step 16384 > $O/n256_pc.json 2> $O/err.txt; val $O/n256_pc.json "N=256 default (per column)"
timeout -k 10 200 $B --n 256 --d 4 --nu 1 --samples-per-step 16384 --task-min-tiles 2 > $O/n256_t.json 2> $O/err.txt; val $O/n256_t.json "N=256 tasks"
for n in 1280 1536 2048; do
nt=$(( n / 128 )); S=$(( 4096 * 2048 * 2048 / n / n ))
timeout -k 10 300 $B --n $n --d 8 --nu 2 --samples-per-step $S > $O/n${n}_def.json 2> $O/err.txt; val $O/n${n}_def.json "N=$n S=$S default"
timeout -k 10 300 $B --n $n --d 8 --nu 2 --samples-per-step $S --panel $nt --task-tiles $nt > $O/n${n}_t.json 2> $O/err.txt; val $O/n${n}_t.json "N=$n S=$S one panel of tasks"
timeout -k 10 300 $B --n $n --d 8 --nu 2 --samples-per-step $S --panel $nt --task-tiles 0 > $O/n${n}_p.json 2> $O/err.txt; val $O/n${n}_p.json "N=$n S=$S one panel, per column"
done
for r in 1 2 3; do
GPSLC_TASK_ROWS=$r timeout -k 10 200 $B --diag-lib --n 1024 --d 4 --nu 1 --samples-per-step 8192 > $O/rows$r.json 2> $O/err.txt; val $O/rows$r.json "N=1024 rows per strip task=$r (measurement build)"
done
timeout -k 10 200 $B --n 1024 --d 4 --nu 1 --samples-per-step 8192 > $O/n1024.json 2> $O/err.txt; val $O/n1024.json "N=1024 S=8192 default"
timeout -k 10 200 $B --n 1024 --d 4 --nu 1 --samples-per-step 1000 --steps 20 > $O/n1024l.json 2> $O/err.txt; val $O/n1024l.json "N=1024 S=1000 default"
