#!/bin/bash
# round 6, call 25: one left-looking panel of tasks above N = 2048 (nt = 20, 24: the descriptor's limit), group 32
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c25; mkdir -p $O
B="python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value']), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for n in 2560 3072; do
nt=$(( n / 128 )); S=$(( 4096 * 2048 * 2048 / n / n ))
timeout -k 10 300 $B --n $n --d 8 --nu 2 --samples-per-step $S > $O/n${n}_def.json 2> $O/err.txt; val $O/n${n}_def.json "N=$n S=$S default"
timeout -k 10 300 $B --n $n --d 8 --nu 2 --samples-per-step $S --panel $nt --task-tiles $nt > $O/n${n}_t.json 2> $O/err.txt; val $O/n${n}_t.json "N=$n S=$S one panel of tasks"
timeout -k 10 300 $B --n $n --d 8 --nu 2 --samples-per-step $S --panel 12 > $O/n${n}_p12.json 2> $O/err.txt; val $O/n${n}_p12.json "N=$n S=$S panel 12"
done
