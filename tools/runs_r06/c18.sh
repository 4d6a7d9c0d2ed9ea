#!/bin/bash
# round 6, call 18: where the task launch starts to pay — matrices per call at N = 1024 and N = 640
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c18; mkdir -p $O
B="python3 bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value']), 'samples/s', round(d['ms_per_step'],3), 'ms per call')"; }
for n in 1024 640; do
for S in 8 32 64 128 256 512; do
timeout -k 10 200 $B --n $n --d 4 --nu 1 --samples-per-step $S > $O/t_${n}_$S.json 2> $O/t.err; val $O/t_${n}_$S.json "n$n S=$S tasks"
timeout -k 10 200 $B --n $n --d 4 --nu 1 --samples-per-step $S --task-tiles 0 > $O/c_${n}_$S.json 2> $O/c.err; val $O/c_${n}_$S.json "n$n S=$S per-column"
done; done
