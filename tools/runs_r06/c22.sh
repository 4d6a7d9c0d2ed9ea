#!/bin/bash
# round 6, call 22: the task order's group size against the batch (S = 1000 in one call = BASELINE config 2 as stated; 8,192), and
# two streams of half chunks (does the next chunk's Gram build fill the drain of the task launch?)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c22; mkdir -p $O
B="python3 bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile --n 1024 --d 4 --nu 1"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value']), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for G in 2 4 8 12 16 24 32 48 64 128; do
timeout -k 10 200 $B --samples-per-step 1000 --task-group $G > $O/s1000_g$G.json 2> $O/err.txt; val $O/s1000_g$G.json "S=1000 group=$G"
done
for G in 8 16 32 64; do
timeout -k 10 200 $B --samples-per-step 8192 --steps 5 --task-group $G > $O/s8192_g$G.json 2> $O/err.txt; val $O/s8192_g$G.json "S=8192 group=$G"
done
timeout -k 10 200 $B --samples-per-step 1000 --max-batch 500 --streams 2 > $O/s1000_2x500.json 2> $O/err.txt; val $O/s1000_2x500.json "S=1000 two streams x 500"
timeout -k 10 200 $B --samples-per-step 1000 --max-batch 500 > $O/s1000_1x500.json 2> $O/err.txt; val $O/s1000_1x500.json "S=1000 one stream, chunks of 500"
timeout -k 10 200 $B --samples-per-step 8192 --steps 5 --max-batch 4096 --streams 2 > $O/s8192_2x4096.json 2> $O/err.txt; val $O/s8192_2x4096.json "S=8192 two streams x 4096"
timeout -k 10 200 $B --samples-per-step 8192 --steps 5 --max-batch 2048 --streams 2 > $O/s8192_2x2048.json 2> $O/err.txt; val $O/s8192_2x2048.json "S=8192 two streams x 2048"
timeout -k 10 200 $B --samples-per-step 8192 --steps 5 --max-batch 2048 --streams 4 > $O/s8192_4x2048.json 2> $O/err.txt; val $O/s8192_4x2048.json "S=8192 four streams x 2048"
