#!/bin/bash
# round 6, call 42: config 2 — two streams of 2,048-matrix chunks with the task launch leaving workgroup slots free (measurement
# build, GPSLC_GEMM_SLOTS): at N = 1024 the task launch keeps the MFMA pipe only 0.63 busy; can the neighbouring chunk's Gram build
# and MeanITE pass use the rest?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c42; mkdir -p $O
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --no-profile --n 1024 --d 4 --nu 1 --samples-per-step 8192"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
timeout -k 10 300 $B > $O/base.json 2> $O/err.txt; val $O/base.json "one stream, one chunk of 8192"
timeout -k 10 300 $B --max-batch 2048 > $O/b2048.json 2> $O/err.txt; val $O/b2048.json "one stream, chunks of 2048"
for sl in 512 480 448 416 384 320; do
GPSLC_GEMM_SLOTS=$sl timeout -k 10 300 $B --max-batch 2048 --streams 2 > $O/s2_$sl.json 2> $O/err.txt; val $O/s2_$sl.json "two streams, chunks of 2048, $sl slots"
done
for sl in 448 384; do
GPSLC_GEMM_SLOTS=$sl timeout -k 10 300 $B --max-batch 1024 --streams 4 > $O/s4_$sl.json 2> $O/err.txt; val $O/s4_$sl.json "four streams, chunks of 1024, $sl slots"
done
