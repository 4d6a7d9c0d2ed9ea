#!/bin/bash
# round 6, call 31: two streams with the persistent task launch leaving workgroup slots free (measurement build:
# GPSLC_GEMM_SLOTS) — does the Gram build / MeanITE pass of the neighbouring chunk run in the free slots beside the task launch?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c31; mkdir -p $O
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --no-profile --samples-per-step 2048"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
timeout -k 10 300 $B > $O/base.json 2> $O/err.txt; val $O/base.json "N=4096 S=2048 default (chunks of 1024, one stream)"
timeout -k 10 300 $B --max-batch 512 > $O/b512.json 2> $O/err.txt; val $O/b512.json "chunks of 512, one stream"
for sl in 512 496 480 448 416; do
GPSLC_GEMM_SLOTS=$sl timeout -k 10 300 $B --max-batch 512 --streams 2 > $O/s2_$sl.json 2> $O/err.txt; val $O/s2_$sl.json "chunks of 512, two streams, $sl slots"
done
for sl in 496 480 448; do
GPSLC_GEMM_SLOTS=$sl timeout -k 10 300 $B --max-batch 512 --streams 3 > $O/s3_$sl.json 2> $O/err.txt; val $O/s3_$sl.json "chunks of 512, three streams, $sl slots"
done
GPSLC_GEMM_SLOTS=480 timeout -k 10 300 $B --max-batch 1024 --streams 2 > $O/s2b1024_480.json 2> $O/err.txt; val $O/s2b1024_480.json "chunks of 1024, two streams, 480 slots"
