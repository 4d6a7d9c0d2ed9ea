#!/bin/bash
# round 6, call 36: chunk size of the task launch at N = 4096 (ramp / tail of the wavefront against the chunk): 512 / 1024 / 1536 / 2048
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c36; mkdir -p $O
B="python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --no-profile --samples-per-step 6144"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for mb in 512 1024 1536 2048 3072; do
timeout -k 10 400 $B --max-batch $mb > $O/mb$mb.json 2> $O/err_$mb.txt; val $O/mb$mb.json "N=4096 S=6144 chunks of $mb"
done
