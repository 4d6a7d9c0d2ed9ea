#!/bin/bash
# round 6, call 32: s_setprio around the MFMAs of a slab in the strip tasks' K loop, from a K-loop length on (measurement build:
# GPSLC_STRIP_PRIO = slabs; 100000 = never, 64 = from 8 tile columns, 8 = always), same box, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c32; mkdir -p $O
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --no-profile"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for rep in 1 2; do
for p in 100000 128 64 32 8; do
GPSLC_STRIP_PRIO=$p timeout -k 10 300 $B > $O/n4096_p${p}_$rep.json 2> $O/err.txt; val $O/n4096_p${p}_$rep.json "N=4096 prio from $p slabs"
done; done
for p in 100000 64 32 8; do
GPSLC_STRIP_PRIO=$p timeout -k 10 300 $B --n 2048 --samples-per-step 4096 > $O/n2048_p$p.json 2> $O/err.txt; val $O/n2048_p$p.json "N=2048 prio from $p slabs"
done
for p in 100000 32 8; do
GPSLC_STRIP_PRIO=$p timeout -k 10 300 $B --n 1024 --d 4 --nu 1 --samples-per-step 8192 > $O/n1024_p$p.json 2> $O/err.txt; val $O/n1024_p$p.json "N=1024 prio from $p slabs"
done
