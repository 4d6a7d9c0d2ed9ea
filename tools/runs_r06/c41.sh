#!/bin/bash
# round 6, call 41: Gram build at 5 / 6 waves per SIMD (__launch_bounds__(256, 5 | 6): 96 / 80 VGPRs with 15 / 43 spilled) against
# the production 127 VGPRs (4 waves), config 2 and N = 4096, same box, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c41; mkdir -p $O
B="python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --no-profile"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for rep in 1 2; do
timeout -k 10 300 $B --n 1024 --d 4 --nu 1 --samples-per-step 8192 > $O/c2_prod_$rep.json 2> $O/err.txt; val $O/c2_prod_$rep.json "config 2 production"
for mb in 5 6; do
timeout -k 10 300 $B --n 1024 --d 4 --nu 1 --samples-per-step 8192 --lib causalgpslc.jl_amd/csrc/libgpslc_hip_var_gram$mb.so > $O/c2_mb${mb}_$rep.json 2> $O/err.txt; val $O/c2_mb${mb}_$rep.json "config 2 gram at $mb waves"
done; done
timeout -k 10 300 $B > $O/n4096_prod.json 2> $O/err.txt; val $O/n4096_prod.json "N=4096 production"
timeout -k 10 300 $B --lib causalgpslc.jl_amd/csrc/libgpslc_hip_var_gram5.so > $O/n4096_mb5.json 2> $O/err.txt; val $O/n4096_mb5.json "N=4096 gram at 5 waves"
