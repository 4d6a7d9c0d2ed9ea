#!/bin/bash
# round 6, call 43: what the MeanITE pass costs in a 101-level sweep (the reference's default fidelity: src/prediction.jl:24-28) —
# two passes of 64 + 37 levels, each re-evaluating B — with / without MeanITE at N = 1024 and N = 4096
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c43; mkdir -p $O
B="python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --no-profile"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for L in 64 65 101 128; do
timeout -k 10 300 $B --n 1024 --d 4 --nu 1 --samples-per-step 4096 --levels $L > $O/n1024_l$L.json 2> $O/err.txt; val $O/n1024_l$L.json "N=1024 L=$L with MeanITE"
timeout -k 10 300 $B --n 1024 --d 4 --nu 1 --samples-per-step 4096 --levels $L --no-mean-ite > $O/n1024_l${L}s.json 2> $O/err.txt; val $O/n1024_l${L}s.json "N=1024 L=$L SATE only"
done
for L in 64 101; do
timeout -k 10 300 $B --levels $L > $O/n4096_l$L.json 2> $O/err.txt; val $O/n4096_l$L.json "N=4096 L=$L with MeanITE"
timeout -k 10 300 $B --levels $L --no-mean-ite > $O/n4096_l${L}s.json 2> $O/err.txt; val $O/n4096_l${L}s.json "N=4096 L=$L SATE only"
done
