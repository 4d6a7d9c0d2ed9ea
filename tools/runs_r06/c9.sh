#!/bin/bash
# round 6, call 9: symmetric MeanITE pass (every pair once) — parity suites, then the A/B at N = 4096 and config 2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c9; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_estimation.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile"
C2="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1))"; }
for rep in 1 2; do
timeout -k 10 200 $B $C2 > $O/c2_sym_$rep.json 2> $O/c2.err; val $O/c2_sym_$rep.json "n1024 symmetric"
GPSLC_ITE_SYM=0 timeout -k 10 200 $B $C2 > $O/c2_old_$rep.json 2> $O/c2.err; val $O/c2_old_$rep.json "n1024 every pair twice"
timeout -k 10 300 $B > $O/n4096_sym_$rep.json 2> $O/n4096.err; val $O/n4096_sym_$rep.json "n4096 symmetric"
GPSLC_ITE_SYM=0 timeout -k 10 300 $B > $O/n4096_old_$rep.json 2> $O/n4096.err; val $O/n4096_old_$rep.json "n4096 every pair twice"
done
