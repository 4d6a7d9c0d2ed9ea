#!/bin/bash
# round 5, call 3: (a) panel width at N = 1024 (nt = 8: until now always ONE panel), (b) unit B against its (sample, level) sub-batch size
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c3.log
: > $O
C2="python bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --n 1024 --d 4 --nu 1 --samples-per-step 8192"
for rep in 1 2; do
for pw in 8 4 2 3; do
  echo "== N=1024 panel $pw (run $rep)" | tee -a $O
  timeout -k 10 300 $C2 --panel $pw 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel'][:24], d['roofline']['frac'], d['roofline'].get('second_kernel',{}).get('frac'))" | tee -a $O
done
done
echo "== N=2048 panel 8 / 4" | tee -a $O
C3="python bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --n 2048 --d 8 --nu 2 --samples-per-step 4096"
for pw in 8 4; do
  timeout -k 10 300 $C3 --panel $pw 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline'].get('second_kernel',{}).get('frac'))" | tee -a $O
done
echo "== unit B vs sub-batch (N=4096, 8 samples x 32 levels = 256 units, 10 draws each)" | tee -a $O
for bb in 16 32 64 128 256; do
  echo "-- GPSLC_UNITB_BATCH=$bb" | tee -a $O
  GPSLC_UNITB_BATCH=$bb timeout -k 10 300 python tools/bench_unit_b.py --diag-lib 4096 8 32 10 2>&1 | tail -1 | tee -a $O
done
