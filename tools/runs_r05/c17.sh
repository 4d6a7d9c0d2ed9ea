#!/bin/bash
# round 5, call 17: item-boundary barriers of the persistent tile kernels order LDS traffic only (no wait for the output stores'
# acknowledgement) — same-box A/B against the previous build, parity suite
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c17.log
: > $O
PREV=causalgpslc.jl_amd/csrc/libgpslc_hip_var_prev.so
run() { timeout -k 10 300 python bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['frac'], r.get('second_kernel',{}).get('frac'))"; }
for rep in 1 2 3; do
  for variant in prev cur; do
    echo "== $variant (run $rep): N=4096 / 1024 / 2048" | tee -a $O
    if [ $variant = prev ]; then L="--lib $PREV"; else L=""; fi
    run $L | tee -a $O
    run $L --n 1024 --d 4 --nu 1 --samples-per-step 8192 | tee -a $O
    run $L --n 2048 --d 8 --nu 2 --samples-per-step 4096 | tee -a $O
  done
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee -a $O
