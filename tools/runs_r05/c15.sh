#!/bin/bash
# round 5, call 15: same-box A/B of the diagonal-block factorisation: round-4 loop (prev build, GPSLC_POTRF_LA=0), lookahead only
# (prev build, GPSLC_POTRF_LA=1), current (single-sweep factor + inverse, paired updates, flat output stores)
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c15.log
: > $O
PREV=causalgpslc.jl_amd/csrc/libgpslc_hip_var_prev.so
run() { timeout -k 10 300 python bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for rep in 1 2; do
  for variant in "r4" "la1" "cur"; do
    echo "== $variant (run $rep): N=1024 / 512 / 2048 / 4096" | tee -a $O
    case $variant in
      r4)  export GPSLC_POTRF_LA=0; L="--lib $PREV" ;;
      la1) export GPSLC_POTRF_LA=1; L="--lib $PREV" ;;
      cur) unset GPSLC_POTRF_LA; L="" ;;
    esac
    run $L --n 1024 --d 4 --nu 1 --samples-per-step 8192 | tee -a $O
    run $L --n 512 --d 4 --nu 1 --samples-per-step 16384 | tee -a $O
    run $L --n 2048 --d 8 --nu 2 --samples-per-step 4096 | tee -a $O
    run $L | tee -a $O
  done
done
