#!/bin/bash
# round 5, call 14: diagonal-block factorisation with the single-sweep factor + inverse and paired block updates — full parity suite,
# phase stamps, throughput at N = 512 / 1024 / 4096
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c14.log
: > $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee -a $O
timeout -k 10 300 python tools/potrf_stamps.py 1024 8192 2>&1 | tail -40 | tee -a $O
run() { timeout -k 10 300 python bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for rep in 1 2; do
  echo "== N=1024 / 512 / 4096 (run $rep)" | tee -a $O
  run --n 1024 --d 4 --nu 1 --samples-per-step 8192 | tee -a $O
  run --n 512 --d 4 --nu 1 --samples-per-step 16384 | tee -a $O
  run | tee -a $O
done
