#!/bin/bash
# round 5, call 20: panel width 8 / 12 / 16 with the round-5 kernels at N = 2048 / 4096 / 8192 / 16384 (two alternating runs)
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c20.log
: > $O
run() { timeout -k 10 400 python bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value'],1), r['kernel'][:22], round(r['frac'],3), round(r.get('second_kernel',{}).get('frac',0),3), round(r['share_of_step_time'],3), round(r.get('second_kernel',{}).get('share_of_step_time',0),3))"; }
for rep in 1 2; do
for pw in 8 12 16; do
  echo "== panel $pw (run $rep): N=4096 / 2048 / 8192" | tee -a $O
  run --panel $pw | tee -a $O
  run --panel $pw --n 2048 --d 8 --nu 2 --samples-per-step 4096 | tee -a $O
  run --panel $pw --n 8192 --d 8 --nu 2 --samples-per-step 256 --steps 2 | tee -a $O
done
done
echo "== N=16384 binary fp32-kernel, panel 8 / 16" | tee -a $O
run --panel 8 --n 16384 --d 16 --nu 4 --samples-per-step 64 --binary-t --fp32-kernel --steps 1 | tee -a $O
run --panel 16 --n 16384 --d 16 --nu 4 --samples-per-step 64 --binary-t --fp32-kernel --steps 1 | tee -a $O
