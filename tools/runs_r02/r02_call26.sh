#!/bin/bash
# A/B: trailing-update kernel with global_load_lds staging (measurement build switch GPSLC_GEMM_GLDS)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_26
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
GPSLC_GEMM_GLDS=1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --cpu-units 1 --no-units --diag-lib > $OUT/bench_parity.json 2> $OUT/bench_parity.err || { echo "parity run failed"; tail -5 $OUT/bench_parity.err; exit 1; }
python3 -c "
import json
d=json.loads(open('$OUT/bench_parity.json').read().strip().splitlines()[-1])
print('glds parity', d['sate_rel_err'])"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --diag-lib"
for P in 0 1 0 1; do
  GPSLC_GEMM_GLDS=$P timeout -k 10 200 $B > $OUT/bench_g$P.json 2> $OUT/bench_g$P.err || { echo "P=$P failed"; tail -5 $OUT/bench_g$P.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$OUT/bench_g$P.json').read().strip().splitlines()[-1])
r=d['roofline']
print('glds=$P', round(d['value'],1), 'trailing', round(r['achieved'],2), r['kernel'][:30], 'fused', round(r['second_kernel']['achieved'],2))"
done
