#!/bin/bash
# round 5, call 2: full GPU suite after the source hygiene / epilogue / rhs changes, the default bench line, more draw-kernel variants
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c2.log
: > $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee -a $O
for v in 0 7 8 9 10 11 2; do
  echo "== GPSLC_DRAWS_VAR=$v" | tee -a $O
  GPSLC_DRAWS_VAR=$v timeout -k 10 300 python tools/bench_draws.py --diag-lib 4096 8 8 10 3 2>&1 | tail -1 | tee -a $O
done
echo "== default bench" | tee -a $O
timeout -k 10 600 python bench.py > gpurun_out/r05/c2_bench.json 2> gpurun_out/r05/c2_bench.err || { tail -20 gpurun_out/r05/c2_bench.err | tee -a $O; exit 1; }
python - <<'PY' | tee -a $O
import json
d=json.loads(open('gpurun_out/r05/c2_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'runs', d['value_runs'])
print('roofline', d['roofline']['frac'], d['roofline']['second_kernel']['frac'], d['roofline']['traffic_note'][:60])
print('C', d['units']['C']['achieved'], d['units']['C']['frac'], 'B', d['units']['B']['frac'], d['units']['B']['single_level']['frac'])
for k,v in d['configs'].items(): print(k, v['value'], v['parity'])
print('c4', d['config4']['value'], d['config4']['parity'])
print('A', d['units']['A'])
PY
