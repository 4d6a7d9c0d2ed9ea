#!/bin/bash
# round 5, call 10: the driver-style bench line (profiles/r05_bench_default.json)
set -e
mkdir -p gpurun_out/r05
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err || { tail -20 gpurun_out/r05/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'runs', d['value_runs'])
print('roofline', d['roofline']['frac'], d['roofline']['second_kernel']['frac'], d['roofline']['traffic'], d['roofline']['traffic_note'][:50])
print('C', d['units']['C']['achieved'], d['units']['C']['frac'], 'B', d['units']['B']['value'], d['units']['B']['frac'], d['units']['B']['single_level']['frac'])
for k,v in d['configs'].items(): print(k, v['value'], v['parity']['ok'], v.get('hbm'))
print('c4', d['config4']['value'], d['config4']['parity']['ok'])
print('A', d['units']['A']['ceiling_shared_datapath_units_per_s'])
print('cpu', d['cpu_baseline']['value'], d['sate_rel_err'])
PY
