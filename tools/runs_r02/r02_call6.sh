#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_6
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
GPSLC_SMALL_STAMPS=6 timeout -k 10 300 python tools/bench_latency.py --diag-lib > $OUT/lat.log 2>&1; grep -v amdgpu.ids $OUT/lat.log | head -20
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units"
for p in 8 12 16 24 32; do
  timeout -k 10 200 $B --panel $p > $OUT/panel$p.json 2> $OUT/panel$p.err
  python - $OUT/panel$p.json panel$p <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; s=r.get("second_kernel",{})
    print(f"{sys.argv[2]:12s} {d['value']:8.1f} samples/s  dominant {r['kernel'][:28]} {r['achieved']:.2f} TF/s share {r['share_of_step_time']:.2f} | second {s.get('achieved',0):.2f} TF/s share {s.get('share_of_step_time',0):.2f}", flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
done
