#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_11
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_estimation.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_model_nodes.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench$i.json 2> $OUT/bench$i.err
python - $OUT/bench$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]; s=r["second_kernel"]; u=d.get("units",{})
print(f"{d['value']:8.1f} samples/s  k0 {r['achieved']:.2f} TF/s share {r['share_of_step_time']:.3f} | fused {s['achieved']:.2f} TF/s ({s['frac']:.3f}) share {s['share_of_step_time']:.3f} | B {u.get('B',{}).get('value',0):.1f}/s", flush=True)
PY
done
