#!/bin/bash
# A/B of scheduling variants of the dominant kernel on one box (measurement build)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_5
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units"
run() { tag=$1; shift; env "$@" timeout -k 10 200 $B $EXTRA > $OUT/$tag.json 2> $OUT/$tag.err; python - "$OUT/$tag.json" "$tag" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; print(f"{sys.argv[2]:28s} {d['value']:8.1f} samples/s  k0 {r['achieved']:.2f} TF/s ({r['avg_launch_ms']:.2f} ms)  k1 {r['second_kernel']['achieved']:.2f} TF/s", flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
}
EXTRA="" run prod X=1
EXTRA="--diag-lib" run diag_default X=1
EXTRA="--diag-lib" run nt_c GPSLC_NT_C=1
EXTRA="--diag-lib" run order4 GPSLC_ORDER_BLOCK=4
EXTRA="--diag-lib" run order16 GPSLC_ORDER_BLOCK=16
EXTRA="--diag-lib" run order1 GPSLC_ORDER_BLOCK=1
EXTRA="--diag-lib" run ntc_order4 GPSLC_NT_C=1 GPSLC_ORDER_BLOCK=4
EXTRA="--panel 4" run panel4 X=1
EXTRA="--panel 16" run panel16 X=1
EXTRA="--streams 2" run streams2 X=1
EXTRA="" run prod_again X=1
