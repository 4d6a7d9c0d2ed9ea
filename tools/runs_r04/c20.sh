#!/bin/bash
# round 4, call 20: in-kernel stamps of the strip kernel (measurement build) at K depth 1 and 4, N = 4096 batch 256 — the
# round-4 counterpart of profiles/r03_ab_experiments.md §1's table — and of one trailing update (mi = 24)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_20
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for k in 1 4; do
  GPSLC_GEMM_DBG_FUSEK=$k timeout -k 10 200 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-units --no-configs --no-config4 --diag-lib --samples-per-step 256 > $OUT/b.json 2> $OUT/b.err
  echo "== strip kernel, K depth $k" | tee -a $OUT/stamps.txt
  python3 tools/gemm_stamps.py | tee -a $OUT/stamps.txt
done
GPSLC_GEMM_DBG=24 timeout -k 10 200 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-units --no-configs --no-config4 --diag-lib --samples-per-step 256 > $OUT/b.json 2> $OUT/b.err
echo "== trailing update, mi = 24 (K depth 8)" | tee -a $OUT/stamps.txt
python3 tools/gemm_stamps.py | tee -a $OUT/stamps.txt
rm -f gpurun_out/gemm_dbg.bin
