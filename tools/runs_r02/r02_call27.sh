#!/bin/bash
# in-kernel stamps of fused in-panel launches (K = 1, 4, 7 tiles) — where an item's time goes
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_27
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for K in 1 4 7; do
  GPSLC_GEMM_DBG_FUSEK=$K timeout -k 10 200 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-units --diag-lib --samples-per-step 256 > $OUT/b$K.json 2> $OUT/b$K.err || { echo "K=$K failed"; tail -3 $OUT/b$K.err; }
  echo "== fused launch with a K loop of $K tiles"; python3 tools/gemm_stamps.py gpurun_out/gemm_dbg.bin; cp gpurun_out/gemm_dbg.bin $OUT/dbg_$K.bin
done
rm -f $OUT/dbg_*.bin
