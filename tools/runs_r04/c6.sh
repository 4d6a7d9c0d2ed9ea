#!/bin/bash
# round 4, call 6: full GPU suite on the production library (strip kernel refactored, bench rehearsal test, config-5 golden),
# then HBM traffic per kernel at BASELINE config 2 (N = 1024): two PMC passes of the c2 bench command
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
C2="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile --n 1024 --d 4 --nu 1 --samples-per-step 8192"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/fetch -- $C2 > $OUT/pmc_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc/write -- $C2 > $OUT/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
for k in tile_fused_strip_kernel tile_syrk_diag_kernel gram_kernel diag_potrf_inv_v2_kernel ite_mean_kernel backsolve_update_kernel "tile_gemm_nt_kernel<0, 0>" rhs_tiles_kernel "tile_gemm_nt_kernel<1, 0>" backsolve_alpha_kernel; do
  python3 tools/pmc_summary.py $OUT/pmc "$k" | tail -3 | sed "s/^/[$k] /" | tee -a $OUT/pmc_c2.txt
done
rm -rf $OUT/pmc
