#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats + PMC passes of the default bench command (round 2).
# Usage: bash tools/profile_r02.sh <tag>
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
# (1) the bench command itself with per-kernel durations: unit A (timed region) + units B and C
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 &&
# (2) PMC passes, one counter group per run, unit A only
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --no-units --no-profile > $OUT/pmc_fetch.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $B --no-units --no-profile > $OUT/pmc_write.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- $B --no-units --no-profile > $OUT/pmc_sq.log 2>&1
grep '"metric"' $OUT/trace.log | cut -c1-400
cd $GRAFT_REPO_ROOT
python3 tools/kernel_stats_md.py $OUT/trace "rocprofv3 --kernel-trace --stats of \`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline\` (4 x 1024 posterior samples of unit A at N=4096, then units B and C: 2 x 64 units with 10 draws)" 4096 > $OUT/kernel_stats.md
python3 tools/pmc_summary.py $OUT "tile_gemm_nt_kernel<1, 0, 0>" $OUT/pmc_tile_gemm.json > $OUT/pmc_tile_gemm.md
python3 tools/pmc_summary.py $OUT "tile_gemm_nt_kernel<1, 0, 1>" $OUT/pmc_fused.json > $OUT/pmc_fused.md
python3 tools/pmc_summary.py $OUT "draws_mfma_kernel" $OUT/pmc_draws.json > $OUT/pmc_draws.md || true
head -30 $OUT/kernel_stats.md; cat $OUT/pmc_tile_gemm.md | tail -8
