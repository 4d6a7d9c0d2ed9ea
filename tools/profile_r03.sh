#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats + PMC passes of the default bench command (round 3).
# Usage: bash tools/profile_r03.sh <tag>      -> gpurun_out/<tag>/ ; then tools/collect_profiles.py <tag> r03
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config4 --no-configs"
UB="python3 $GRAFT_REPO_ROOT/tools/bench_unit_b.py 4096 64 1 10"
C2="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config4 --no-configs --no-units --n 1024 --d 4 --nu 1 --samples-per-step 8192"
# (1) unit A: the bench's timed region, per-kernel durations (HIP-event profiling on, as in the driver's run)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --no-units > $OUT/trace.log 2>&1 &&
# (2) units B and C: 2 x 64 (sample, level) units with 10 draws each (warm-up call + timed call)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_b -- $UB > $OUT/trace_b.log 2>&1 &&
# (2b) BASELINE config 2 (N = 1024)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2 -- $C2 > $OUT/trace_c2.log 2>&1 &&
# (3) PMC passes, one counter group per run, unit A
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --no-units --no-profile > $OUT/pmc_fetch.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $B --no-units --no-profile > $OUT/pmc_write.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- $B --no-units --no-profile > $OUT/pmc_sq.log 2>&1 &&
# (3b) VALU issue counters (the two fp64-VALU kernels of unit A: Gram build, MeanITE pass)
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_valu -- $B --no-units --no-profile > $OUT/pmc_valu.log 2>&1 &&
# (4) PMC passes for the draws kernel (unit C): bytes of L_c actually fetched per launch
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/b_pmc_fetch -- $UB > $OUT/b_pmc_fetch.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/b_pmc_write -- $UB > $OUT/b_pmc_write.log 2>&1
grep '"metric"' $OUT/trace.log | cut -c1-300
tail -1 $OUT/trace_b.log
cd $GRAFT_REPO_ROOT
python3 tools/kernel_stats_md.py $OUT/trace "rocprofv3 --kernel-trace --stats of \`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config4 --no-configs --no-units\` (4 x 1024 posterior samples of unit A at N=4096 D=8 nU=2: 1 warm-up + 3 timed steps)" 4096 > $OUT/kernel_stats.md
python3 tools/kernel_stats_md.py $OUT/trace_b "rocprofv3 --kernel-trace --stats of \`python3 tools/bench_unit_b.py 4096 64 1 10\` (2 x 64 (sample, level) units of B + C at N=4096: warm-up call + timed call, 10 draws per unit)" 0 > $OUT/kernel_stats_unit_b.md
python3 tools/kernel_stats_md.py $OUT/trace_c2 "rocprofv3 --kernel-trace --stats of the unit-A bench at BASELINE config 2 (N=1024 D=4 nU=1, 4 x 8192 posterior samples: 1 warm-up + 3 timed steps)" 32768 > $OUT/kernel_stats_c2.md
mkdir -p $OUT/pmcA $OUT/pmcB
cp -r $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_valu $OUT/pmcA/ ; cp -r $OUT/b_pmc_fetch $OUT/b_pmc_write $OUT/pmcB/
python3 tools/pmc_summary.py $OUT/pmcA "tile_gemm_nt_kernel<1, 0>" $OUT/pmc_tile_gemm.json > $OUT/pmc_tile_gemm.md
python3 tools/pmc_summary.py $OUT/pmcA "tile_fused_strip_kernel" $OUT/pmc_fused.json > $OUT/pmc_fused.md
python3 tools/pmc_summary.py $OUT/pmcA "gram_kernel" $OUT/pmc_gram.json > $OUT/pmc_gram.md
python3 tools/pmc_summary.py $OUT/pmcA "ite_mean_kernel" $OUT/pmc_ite_mean.json > $OUT/pmc_ite_mean.md
python3 tools/pmc_summary.py $OUT/pmcB "draws_mfma_kernel" $OUT/pmc_draws.json > $OUT/pmc_draws.md
rm -rf $OUT/pmcA $OUT/pmcB
head -22 $OUT/kernel_stats.md; tail -4 $OUT/pmc_tile_gemm.md; tail -4 $OUT/pmc_fused.md; tail -6 $OUT/pmc_draws.md
