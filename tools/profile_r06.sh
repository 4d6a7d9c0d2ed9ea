#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats + PMC passes of the bench commands (round 6), in two parts so that each
# fits one gpurun call:   bash tools/profile_r06.sh <tag> a     (kernel traces + config-2 PMC)
#                         bash tools/profile_r06.sh <tag> b     (PMC passes of unit A and of the draws kernel)
# both write gpurun_out/<tag>/ ; then  python3 tools/collect_profiles.py <tag> r06
TAG=${1:-r06}
PART=${2:-a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-panel-leg"
UB="python3 $GRAFT_REPO_ROOT/tools/bench_unit_b.py 4096 64 1 10"
C2="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --n 1024 --d 4 --nu 1 --samples-per-step 8192"
C2L="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --n 1024 --d 4 --nu 1 --samples-per-step 1000"
KS="python3 $GRAFT_REPO_ROOT/tools/kernel_stats_md.py"
PS="python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py"
if [ "$PART" = a ]; then
# (1) unit A: the bench's timed region, per-kernel durations (HIP-event profiling on, as in the driver's run)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --no-units > $OUT/trace.log 2>&1 &&
# (1b) the same with one launch per tile column (the schedule of rounds 1-5: panels of 8 + trailing updates)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_p -- $B --no-units --task-tiles 0 > $OUT/trace_p.log 2>&1 &&
# (2) units B and C: 64 (sample, level) units with 10 draws each per call (warm-up call + two full-size calls)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_b -- $UB > $OUT/trace_b.log 2>&1 &&
# (2b) BASELINE config 2 (N = 1024), 8,192 samples per step
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2 -- $C2 > $OUT/trace_c2.log 2>&1 &&
# (2c) BASELINE config 2 AS STATED: ONE call with S = 1000 per step (4 calls: 1 warm-up + 3 timed): launches per call
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2l -- $C2L > $OUT/trace_c2l.log 2>&1 &&
# (2d) HBM traffic per kernel at config 2: two PMC passes
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c2pmc/fetch -- $C2 --no-profile > $OUT/c2pmc_fetch.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/c2pmc/write -- $C2 --no-profile > $OUT/c2pmc_write.log 2>&1 &&
# (2e) the persistent factorisation launch at config 2: MFMA pipe and outstanding-read counters
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $OUT/c2pmc/sq -- $C2 --no-profile > $OUT/c2pmc_sq.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum --output-format csv -d $OUT/c2pmc/ea -- $C2 --no-profile > $OUT/c2pmc_ea.log 2>&1
grep '"metric"' $OUT/trace.log | cut -c1-300
tail -1 $OUT/trace_b.log
$KS $OUT/trace "rocprofv3 --kernel-trace --stats of \`python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg\` (4 x 1024 posterior samples of unit A at N=4096 D=8 nU=2: 1 warm-up + 3 timed steps)" 4096 > $OUT/kernel_stats.md
$KS $OUT/trace_p "rocprofv3 --kernel-trace --stats of \`python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --task-tiles 0\` (the panel schedule of rounds 1-5 at N=4096: 4 x 1024 posterior samples)" 4096 > $OUT/kernel_stats_panel_schedule.md
$KS $OUT/trace_b "rocprofv3 --kernel-trace --stats of \`python3 tools/bench_unit_b.py 4096 64 1 10\` (3 x 64 (sample, level) units of B + C at N=4096: warm-up call + two full-size calls, 10 draws per unit)" 0 > $OUT/kernel_stats_unit_b.md
$KS $OUT/trace_c2 "rocprofv3 --kernel-trace --stats of the unit-A bench at BASELINE config 2 (N=1024 D=4 nU=1, 4 x 8192 posterior samples: 1 warm-up + 3 timed steps)" 32768 > $OUT/kernel_stats_c2.md
$KS $OUT/trace_c2l "rocprofv3 --kernel-trace --stats of BASELINE config 2 as stated (N=1024 D=4 nU=1, ONE gpslc_predict_dev call with S = 1000 per step; 4 calls: 1 warm-up + 3 timed): kernel launches per call = calls / 4" 4000 > $OUT/kernel_stats_c2_literal.md
{ echo "## HBM traffic per kernel at BASELINE config 2 (N = 1024, D 4, nU 1; a launch = one chunk of 8,192 posterior samples)"; echo;
i=0; mkdir -p $OUT/c2json
for k in potrf_tasks_kernel gram_kernel ite_mean_kernel rhs_tiles_kernel rhs_prepare_kernel epilogue_kernel tile_fused_strip_kernel diag_update_potrf_kernel diag_potrf_inv_la_kernel backsolve_update_kernel backsolve_alpha_kernel extract_z_kernel; do
  i=$((i+1)); echo "### $k"; $PS $OUT/c2pmc "$k" $OUT/c2json/k$i.json | tail -2; echo; done; } > $OUT/pmc_c2_per_kernel.md
rm -rf $OUT/c2pmc $OUT/trace $OUT/trace_p $OUT/trace_b $OUT/trace_c2 $OUT/trace_c2l
head -22 $OUT/kernel_stats.md; head -8 $OUT/kernel_stats_c2_literal.md
else
rm -f $OUT/pmc_*.json $OUT/pmc_*.md      # a kernel without rows must not leave an older summary behind for the collector
# (3) PMC passes, one counter group per run, unit A
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmcA/pmc_fetch -- $B --no-units --no-profile > $OUT/pmc_fetch.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmcA/pmc_write -- $B --no-units --no-profile > $OUT/pmc_write.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmcA/pmc_sq -- $B --no-units --no-profile > $OUT/pmc_sq.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum --output-format csv -d $OUT/pmcA/pmc_ea -- $B --no-units --no-profile > $OUT/pmc_ea.log 2>&1 &&
# (3a) the panel schedule of rounds 1-5 (--task-tiles 0: what roofline.panel_schedule times): trailing-update and strip kernels
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmcP/pmc_fetch -- $B --no-units --no-profile --task-tiles 0 > $OUT/pmcP_fetch.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmcP/pmc_write -- $B --no-units --no-profile --task-tiles 0 > $OUT/pmcP_write.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmcP/pmc_sq -- $B --no-units --no-profile --task-tiles 0 > $OUT/pmcP_sq.log 2>&1 &&
# (3b) VALU issue counters (the two fp64-VALU kernels of unit A: Gram build, MeanITE pass)
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmcA/pmc_valu -- $B --no-units --no-profile > $OUT/pmc_valu.log 2>&1 &&
# (4) PMC passes for the draws kernel (unit C): bytes of L_c actually fetched per launch
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmcB/b_pmc_fetch -- $UB > $OUT/b_pmc_fetch.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmcB/b_pmc_write -- $UB > $OUT/b_pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
# unit A at N = 4096 is ONE persistent launch of tile tasks since the second half of round 6 (the panel leg of the bench line still
# runs the trailing-update and strip kernels: their rows are that leg's launches)
$PS $OUT/pmcA "potrf_tasks_kernel" $OUT/pmc_potrf_tasks.json > $OUT/pmc_potrf_tasks.md
$PS $OUT/pmcP "tile_gemm_nt_kernel<1, 0>" $OUT/pmc_tile_gemm.json > $OUT/pmc_tile_gemm.md
$PS $OUT/pmcP "tile_fused_strip_kernel" $OUT/pmc_fused.json > $OUT/pmc_fused.md
$PS $OUT/pmcA "gram_kernel" $OUT/pmc_gram.json > $OUT/pmc_gram.md
$PS $OUT/pmcA "ite_mean_kernel" $OUT/pmc_ite_mean.json > $OUT/pmc_ite_mean.md
$PS $OUT/pmcB "draws_stream_kernel" $OUT/pmc_draws.json > $OUT/pmc_draws.md
$PS $OUT/pmcB "draws_zstage_kernel" $OUT/pmc_draws_zstage.json > $OUT/pmc_draws_zstage.md
rm -rf $OUT/pmcA $OUT/pmcB $OUT/pmcP
tail -4 $OUT/pmc_potrf_tasks.md; tail -4 $OUT/pmc_tile_gemm.md; tail -6 $OUT/pmc_draws.md
fi
