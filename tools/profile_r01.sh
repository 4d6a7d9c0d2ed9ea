#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats + PMC passes of the default bench command.
# Usage: bash tools/profile_r01.sh <tag>
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --no-profile > $OUT/pmc_fetch.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $B --no-profile > $OUT/pmc_write.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- $B --no-profile > $OUT/pmc_sq.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- $B --no-profile > $OUT/pmc_grbm.log 2>&1
grep metric $OUT/trace.log
