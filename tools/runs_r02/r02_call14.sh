#!/bin/bash
# PMC passes for the two VALU kernels of unit A (gram_kernel, ite_mean_kernel): where do their wave cycles go
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_14
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-units --no-profile"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_a -- $B > $OUT/pmc_a.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc_b -- $B > $OUT/pmc_b.log 2>&1
echo "rc=$?"; tail -3 $OUT/pmc_a.log | cut -c1-300; tail -3 $OUT/pmc_b.log | cut -c1-300
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT "gram_kernel" > $OUT/pmc_gram.md; python3 tools/pmc_summary.py $OUT "ite_mean_kernel" > $OUT/pmc_ite_mean.md
cat $OUT/pmc_gram.md $OUT/pmc_ite_mean.md
find $OUT -name "*.csv" -size +2M -delete
