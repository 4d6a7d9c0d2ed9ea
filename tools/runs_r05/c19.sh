#!/bin/bash
# round 5, call 19: memory-side read counters of the L2 (TCC_EA0_RDREQ / _DRAM / _LEVEL) at config 2 — does any counter separate Infinity-Cache
# hits from HBM reads?
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05/ea
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C2="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile --n 1024 --d 4 --nu 1 --samples-per-step 8192"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/p1 -- $C2 > $OUT/p1.log 2>&1
UB="python3 $GRAFT_REPO_ROOT/tools/bench_unit_b.py 4096 64 1 10"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/p2 -- $UB > $OUT/p2.log 2>&1
cd $GRAFT_REPO_ROOT
for k in tile_fused_strip_kernel diag_update_potrf_kernel backsolve_update_kernel gram_kernel; do python3 tools/pmc_summary.py $OUT/p1 "$k" | tail -8; done | tee $OUT/summary.txt
python3 tools/pmc_summary.py $OUT/p2 "draws_stream_kernel" | tail -8 | tee -a $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2
