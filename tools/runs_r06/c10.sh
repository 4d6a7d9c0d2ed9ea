#!/bin/bash
# round 6, call 10: kernel durations of the symmetric MeanITE pass against the two-sided one (rocprofv3 kernel trace)
O=$GRAFT_REPO_ROOT/gpurun_out/r06c10; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --diag-lib --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-profile"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_sym -- $B --n 1024 --d 4 --nu 1 --samples-per-step 8192 > $O/sym.log 2>&1
export GPSLC_ITE_SYM=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_old -- $B --n 1024 --d 4 --nu 1 --samples-per-step 8192 > $O/old.log 2>&1
unset GPSLC_ITE_SYM
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_sym4 -- $B > $O/sym4.log 2>&1
export GPSLC_ITE_SYM=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_old4 -- $B > $O/old4.log 2>&1
for d in t_sym t_old t_sym4 t_old4; do echo "== $d"; f=$(find $O/$d -name "*kernel_stats.csv" | head -1); grep -i "ite_mean\|potrf_tasks\|gram_kernel" $f | cut -c1-160; done
rm -rf $O/t_sym $O/t_old $O/t_sym4 $O/t_old4
