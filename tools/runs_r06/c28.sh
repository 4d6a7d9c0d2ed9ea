#!/bin/bash
# round 6, call 28: does one tile row per strip task (more concurrent sharers of the B panel) raise the task launch's L2 hit rate?
# config 2, measurement build, GPSLC_TASK_ROWS = 1 / 2 / 3; FETCH_SIZE and TCC hit / miss passes
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06c28; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C2="python3 $GRAFT_REPO_ROOT/bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --n 1024 --d 4 --nu 1 --samples-per-step 8192 --no-profile"
for r in 1 2 3; do
export GPSLC_TASK_ROWS=$r
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/r$r/fetch -- $C2 > $OUT/r${r}_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/r$r/write -- $C2 > $OUT/r${r}_write.log 2>&1
echo "rows per strip task = $r"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/r$r potrf_tasks_kernel | tail -2
grep '"metric"' $OUT/r${r}_fetch.log | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']), 'samples/s under the FETCH_SIZE pass')"
rm -rf $OUT/r$r
done
