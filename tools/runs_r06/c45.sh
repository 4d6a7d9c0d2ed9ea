#!/bin/bash
# round 6, call 45: per-task stamps of the persistent launch at config 2 on the final schedule (groups of 32, two rows per strip task)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c45; mkdir -p $O
B="python3 bench.py --diag-lib --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --no-profile --n 1024 --d 4 --nu 1 --samples-per-step 8192"
GPSLC_TASK_DBG=2 timeout -k 10 300 $B > $O/n1024.json 2> $O/err.txt
python3 tools/task_stamps.py gpurun_out/task_dbg.bin > $O/stamps_n1024.md; cat $O/stamps_n1024.md
rm -f gpurun_out/task_dbg.bin
