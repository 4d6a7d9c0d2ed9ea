#!/bin/bash
# round 6, call 33: in-kernel stamps per task of the persistent launch at N = 4096 (1,024 matrices, nt = 32) and N = 2048
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c33; mkdir -p $O
B="python3 bench.py --diag-lib --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --no-profile"
GPSLC_TASK_DBG=2 timeout -k 10 300 $B > $O/n4096.json 2> $O/err.txt
python3 tools/task_stamps.py gpurun_out/task_dbg.bin > $O/stamps_n4096.md; cat $O/stamps_n4096.md
rm -f gpurun_out/task_dbg.bin
