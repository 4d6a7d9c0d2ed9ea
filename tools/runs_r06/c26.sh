#!/bin/bash
# round 6, call 26: task launch for 2 <= nt <= 24 with groups of 32 (new defaults): the task tests, the whole GPU suite, smoke, sizes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c26; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_tasks.py -m gpu -x -q > $O/tasks.log 2>&1; echo "tasks rc=$?"; tail -3 $O/tasks.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
bash tools/r06_sizes.sh > $O/sizes.txt 2>&1; cat $O/sizes.txt
