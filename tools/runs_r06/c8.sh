#!/bin/bash
# round 6, call 8: the whole GPU suite on the production library with the task launch's final defaults (5..8 tiles, group 8, 2 rows)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c8; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
