#!/bin/bash
# round 6, call 1: first run of the persistent factorisation launch — its own tests, then the suites that now go through it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c1
timeout -k 10 300 python -m pytest tests/test_gpu_tasks.py -x -q > gpurun_out/r06c1/tasks.log 2>&1; echo "tasks rc=$?"; tail -5 gpurun_out/r06c1/tasks.log
