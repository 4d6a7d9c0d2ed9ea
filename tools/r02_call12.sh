#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_12
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for i in 1 2; do python tools/bench_unit_b.py 4096 128 1 10 2>&1 | grep -v amdgpu | tail -1; done
python tools/bench_unit_b.py 4096 64 1 100 2>&1 | grep -v amdgpu | tail -1
