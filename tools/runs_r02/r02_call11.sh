#!/bin/bash
# mid-size node-score kernel: parity tests + latency
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_11
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_small_path.py tests/test_gpu_model_nodes.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] && timeout -k 10 300 python tools/bench_latency.py > $OUT/latency.log 2>&1; echo "lat rc=$?"; cat $OUT/latency.log
