#!/bin/bash
# node-score kernels + robust factorisation: parity tests + latency
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_12
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_small_path.py tests/test_gpu_model_nodes.py tests/test_gpu_neec.py tests/test_gpu_estimation.py tests/test_gpu_abi_edges.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] && timeout -k 10 300 python tools/bench_latency.py > $OUT/latency.log 2>&1; echo "lat rc=$?"; cat $OUT/latency.log
[ $rc -eq 0 ] && timeout -k 10 300 python tools/bench_neec_example.py > $OUT/neec_example.log 2>&1; echo "ex rc=$?"; tail -5 $OUT/neec_example.log
