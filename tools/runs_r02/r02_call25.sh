#!/bin/bash
# gpslc_nodes_draw: parity, ABI tests, chain tests and timings
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_25
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_small_path.py tests/test_gpu_model_nodes.py tests/test_gpu_neec.py tests/test_gpu_abi_edges.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_latency.py 2>&1 | grep -E "n=(150|272) |gpslc"
