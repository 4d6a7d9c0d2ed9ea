#!/bin/bash
# round 6, call 12: node / U-prior draws on the tiled path, the chains that use them, the one-rank RCCL bench test
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c12; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_small_path.py tests/test_gpu_neec.py tests/test_gpu_model_nodes.py tests/test_bench_contract.py tests/test_gpu_multi.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
