#!/bin/bash
# LDS-resident node-score kernel with lookahead: parity tests, phase clocks, latency, chain times
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_20
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_small_path.py tests/test_gpu_model_nodes.py tests/test_gpu_neec.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
GPSLC_SMALL_STAMPS=2 timeout -k 10 100 python tools/mid_stamps.py 150 2 2>&1 | grep small_gp
timeout -k 10 300 python tools/bench_latency.py > $OUT/latency.log 2>&1; echo "lat rc=$?"; grep -E "n=(150|160|272)|gpslc" $OUT/latency.log
timeout -k 10 300 python tools/bench_neec_example.py 2>&1 | grep "gpslc("
