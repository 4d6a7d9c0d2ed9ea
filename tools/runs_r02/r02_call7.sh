#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_7
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_small_path.py tests/test_gpu_model_nodes.py tests/test_gpu_neec.py tests/test_gpu_abi_edges.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
GPSLC_SMALL_STAMPS=4 timeout -k 10 300 python tools/bench_latency.py --diag-lib > $OUT/lat_diag.log 2>&1; grep -v amdgpu.ids $OUT/lat_diag.log | head -12
timeout -k 10 300 python tools/bench_latency.py > $OUT/lat.log 2>&1; grep -v amdgpu.ids $OUT/lat.log
