#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_4
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_small_path.py tests/test_gpu_model_nodes.py tests/test_gpu_neec.py tests/test_gpu_abi_edges.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -30 $OUT/pytest.log
timeout -k 10 300 python tools/bench_latency.py > $OUT/lat.log 2>&1; cat $OUT/lat.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lat_prof -- python3 $GRAFT_REPO_ROOT/tools/bench_latency.py > $OUT/lat_prof.log 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/kernel_stats_md.py $OUT/lat_prof "latency tool" 0 | head -20
