#!/bin/bash
# LDS-broadcast pivot chains in all three users: parity tests, latency, unit B
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_21
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 100 python tools/mid_stamps.py 272 8 2>&1 | grep mid_gp
timeout -k 10 300 python tools/bench_latency.py > $OUT/latency.log 2>&1; echo "lat rc=$?"; grep -E "n=(150|160|272|400|640)|gpslc" $OUT/latency.log
timeout -k 10 300 python tools/bench_neec_example.py 2>&1 | grep -E "gpslc\(|predict"
timeout -k 10 300 python tools/bench_unit_b.py 4096 64 1 10 2>&1 | tail -2
