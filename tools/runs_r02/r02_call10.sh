#!/bin/bash
# round-end rehearsal: smoke(), full gpu suite, default bench
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_10
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout -k 10 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-1500 $OUT/bench.json
