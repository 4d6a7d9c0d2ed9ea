#!/bin/bash
# GPU box: gpu test-suite + kernel trace of unit B (current state)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/unitb -- python3 $GRAFT_REPO_ROOT/tools/bench_unit_b.py 4096 128 1 8 > $OUT/unitb.log 2>&1
tail -2 $OUT/unitb.log
cd $GRAFT_REPO_ROOT && python3 tools/kernel_stats_md.py $OUT/unitb "unit B N=4096 S=128 L=1 spp=8 (round-2 start)" 0 > $OUT/unitb_stats.md; cat $OUT/unitb_stats.md
