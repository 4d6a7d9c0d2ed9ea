#!/bin/bash
# k_gram.hip compiled with the max-ILP machine scheduler: parity + unit-A throughput + per-kernel times
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_24
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_estimation.py tests/test_gpu_kernel.py tests/test_gpu_abi_edges.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units > $OUT/bench_$i.json 2> $OUT/bench_$i.err; python3 -c "
import json
d=json.loads(open('$OUT/bench_$i.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['sate_rel_err']['mean'])"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kernel_stats_md.py $OUT/trace "max-ilp gram" 4096 > $OUT/stats.md; grep -E "gram|ite_mean|sum of" $OUT/stats.md
find $OUT -name "*.csv" -size +2M -delete
