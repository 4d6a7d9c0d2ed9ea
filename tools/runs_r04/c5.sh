#!/bin/bash
# round 4, call 5: (a) diagonal-block kernel with LDS-only barriers + two trailing blocks per pass: parity + its rocprofv3
# duration at N = 1024; (b) chained launch with longest-items-first order against the three-launch column
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_05
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 -m pytest tests/test_gpu_estimation.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_model_nodes.py -m gpu -x -q > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
run() {
  label=$1; shift
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --no-configs --no-config4 --diag-lib "$@" > $OUT/c.json 2> $OUT/c.err || tail -3 $OUT/c.err
  python3 -c "
import json
d=json.loads(open('$OUT/c.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$label:', round(d['value'],1), r['kernel'][:24], round(r['achieved'],2), r.get('second_kernel',{}).get('achieved'))" | tee -a $OUT/log.txt
}
N1="--n 1024 --d 4 --nu 1 --samples-per-step 8192"
for rep in 1 2; do
for m in 0 1; do
GPSLC_CHAIN=$m run "N=1024 chain=$m" $N1
done
for m in 0 1; do
GPSLC_CHAIN=$m run "N=4096 chain=$m"
done
done
cd /tmp && export TMPDIR=/tmp
GPSLC_CHAIN=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config4 --no-configs --no-units --diag-lib $N1 > $OUT/trace_c2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kernel_stats_md.py $OUT/trace_c2 "rocprofv3 --kernel-trace --stats, N=1024 D=4 nU=1, 4 x 8192 samples, GPSLC_CHAIN=0 (measurement build)" 32768 > $OUT/kernel_stats_c2.md
head -16 $OUT/kernel_stats_c2.md
rm -rf $OUT/trace_c2
