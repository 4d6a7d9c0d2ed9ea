#!/bin/bash
# round 6, call 37: final sources — unit A across sizes and the driver-style bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c37; mkdir -p $O
bash tools/r06_sizes.sh > $O/sizes.txt 2>&1; cat $O/sizes.txt
S=$(date +%s); timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - S )) s"; cut -c1-300 $O/bench_default.json
