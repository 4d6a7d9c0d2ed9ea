#!/bin/bash
# panel width at the smaller sizes (config 2: N = 1024; N = 2048)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_17
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { # n d nu sps panel
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-units --n $1 --d $2 --nu $3 --samples-per-step $4 --panel $5 > $OUT/b_$1_$5.json 2> $OUT/b_$1_$5.err || { echo "failed $1 $5"; tail -3 $OUT/b_$1_$5.err; return 1; }
  python3 -c "
import json
d=json.loads(open('$OUT/b_$1_$5.json').read().strip().splitlines()[-1])
print('N=$1 panel=$5', round(d['value'],1), d['unit'], 'roofline', d['roofline'].get('achieved'), d['roofline'].get('kernel','')[:40])"
}
for P in 0 2 4 8; do run 1024 4 1 8192 $P || exit 1; done
for P in 0 4 8 16; do run 2048 8 2 4096 $P || exit 1; done
for P in 0 2 4; do run 512 4 1 16384 $P || exit 1; done
