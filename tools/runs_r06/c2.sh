#!/bin/bash
# round 6, call 2: the whole GPU suite through the persistent factorisation launch, then the same-box A/B at config 2 / N = 512
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
B="python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units"
for rep in 1 2; do
for tt in 8 0; do
  timeout -k 10 200 $B --n 1024 --d 4 --nu 1 --samples-per-step 8192 --task-tiles $tt > $O/c2_t${tt}_$rep.json 2> $O/c2_t${tt}_$rep.err
  python3 -c "import json,sys; d=json.loads(open('$O/c2_t${tt}_$rep.json').read().strip().splitlines()[-1]); print('n1024 tiles=$tt', round(d['value']), d.get('roofline',{}).get('kernel','')[:24], round(d.get('roofline',{}).get('frac',0),3))"
done; done
for tt in 8 0; do
  timeout -k 10 200 $B --n 512 --d 4 --nu 1 --samples-per-step 16384 --task-tiles $tt > $O/n512_t${tt}.json 2> $O/n512_t${tt}.err
  python3 -c "import json,sys; d=json.loads(open('$O/n512_t${tt}.json').read().strip().splitlines()[-1]); print('n512 tiles=$tt', round(d['value']))"
  timeout -k 10 200 $B --n 1024 --d 4 --nu 1 --samples-per-step 1000 --steps 5 --task-tiles $tt > $O/c2l_t${tt}.json 2> $O/c2l_t${tt}.err
  python3 -c "import json,sys; d=json.loads(open('$O/c2l_t${tt}.json').read().strip().splitlines()[-1]); print('n1024 S=1000 tiles=$tt', round(d['value']))"
done
