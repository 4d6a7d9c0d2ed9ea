#!/bin/bash
# unit A across problem sizes, production library — round 6
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-units --no-config4 --no-configs"
run() { tag=$1; shift; timeout -k 10 300 $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); s=r.get('second_kernel',{})
print('$tag', round(d['value'],1), 'samples/s |', r.get('kernel','-')[:28], round(r.get('achieved',0),1), '|', s.get('kernel','-')[:28], round(s.get('achieved',0),1))
"; }
run "N=128"   --n 128 --d 4 --nu 1 --samples-per-step 32768
run "N=256"   --n 256 --d 4 --nu 1 --samples-per-step 16384
run "N=256-per-column" --n 256 --d 4 --nu 1 --samples-per-step 16384 --task-tiles 0
run "N=384"   --n 384 --d 4 --nu 1 --samples-per-step 16384
run "N=384-per-column" --n 384 --d 4 --nu 1 --samples-per-step 16384 --task-tiles 0
run "N=512"   --n 512 --d 4 --nu 1 --samples-per-step 16384
run "N=512-per-column" --n 512 --d 4 --nu 1 --samples-per-step 16384 --task-tiles 0
run "N=640"   --n 640 --d 4 --nu 1 --samples-per-step 8192
run "N=1024"  --n 1024 --d 4 --nu 1 --samples-per-step 8192
run "N=1024-per-column" --n 1024 --d 4 --nu 1 --samples-per-step 8192 --task-tiles 0
run "N=1024-sate-only" --n 1024 --d 4 --nu 1 --samples-per-step 8192 --no-mean-ite
run "N=1024-S1000" --n 1024 --d 4 --nu 1 --samples-per-step 1000 --steps 10
run "N=1024-S1000-per-column" --n 1024 --d 4 --nu 1 --samples-per-step 1000 --steps 10 --task-tiles 0
run "N=1536"  --n 1536 --d 8 --nu 2 --samples-per-step 4096
run "N=1536-panels-of-8" --n 1536 --d 8 --nu 2 --samples-per-step 4096 --task-tiles 0
run "N=2048"  --n 2048 --d 8 --nu 2 --samples-per-step 4096
run "N=2048-panels-of-8" --n 2048 --d 8 --nu 2 --samples-per-step 4096 --task-tiles 0
run "N=3072"  --n 3072 --d 8 --nu 2 --samples-per-step 1820
run "N=3072-panels-of-8" --n 3072 --d 8 --nu 2 --samples-per-step 1820 --task-tiles 0
run "N=4096"  --n 4096 --d 8 --nu 2 --samples-per-step 1024
run "N=4096-sate-only" --n 4096 --d 8 --nu 2 --samples-per-step 1024 --no-mean-ite
run "N=4096-L64" --n 4096 --d 8 --nu 2 --samples-per-step 1024 --levels 64
run "N=4096-L64-sate-only" --n 4096 --d 8 --nu 2 --samples-per-step 1024 --levels 64 --no-mean-ite
run "N=4096-fp32kernel" --n 4096 --d 8 --nu 2 --samples-per-step 1024 --fp32-kernel
run "N=8192"  --n 8192 --d 8 --nu 2 --samples-per-step 256
run "N=16384-binary" --n 16384 --d 16 --nu 4 --samples-per-step 64 --binary-t
run "N=16384-binary-fp32kernel" --n 16384 --d 16 --nu 4 --samples-per-step 64 --binary-t --fp32-kernel
