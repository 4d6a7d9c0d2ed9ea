#!/bin/bash
# round 6, call 47: one inlined copy of every task body in potrf_tasks_kernel (code 128 KB -> 84 KB; the instruction cache two CUs
# share holds 64 KB) against HEAD's build (libgpslc_hip_var_head.so), same box, alternating; bit-identity tests first
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c47; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_tasks.py -m gpu -x -q > $O/tasks.log 2>&1; echo "tasks rc=$?"; tail -2 $O/tasks.log
B="python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --no-profile"
H="--lib causalgpslc.jl_amd/csrc/libgpslc_hip_var_head.so"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for rep in 1 2 3; do
timeout -k 10 300 $B --n 1024 --d 4 --nu 1 --samples-per-step 8192 > $O/c2_new_$rep.json 2> $O/err.txt; val $O/c2_new_$rep.json "config 2 one copy per body"
timeout -k 10 300 $B --n 1024 --d 4 --nu 1 --samples-per-step 8192 $H > $O/c2_head_$rep.json 2> $O/err.txt; val $O/c2_head_$rep.json "config 2 HEAD"
done
for rep in 1 2; do
timeout -k 10 300 $B --n 512 --d 4 --nu 1 --samples-per-step 16384 > $O/n512_new_$rep.json 2> $O/err.txt; val $O/n512_new_$rep.json "N=512 one copy per body"
timeout -k 10 300 $B --n 512 --d 4 --nu 1 --samples-per-step 16384 $H > $O/n512_head_$rep.json 2> $O/err.txt; val $O/n512_head_$rep.json "N=512 HEAD"
timeout -k 10 300 $B > $O/n4096_new_$rep.json 2> $O/err.txt; val $O/n4096_new_$rep.json "N=4096 one copy per body"
timeout -k 10 300 $B $H > $O/n4096_head_$rep.json 2> $O/err.txt; val $O/n4096_head_$rep.json "N=4096 HEAD"
done
timeout -k 10 300 $B --n 1024 --d 4 --nu 1 --samples-per-step 1000 --steps 20 > $O/c2l_new.json 2> $O/err.txt; val $O/c2l_new.json "config 2 as stated, one copy per body"
timeout -k 10 300 $B --n 1024 --d 4 --nu 1 --samples-per-step 1000 --steps 20 $H > $O/c2l_head.json 2> $O/err.txt; val $O/c2l_head.json "config 2 as stated, HEAD"
