#!/bin/bash
# round 6, call 35: diagonal-tile update of the task launch in the strip layout (syrk_strip_wave) from tile column k on
# (measurement build: GPSLC_TASK_SYRK_STRIP = 1 always / 4 / 8 / 255 never); bit-identity tests first
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c35; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_tasks.py -m gpu -x -q > $O/tasks.log 2>&1; echo "tasks rc=$?"; tail -2 $O/tasks.log
B="python3 bench.py --diag-lib --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --no-profile"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for rep in 1 2; do
for t in 255 1 4 8; do
GPSLC_TASK_SYRK_STRIP=$t timeout -k 10 300 $B > $O/n4096_t${t}_$rep.json 2> $O/err.txt; val $O/n4096_t${t}_$rep.json "N=4096 strip-layout diagonal update from k=$t"
done; done
for t in 255 1 4; do
GPSLC_TASK_SYRK_STRIP=$t timeout -k 10 300 $B --n 2048 --samples-per-step 4096 > $O/n2048_t$t.json 2> $O/err.txt; val $O/n2048_t$t.json "N=2048 from k=$t"
done
for t in 255 1 2 4; do
GPSLC_TASK_SYRK_STRIP=$t timeout -k 10 300 $B --n 1024 --d 4 --nu 1 --samples-per-step 8192 > $O/n1024_t$t.json 2> $O/err.txt; val $O/n1024_t$t.json "N=1024 from k=$t"
done
for t in 255 1 2; do
GPSLC_TASK_SYRK_STRIP=$t timeout -k 10 300 $B --n 512 --d 4 --nu 1 --samples-per-step 16384 > $O/n512_t$t.json 2> $O/err.txt; val $O/n512_t$t.json "N=512 from k=$t"
done
GPSLC_TASK_SYRK_STRIP=1 GPSLC_TASK_DBG=2 timeout -k 10 300 $B --steps 1 > $O/dbg.json 2> $O/err.txt
python3 tools/task_stamps.py gpurun_out/task_dbg.bin > $O/stamps_strip_layout.md; grep -E "diag k=(0|1|4|8|16|24|30) |strip k=(0|1|8|16|24) |back|shares" $O/stamps_strip_layout.md
rm -f gpurun_out/task_dbg.bin
