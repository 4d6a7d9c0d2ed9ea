#!/bin/bash
# round 6, call 34: the diagonal tasks at N = 4096 (stamps: 261 k + 122 k clocks per K tile against a strip's 53 k + 72 k for 16 / 9 of
# the MFMAs): issue priority of the diagonal tasks 0 / 1 / 3 (measurement build, GPSLC_TASK_FENCE bits 4..5) and staging depth 4
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c34; mkdir -p $O
B="python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units --no-panel-leg --no-profile"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for rep in 1 2; do
for f in 48 16 0; do
GPSLC_TASK_FENCE=$f timeout -k 10 300 $B --diag-lib > $O/prio${f}_$rep.json 2> $O/err.txt; val $O/prio${f}_$rep.json "N=4096 diagonal-task priority bits $f (48 = 3, 16 = 1, 0 = 0)"
done
timeout -k 10 300 $B > $O/prod_$rep.json 2> $O/err.txt; val $O/prod_$rep.json "N=4096 production"
timeout -k 10 300 $B --lib causalgpslc.jl_amd/csrc/libgpslc_hip_var_pf4.so > $O/pf4_$rep.json 2> $O/err.txt; val $O/pf4_$rep.json "N=4096 SYRK_PF=4"
done
GPSLC_TASK_FENCE=0 GPSLC_TASK_DBG=2 timeout -k 10 300 $B --diag-lib --steps 1 > $O/dbg.json 2> $O/err.txt
python3 tools/task_stamps.py gpurun_out/task_dbg.bin > $O/stamps_prio0.md; grep -E "diag k=(0|1|4|8|16|24|30) |strip k=(0|1|8|16|24) |back|shares" $O/stamps_prio0.md
rm -f gpurun_out/task_dbg.bin
