#!/bin/bash
# round 6, call 27: the descriptor widened to nt = 32 — N = 4096 as ONE left-looking panel of tasks against the production
# schedule (panels of 8 + trailing updates); bit-identity first
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c27; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_tasks.py -m gpu -x -q > $O/tasks.log 2>&1; echo "tasks rc=$?"; tail -3 $O/tasks.log
timeout -k 10 300 python - > $O/ident.log 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import causalgpslc_jl_amd as gp, cases
for n, S in ((3500, 3), (4096, 9)):
    c = cases.make_case(n, "UX", False, S=S, seed=n)
    out = []
    for tiles in (32, 0):
        g = cases.gpslc_object(gp, c)
        g.ctx().set_task_schedule(2, tiles, 1, 0)
        out.append(gp.predict(g, [0.1, 0.4], want_mean_ite=True))
    print(n, all(np.array_equal(x, y) for x, y in zip(*out)))
PY
echo "ident rc=$?"; cat $O/ident.log | tail -3
B="python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-config4 --no-configs --no-units"
val() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')"; }
for r in 1 2; do
timeout -k 10 300 $B > $O/def$r.json 2> $O/err.txt; val $O/def$r.json "N=4096 default"
timeout -k 10 300 $B --task-tiles 32 > $O/t$r.json 2> $O/err.txt; val $O/t$r.json "N=4096 one panel of tasks"
done
timeout -k 10 300 $B --task-tiles 32 --task-group 16 > $O/tg16.json 2> $O/err.txt; val $O/tg16.json "N=4096 one panel of tasks, group 16"
timeout -k 10 300 $B --task-tiles 32 --task-group 64 > $O/tg64.json 2> $O/err.txt; val $O/tg64.json "N=4096 one panel of tasks, group 64"
timeout -k 10 300 $B --task-tiles 32 --levels 64 > $O/tl64.json 2> $O/err.txt; val $O/tl64.json "N=4096 L=64 one panel of tasks"
timeout -k 10 300 $B --levels 64 > $O/dl64.json 2> $O/err.txt; val $O/dl64.json "N=4096 L=64 default"
