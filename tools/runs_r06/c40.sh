#!/bin/bash
# round 6, call 40: final sources — the whole GPU suite, smoke, the driver-style bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c40; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -2 $O/suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
S=$(date +%s); timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - S )) s"; cut -c1-300 $O/bench_default.json
