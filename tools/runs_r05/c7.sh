#!/bin/bash
# round 5, call 7: smoke(), the full GPU suite, then the kernel traces + config-2 PMC passes (tools/profile_r05.sh part a)
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c7.log
: > $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee -a $O
bash tools/profile_r05.sh r05p a 2>&1 | tail -40 | tee -a $O
