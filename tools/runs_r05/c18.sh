#!/bin/bash
# round 5, call 18: final confirmation on the committed tree — smoke(), the full GPU suite
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05/c18.log
: > $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee -a $O
