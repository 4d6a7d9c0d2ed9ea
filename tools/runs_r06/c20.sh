#!/bin/bash
# round 6, call 20: the whole GPU suite + smoke on the production library, then unit A across sizes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c20; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
bash tools/r06_sizes.sh > $O/sizes.txt 2>&1; cat $O/sizes.txt
