#!/bin/bash
# unit B: W-solve fusion depth A/B (measurement build)
cd $GRAFT_REPO_ROOT
for mk in 32 0 4 8 12 16 32 0; do
  echo -n "FUSE_W_MAXK=$mk: "; GPSLC_FUSE_W_MAXK=$mk python tools/bench_unit_b.py 4096 128 1 10 --diag-lib 2>&1 | grep -v amdgpu | tail -1
done
