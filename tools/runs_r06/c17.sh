#!/bin/bash
# round 6, call 17: latency of single node scores / draws on the tiled path, task launch against per-column launches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c17; mkdir -p $O
timeout -k 10 300 python3 tools/bench_latency_tiled.py > $O/lat.txt 2> $O/lat.err; cat $O/lat.txt; tail -2 $O/lat.err
