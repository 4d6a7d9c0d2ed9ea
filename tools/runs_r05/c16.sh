#!/bin/bash
# round 5, call 16: final kernel traces + config-2 PMC passes (tools/profile_r05.sh part a) on the final sources
set -e
mkdir -p gpurun_out/r05
bash tools/profile_r05.sh r05p a 2>&1 | tail -30 | tee gpurun_out/r05/c16.log
