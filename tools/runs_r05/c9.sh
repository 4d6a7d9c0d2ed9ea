#!/bin/bash
# round 5, call 9: PMC passes of unit A and of the draw kernels (tools/profile_r05.sh part b)
set -e
mkdir -p gpurun_out/r05
bash tools/profile_r05.sh r05p b 2>&1 | tail -30 | tee gpurun_out/r05/c9.log
