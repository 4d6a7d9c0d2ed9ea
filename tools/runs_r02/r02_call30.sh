#!/bin/bash
# shader clock during the three kinds of kernels of unit A (s_memtime against the 100 MHz s_memrealtime, measurement build)
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-units --diag-lib"
echo "== Gram build (fp64 VALU + 67 MB of tile stores per sample)"
GPSLC_GRAM_DBG=1 timeout -k 10 200 $B > /dev/null 2>&1; python3 tools/gram_stamps.py | head -1
echo "== trailing update, m = 25 tile rows (fp64 MFMA)"
GPSLC_GEMM_DBG=25 timeout -k 10 200 $B > /dev/null 2>&1; python3 tools/gemm_stamps.py | tail -1
echo "== fused in-panel launch, K = 4 tiles (fp64 MFMA)"
GPSLC_GEMM_DBG_FUSEK=4 timeout -k 10 200 $B > /dev/null 2>&1; python3 tools/gemm_stamps.py | tail -1
