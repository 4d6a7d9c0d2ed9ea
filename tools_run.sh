mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest13.log 2>&1; tail -5 gpurun_out/pytest13.log
timeout -k 10 300 python tools_lat.py 2>&1 | grep -E "ms$"
