mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q -x > gpurun_out/pytest15.log 2>&1; tail -8 gpurun_out/pytest15.log
