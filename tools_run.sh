mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest14.log 2>&1; tail -12 gpurun_out/pytest14.log
