mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest18.log 2>&1; tail -6 gpurun_out/pytest18.log
