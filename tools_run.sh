mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_neec.py -m gpu -q -x --durations=4 > gpurun_out/pytest20.log 2>&1; tail -12 gpurun_out/pytest20.log
