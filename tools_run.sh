mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_neec.py -m gpu -q -x --durations=6 > gpurun_out/pytest19.log 2>&1; tail -14 gpurun_out/pytest19.log
