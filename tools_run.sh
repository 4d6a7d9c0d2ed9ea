mkdir -p gpurun_out
GPSLC_PAIR=1 timeout -k 10 800 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_pair.log 2>&1; tail -4 gpurun_out/pytest_pair.log
timeout -k 10 900 python tools_ab.py 2>&1 | tail -14
