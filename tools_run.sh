GPSLC_XSYNC=1 timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -2
timeout -k 10 900 python tools_ab.py 2>&1 | tail -6
