timeout -k 10 600 python tools_diag.py 2>&1 | tail -8
