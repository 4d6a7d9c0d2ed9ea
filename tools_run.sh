timeout -k 10 600 python tools_neec.py 2>&1 | tail -9
