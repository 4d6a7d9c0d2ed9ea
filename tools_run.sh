for q in 16384 32768 65536 131072; do echo "GPSLC_UNITB_Q=$q"; GPSLC_UNITB_Q=$q timeout -k 10 300 python tools_unitb.py 2>&1 | tail -2; done
