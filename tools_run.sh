mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q -x > gpurun_out/pytest17.log 2>&1; tail -4 gpurun_out/pytest17.log
timeout -k 10 400 python bench.py --no-cpu-baseline --n 16384 --d 16 --nu 4 --samples-per-step 32 --steps 2 --warmup 1 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('N=16384 value=%.2f samples/s  ms/step=%.1f  mfma=%.1f TF share=%.2f'%(d['value'],d['ms_per_step'],r['achieved'],r['share_of_step_time']))"
