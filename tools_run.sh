mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q -x > gpurun_out/pytest16.log 2>&1; tail -8 gpurun_out/pytest16.log
for L in 1 8 64; do timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --levels $L 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('L=%d value=%.1f samples/s  ms/step=%.1f  mfma=%.1f TF share=%.2f'%(d['config']['levels'],d['value'],d['ms_per_step'],r['achieved'],r['share_of_step_time']))"; done
