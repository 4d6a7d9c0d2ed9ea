for f in "" "--fp32-kernel" "--fp32-kernel --no-mean-ite" "--no-mean-ite"; do timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $f 2>&1 | grep -E "metric|Error" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(d['dtype'], d['config']['mean_ite'], 'value=%.1f samples/s  ms/step=%.1f  mfma=%.1f TF share=%.2f'%(d['value'],d['ms_per_step'],r['achieved'],r['share_of_step_time']))"; done
