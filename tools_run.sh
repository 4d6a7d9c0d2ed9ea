mkdir -p gpurun_out
run() { echo "== samples=$4 streams=$1 panel=$2 batch=$3" >> gpurun_out/sweep8.log
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --samples-per-step $4 --no-cpu-baseline --streams $1 --panel $2 --max-batch $3 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print('value=%.1f samples/s  ms/step=%.1f  mfma=%.1f TF  avg_launch_ms=%.3f share=%.2f'%(d['value'],d['ms_per_step'],r['achieved'],r['avg_launch_ms'],r['share_of_step_time']))" >> gpurun_out/sweep8.log; }
run 1 8 256 512
run 2 8 256 512
run 2 8 128 512
run 1 8 512 512
run 2 4 128 512
run 3 8 128 768
cat gpurun_out/sweep8.log
