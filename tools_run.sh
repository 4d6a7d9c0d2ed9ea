mkdir -p gpurun_out
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_estimation.py tests/test_gpu_fuzz.py tests/test_gpu_kernel.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/pytest24.log 2>&1; tail -4 gpurun_out/pytest24.log
cd /tmp && export TMPDIR=/tmp
for m in 1 64; do
rm -rf /root/repo/gpurun_out/prof_$m
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_$m -o ks -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --levels $m > /root/repo/gpurun_out/bench_$m.log 2>&1 || exit 1
grep metric /root/repo/gpurun_out/bench_$m.log | cut -c1-120
python3 /root/repo/tools/kernel_stats_md.py /root/repo/gpurun_out/prof_$m "L=$m" | grep "ite_mean\|gram_kernel\|gemm_nt_kernel<1"
done
