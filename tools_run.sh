mkdir -p gpurun_out
cd /root/repo
GPSLC_BENCH_FORCE_DIST=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_nccl1.log 2>&1; tail -3 gpurun_out/bench_nccl1.log | cut -c1-600
