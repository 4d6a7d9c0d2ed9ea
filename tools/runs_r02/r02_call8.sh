#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_8
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
# (1) RCCL path with one rank (real nccl init, all_gather, barrier, all_reduce)
GPSLC_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --samples-per-step 256 --no-cpu-baseline --no-units > $OUT/nccl1.log 2>&1; echo "nccl1 rc=$?"; grep '"metric"' $OUT/nccl1.log | cut -c1-300
# (2) two ranks on the one GPU, gloo (functional rehearsal of the N > 1 code path)
GPSLC_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 1 --warmup 1 --samples-per-step 128 --no-cpu-baseline > $OUT/reh2.log 2>&1; echo "rehearsal rc=$?"; grep '"metric"' $OUT/reh2.log | cut -c1-400
# (3) --gpus 2 on a one-GPU box must refuse loudly
timeout -k 10 120 python bench.py --gpus 2 --steps 1 > $OUT/gpus2.log 2>&1; echo "gpus2 rc=$? (expected 2)"; tail -2 $OUT/gpus2.log
# (4) wrong world size under torchrun must refuse
timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 1 > $OUT/mismatch.log 2>&1; echo "mismatch rc=$? (expected non-zero)"; grep "bench.py:" $OUT/mismatch.log | head -2
# (5) full GPU test-suite
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
