#!/bin/bash
# functional check of bench.py's multi-rank path on a 1-GPU box: 2 ranks share device 0 and talk over gloo
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
GSR_BENCH_SHARE_GPU=1 GSR_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --config C2 --no-cpu-baseline \
  > gpurun_out/dist_smoke.log 2>&1
tail -3 gpurun_out/dist_smoke.log | cut -c1-600
