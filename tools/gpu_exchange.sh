#!/bin/bash
# view-factored exchange on the 1-GPU box: parity tests, kernel costs, bench.py's data-parallel path with one rank over
# RCCL (both exchanges), and 2 ranks sharing the GPU over gloo
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 600 python -m pytest tests -m gpu -x -q -k "view_factored or cpp_host" > gpurun_out/test_exchange.log 2>&1; tail -4 gpurun_out/test_exchange.log
timeout 300 python tools/exchange_probe.py > gpurun_out/exchange_probe.log 2>&1; tail -1 gpurun_out/exchange_probe.log
for ex in factored allreduce; do
  GSR_BENCH_FORCE_DP=1 GSR_BENCH_EXCHANGE=$ex timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_dp1_$ex.log 2>&1
  tail -1 gpurun_out/bench_dp1_$ex.log | cut -c1-160
done
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_plain.log 2>&1; tail -1 gpurun_out/bench_plain.log | cut -c1-160
GSR_BENCH_SHARE_GPU=1 GSR_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --config C2 --no-cpu-baseline \
  > gpurun_out/dist_smoke.log 2>&1
tail -1 gpurun_out/dist_smoke.log | cut -c1-400
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --densify-interval 10 > gpurun_out/bench_densify_cpp.log 2>&1; tail -1 gpurun_out/bench_densify_cpp.log | cut -c1-200
