#!/bin/bash
# round-2 session C: the reworked bench.py (stationary protocol, per-step medians, CPU train-step baseline), gpu tests of the hosts
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_driver_cmd.log 2>&1; tail -5 gpurun_out/bench_driver_cmd.log | cut -c1-3000
timeout 400 python bench.py --steps 100 --warmup 20 --densify-interval 100 --no-cpu-baseline > gpurun_out/bench_densify.log 2>&1; tail -1 gpurun_out/bench_densify.log | cut -c1-1200
timeout 400 python bench.py --config C2 --no-cpu-baseline > gpurun_out/bench_C2.log 2>&1; tail -1 gpurun_out/bench_C2.log | cut -c1-400
GSR_BENCH_FORCE_DP=1 timeout 400 python bench.py --no-cpu-baseline > gpurun_out/bench_dp1.log 2>&1; tail -1 gpurun_out/bench_dp1.log | cut -c1-600
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/test_gpu.log 2>&1; tail -4 gpurun_out/test_gpu.log | cut -c1-300
