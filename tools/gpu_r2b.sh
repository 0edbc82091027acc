#!/bin/bash
# round-2 session B: densification on the HIP stream-compaction kernels -- reference parity on the GPU, cost per call, bench with densification in the loop
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests/test_densify_reference.py tests/test_cpp_host.py tests/test_train_step.py -m gpu -q -x > gpurun_out/test_densify_gpu.log 2>&1; tail -8 gpurun_out/test_densify_gpu.log | cut -c1-300
timeout 300 python tools/densify_probe.py C3 > gpurun_out/densify_probe.log 2>&1; grep -E "densifyAndPrune #|gsr::|Self CUDA time" gpurun_out/densify_probe.log | cut -c1-200 | head -20
timeout 400 python bench.py --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/bench_nodensify_300.log 2>&1; tail -1 gpurun_out/bench_nodensify_300.log | cut -c1-200
timeout 400 python bench.py --steps 300 --warmup 20 --densify-interval 100 --no-cpu-baseline > gpurun_out/bench_densify_300.log 2>&1; tail -1 gpurun_out/bench_densify_300.log | cut -c1-200
