#!/bin/bash
# round-2 session A: VALU issue-rate microbenchmark, the whole gpu suite (incl. the full-size C3/C4/C5 oracle comparisons), raster bench
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate && timeout 120 /tmp/valu_rate > gpurun_out/valu_rate.json; cat gpurun_out/valu_rate.json | cut -c1-260
timeout 1500 python -m pytest tests -m gpu -q -s --durations=8 > gpurun_out/test_gpu.log 2>&1; grep -vE "amdgpu.ids" gpurun_out/test_gpu.log | tail -25 | cut -c1-400
timeout 400 python bench.py --steps 20 --warmup 5 --raster-only --no-cpu-baseline > gpurun_out/bench_raster.log 2>&1; tail -1 gpurun_out/bench_raster.log | cut -c1-600
