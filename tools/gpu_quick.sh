#!/bin/bash
# quick GPU iteration: parity + stage tests, then raster-only bench (+ optional full bench)
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/test_gpu.log 2>&1; tail -6 gpurun_out/test_gpu.log
timeout 400 python bench.py --steps 20 --warmup 5 --raster-only --no-cpu-baseline > gpurun_out/bench_raster.log 2>&1; tail -1 gpurun_out/bench_raster.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('ms_per_step', d['ms_per_step'], 'raster_ms', d['raster_fwd_bwd_ms'], 'mpix/s', d['mpix_per_s'], 'frac', d['roofline']['raster_fwd_bwd_frac'])
for k, v in d['roofline']['stages'].items(): print(f'  {k:16s} {v[\"ms\"]:8.4f} ms  {v[\"GBps\"]:8.1f} GB/s')
"
if [ "$1" = "full" ]; then timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_full.log 2>&1; tail -1 gpurun_out/bench_full.log | cut -c1-400; fi
