#!/bin/bash
# experiment runner: for each flag set in "$@" rebuild and run the raster-only bench
mkdir -p gpurun_out; export TMPDIR=/tmp
for flags in "$@"; do
  echo "=== FLAGS: $flags"
  GSR_EXTRA_FLAGS="$flags" python photo-slam_amd/build.py > gpurun_out/build_exp.log 2>&1 || { tail -20 gpurun_out/build_exp.log; continue; }
  timeout 300 python bench.py --steps 20 --warmup 5 --raster-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('raster_ms', d['raster_fwd_bwd_ms'], ' '.join(f'{k}={v[\"ms\"]:.3f}' for k, v in d['roofline']['stages'].items()))
"
done
