#!/bin/bash
# shader clock / power while the full train step (and, for comparison, the rasterizer-only loop) is running
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1
for mode in "" "--raster-only"; do
  echo "=== bench.py $mode"
  python bench.py --steps 12000 --warmup 5 --no-cpu-baseline $mode > /tmp/clk.log 2>&1 &
  BPID=$!
  sleep 16
  for i in 1 2 3 4; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power \(W\)|Socket Power|Average" | tr '\n' ' '; echo
    sleep 0.7
  done
  wait $BPID
  tail -1 /tmp/clk.log | cut -c1-200
done
