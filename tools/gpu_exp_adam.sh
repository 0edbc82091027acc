#!/bin/bash
# fused SH Adam: probe for each flag set
mkdir -p gpurun_out; export TMPDIR=/tmp
for flags in "$@"; do
  echo "=== FLAGS: $flags"
  GSR_EXTRA_FLAGS="$flags" python photo-slam_amd/build.py > gpurun_out/build.log 2>&1 || { tail -5 gpurun_out/build.log; continue; }
  GSR_EXTRA_FLAGS="$flags" timeout 300 python tools/fused_adam_probe.py 2>&1 | tail -1
done
GSR_EXTRA_FLAGS="" python photo-slam_amd/build.py > /dev/null 2>&1
