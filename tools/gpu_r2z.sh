#!/bin/bash
# round-2 session Z: one random gather (the rectangle, in the offset scan) instead of two (tile count there, rectangle in the
# instance emission): parity, stage times
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -x -q -m gpu 2>&1 | tail -2
run() {
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['roofline']['stages']
print('  ms/step', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'], 'offset_scan', s['offset_scan']['ms'], 'emit', s['emit_instances']['ms'])
"
}
for rep in 1 2 3; do run; done
