#!/bin/bash
# round-2 session V: preprocess_fwd with the lazy rows' step count loaded up front: stage times, 3 repeats; link test
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 600 python -m pytest tests/test_reference_link.py tests/test_lazy_sh_adam.py -x -q -m gpu 2>&1 | tail -2
run() {
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['roofline']['stages']
print('  ms/step', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'], 'preprocess_fwd', s['preprocess_fwd']['ms'], 'unfused leg', d['rasterizer_only']['stages_ms']['preprocess_fwd'])
"
}
for rep in 1 2 3; do run; done
