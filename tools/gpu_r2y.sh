#!/bin/bash
# round-2 session Y: the same C3 cloud re-indexed along a Z-order curve (the index coherence of a real SLAM map) against the
# generated (random) index order: step and stage times
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
run() {
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric"' | tee gpurun_out/bench_$1_$2.json | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['roofline']['stages']
print('  ms/step', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'], 'V', d['config']['visible'], 'R', d['config']['instances'])
print('   ', ' '.join(f'{k} {v[\"ms\"]}' for k, v in s.items()))
"
}
echo "random index order"; run --scene-order random
echo "morton index order"; run --scene-order morton
