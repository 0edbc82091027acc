#!/bin/bash
# round-2 session J: why do steps get slower after each densification?  block medians over 10 densifications, both hosts, stage tables
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for host in cpp py; do
echo "== host $host"
timeout 300 python bench.py --host $host --steps 10 --warmup 5 --densify-interval 40 --median-steps 420 --dump-steps --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json, statistics
d = json.loads(sys.stdin.readline())
s = d['protocol']['step_ms']
print('block medians (40 steps each):', ' '.join(f'{statistics.median(s[i:i+40]):.3f}' for i in range(0, len(s), 40)))
print('gaussians_after', d['config']['gaussians_after'])
print(' '.join(f'{k}={v[\"ms\"]:.4f}' for k, v in d['roofline']['stages'].items()))
"
done
echo "== no densify, cpp"
timeout 300 python bench.py --steps 10 --warmup 5 --median-steps 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('median', d['protocol']['median_ms_per_step'])
print(' '.join(f'{k}={v[\"ms\"]:.4f}' for k, v in d['roofline']['stages'].items()))
"
