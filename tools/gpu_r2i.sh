#!/bin/bash
# round-2 session I: step-time series around densification; side-stream grid sweep (small grids)
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 300 python bench.py --steps 10 --warmup 5 --densify-interval 40 --median-steps 130 --dump-steps --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['protocol']['step_ms']
print('median', d['protocol']['median_ms_per_step'])
for i in range(0, len(s), 10): print(i, ' '.join(f'{x:6.3f}' for x in s[i:i+10]))
"
run() {
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['roofline']['stages']
print('  median', d['protocol']['median_ms_per_step'], 'blend_bwd', s['blend_bwd']['ms'], 'preprocess_bwd', s['preprocess_bwd']['ms'], 'raster-only', d.get('rasterizer_only', {}).get('fwd_bwd_ms'))
"
}
for b in 64 128 192 256 384; do echo "side blocks $b"; GSR_SH_ADAM_SIDE_BLOCKS=$b run; done
