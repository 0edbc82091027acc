#!/bin/bash
# round-2 session H: tuning of the side-stream culled-rows Adam (grid size), full-step medians
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
run() {
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['roofline']['stages']
print('  median', d['protocol']['median_ms_per_step'], 'p10', d['protocol']['p10_ms'], 'blend_bwd', s['blend_bwd']['ms'], 'preprocess_bwd', s['preprocess_bwd']['ms'])
"
}
echo "side stream off"; GSR_SH_ADAM_SIDE_STREAM=0 run
for b in 256 512 1024 2048 4096 0; do echo "side blocks $b"; GSR_SH_ADAM_SIDE_BLOCKS=$b run; done
