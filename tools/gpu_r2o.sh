#!/bin/bash
# round-2 session O: where to run the lazy rows' slice kernel -- next to the HBM-bound per-Gaussian backward kernels (default) or
# next to the VALU-bound backward blend (GSR_LAZY_FORK_EARLY=1); windows 32 / 16
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
run() {
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['roofline']['stages']
print('  ms/step', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'], 'preprocess_fwd', s['preprocess_fwd']['ms'], 'blend_bwd', s['blend_bwd']['ms'], 'preprocess_bwd', s['preprocess_bwd']['ms'])
"
}
echo "slice behind the blend, window 32"; run
echo "slice next to the blend, window 32"; GSR_LAZY_FORK_EARLY=1 run
echo "slice behind the blend, window 16"; run --sh-adam-window 16
echo "slice on the caller's stream, window 32"; GSR_SH_ADAM_SIDE_STREAM=0 run
echo "slice behind the blend, window 32 (again)"; run
