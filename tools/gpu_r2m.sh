#!/bin/bash
# round-2 session M: loss kernels built without the SLP vectoriser (per-kernel times), and the same switch on every translation unit
mkdir -p gpurun_out; export TMPDIR=/tmp
kern() {
  cd /tmp; rm -rf /tmp/prof_m; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_m -o r --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --median-steps 0 --no-cpu-baseline > /dev/null 2>&1
  python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_m/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('loss_', 'blend_', 'adam_kernel', 'sh_bwd', 'preprocess', 'emit', 'radix_scatter')): print(' ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us')
PY
  cd $GRAFT_REPO_ROOT
}
step() {
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('  ms/step', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'])
"
}
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
echo "product build"; timeout 600 python -m pytest tests/test_train_ops.py -x -q -m gpu 2>&1 | tail -1; step; kern
echo "every TU with -fno-slp-vectorize"
GSR_EXTRA_FLAGS=-fno-slp-vectorize python __graft_entry__.py > gpurun_out/build2.log 2>&1 || { tail -20 gpurun_out/build2.log; exit 1; }
GSR_EXTRA_FLAGS=-fno-slp-vectorize step; GSR_EXTRA_FLAGS=-fno-slp-vectorize kern
