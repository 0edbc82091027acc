#!/bin/bash
# round-2 session N: lazy Adam steps for the SH rows of culled Gaussians -- parity on the GPU (C-ABI level and both hosts),
# then the step with window 32 / 8 / 0 (eager), stage tables and per-kernel times
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests/test_lazy_sh_adam.py tests/test_gpu_parity.py tests/test_densify_reference.py tests/test_train_ops.py -x -q -m gpu 2>&1 | tail -15
run() {
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['roofline']['stages']
print('  ms/step', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'], 'preprocess_fwd', s['preprocess_fwd']['ms'], 'blend_bwd', s['blend_bwd']['ms'], 'preprocess_bwd', s['preprocess_bwd']['ms'], 'frac', d['roofline']['frac'])
"
}
for w in 32 8 0; do echo "window $w"; run --sh-adam-window $w; done
echo "driver command"; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('  value', d['value'], 'ms/step', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'], 'priming', d['priming_steps_before_warmup'])"
cd /tmp; rm -rf /tmp/prof_n; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_n -o r --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --median-steps 0 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_n/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]: print(' ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us')
PY
