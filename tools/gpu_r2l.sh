#!/bin/bash
# round-2 session L: upper bound of a lazy (temporally blocked) Adam step of the culled SH rows: the step with the culled rows'
# update simply left out (GSR_DEV_SKIP_CULLED, a timing-only switch that exists in this experiment's build only)
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
run() {
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['roofline']['stages']
print('  ms/step', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'], 'blend_bwd', s['blend_bwd']['ms'], 'preprocess_bwd', s['preprocess_bwd']['ms'])
"
}
echo "default (side stream)"; run
echo "culled rows skipped"; GSR_DEV_SKIP_CULLED=1 run
echo "no side stream"; GSR_SH_ADAM_SIDE_STREAM=0 run
echo "loss kernels (gpu tests + per-kernel times)"
timeout 600 python -m pytest tests/test_train_ops.py tests/test_train_step.py -x -q -m gpu 2>&1 | tail -2
cd /tmp; rm -rf /tmp/prof_l; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_l -o r --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --median-steps 0 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_l/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('loss_', 'blend_', 'adam', 'sh_bwd')): print(' ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us')
PY
