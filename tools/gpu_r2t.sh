#!/bin/bash
# round-2 session T: after the removal of five tiny launches per step -- gpu suite, step time (3 repeats), kernel list
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -2
run() {
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('  ms/step', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'], 'it/s', d['value'])
"
}
for rep in 1 2 3; do run; done
cd /tmp; rm -rf /tmp/prof_t; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o r --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --median-steps 0 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_t/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows[:40]:
    c = int(r['Calls'])
    if c >= 60: print(' ', r['Name'][:70], c, round(float(r['AverageNs'])/1e3, 1), 'us', 'per step', round(c / 77.0, 2))
PY
