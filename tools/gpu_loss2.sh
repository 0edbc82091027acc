#!/bin/bash
# loss kernel times inside the fused step (kernel trace statistics), gpu test of the loss
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 600 python -m pytest tests/test_train_ops.py -x -q -m gpu 2>&1 | tail -1
cd /tmp; rm -rf /tmp/prof_l2; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_l2 -o r --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --median-steps 0 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_l2/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'loss_' in r['Name']: print(' ', r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us')
PY
