#!/bin/bash
# the round's last GPU session: whole gpu suite, smoke(), the driver's bench command (with the CPU baseline), kernel statistics of
# the same command -> gpurun_out/r02_$1/
TAG=${1:-h}; OUT=gpurun_out/r02_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" 2>&1 | tail -1
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_cmd_C3.log 2>&1; grep "^{\"metric\"" $OUT/bench_driver_cmd_C3.log > $OUT/bench_driver_cmd_C3.json; grep real $OUT/bench_driver_cmd_C3.log; cut -c1-330 $OUT/bench_driver_cmd_C3.json
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_full_C3.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_full_C3.json')); print('100 steps', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'], 'views', d['changing_views_run']['ms_per_step'])"
rm -rf /tmp/kp; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o kp -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/kp.log 2>&1)
cp $(find /tmp/kp -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_bench_full_C3.csv
head -12 $OUT/kernel_stats_bench_full_C3.csv | cut -c1-120
