#!/bin/bash
# round-2 profile set: $1 = tag (profiles/r02_<tag>_*).  Kernel stats of the driver's bench command, TCC traffic, SQ counters,
# static issue cost of the blend loops, benches of the other configurations.
TAG=${1:-x}; OUT=gpurun_out/r02_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_cmd_C3.log 2>&1; grep "^{\"metric\"" $OUT/bench_driver_cmd_C3.log > $OUT/bench_driver_cmd_C3.json; grep real $OUT/bench_driver_cmd_C3.log; cut -c1-400 $OUT/bench_driver_cmd_C3.json
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_full_C3.json 2>/dev/null
timeout 300 python bench.py --sh-adam-window 0 --no-cpu-baseline > $OUT/bench_full_C3_eager_sh_adam.json 2>/dev/null
timeout 300 python bench.py --raster-only --no-cpu-baseline > $OUT/bench_raster_only_C3.json 2>/dev/null
for c in C2 C4 C5; do timeout 300 python bench.py --config $c --no-cpu-baseline > $OUT/bench_full_$c.json 2>/dev/null; python -c "
import json,sys; d=json.load(open('$OUT/bench_full_$c.json')); print('$c', d['value'], 'it/s', d['protocol']['median_ms_per_step'], 'ms median')"; done
timeout 300 python bench.py --densify-interval 100 --steps 300 --no-cpu-baseline > $OUT/bench_densify100_300steps_C3.json 2>/dev/null
rm -rf /tmp/kp; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o kp -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/kp.log 2>&1)
cp $(find /tmp/kp -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_bench_full_C3.csv
head -14 $OUT/kernel_stats_bench_full_C3.csv | cut -c1-150
bash tools/gpu_pmc.sh > $OUT/pmc.log 2>&1; cp gpurun_out/pmc_traffic.json $OUT/pmc_traffic_C3_raster_only.json; tail -12 $OUT/pmc.log | cut -c1-200
bash tools/gpu_sq.sh > $OUT/sq.log 2>&1; cp gpurun_out/sq_counters.json $OUT/sq_counters_C3_raster_only.json; tail -6 $OUT/sq.log | cut -c1-300
python tools/isa_cost.py > $OUT/blend_issue_cost.json 2>/dev/null
