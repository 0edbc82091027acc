#!/bin/bash
# Full GPU session: build, all gpu tests, smoke, benches, rocprof stats.  $1 = profile set tag (e.g. r01_b)
TAG=${1:-r01_x}
mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/test_gpu.log 2>&1; tail -3 gpurun_out/test_gpu.log
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 500 python bench.py > gpurun_out/$TAG/bench_full_C3.json 2>gpurun_out/bench_err.log; cut -c1-300 gpurun_out/$TAG/bench_full_C3.json
timeout 400 python bench.py --raster-only --no-cpu-baseline > gpurun_out/$TAG/bench_raster_only_C3.json 2>>gpurun_out/bench_err.log
timeout 400 python bench.py --config C2 --no-cpu-baseline > gpurun_out/$TAG/bench_full_C2.json 2>>gpurun_out/bench_err.log; cut -c1-200 gpurun_out/$TAG/bench_full_C2.json
rm -rf /tmp/kp; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o kp -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/kp.log 2>&1)
cp $(find /tmp/kp -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/kernel_stats_bench_full_C3.csv
head -12 gpurun_out/$TAG/kernel_stats_bench_full_C3.csv | cut -c1-150
