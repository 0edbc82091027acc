#!/bin/bash
# One GPU-box session: tests, bench, rocprof.  Everything is logged under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing|gfx|Compute Unit" | head -6 > gpurun_out/gpu_info.txt
nproc >> gpurun_out/gpu_info.txt
echo "== build" ; python __graft_entry__.py > gpurun_out/build.log 2>&1; tail -2 gpurun_out/build.log
echo "== stage tests"
timeout 600 python -m pytest tests/test_gpu_stages.py -m gpu -x -q > gpurun_out/test_stages.log 2>&1; tail -15 gpurun_out/test_stages.log
echo "== parity tests"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s > gpurun_out/test_parity.log 2>&1; tail -40 gpurun_out/test_parity.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
echo "== bench raster-only C3"
timeout 400 python bench.py --steps 10 --warmup 3 --raster-only --no-cpu-baseline > gpurun_out/bench_raster.log 2>&1; tail -3 gpurun_out/bench_raster.log
echo "== bench full C3"
timeout 500 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_full.log 2>&1; tail -3 gpurun_out/bench_full.log
echo "== rocprof"
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof_r1 | head -20; tail -3 gpurun_out/rocprof.log
