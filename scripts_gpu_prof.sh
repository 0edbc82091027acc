#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof_r1 | head; tail -2 gpurun_out/rocprof.log
head -40 gpurun_out/prof_r1/*kernel_stats.csv
