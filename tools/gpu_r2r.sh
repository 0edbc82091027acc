#!/bin/bash
# round-2 session R: the whole gpu suite + smoke(), then the profile set (tag from $1)
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as e; e.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/gpu_r2_profiles.sh ${1:-c}
