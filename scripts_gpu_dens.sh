#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
python __graft_entry__.py > gpurun_out/build.log 2>&1
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host py 2>/dev/null | cut -c1-330
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --densify-interval 10 2>&1 | tail -1 | cut -c1-700
