#!/bin/bash
# round-2 session F: parity + stage times after a kernel change (raster-only bench stage table + per-kernel stats)
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -m gpu -q -x > gpurun_out/test_gpu_f.log 2>&1; tail -3 gpurun_out/test_gpu_f.log | cut -c1-300
timeout 300 python bench.py --steps 50 --warmup 10 --raster-only --no-cpu-baseline --median-steps 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('raster_ms', d['raster_fwd_bwd_ms'], ' '.join(f'{k}={v[\"ms\"]:.4f}' for k, v in d['roofline']['stages'].items()))
"
bash tools/gpu_kprof.sh "" 2>&1 | head -24
