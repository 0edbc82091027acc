#!/bin/bash
# round-2 session D: CPU baseline diagnostics on the box's host cores; MFMA vs butterfly blend_bwd (per-kernel times); microbench v2
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
nproc; lscpu | grep -E "Model name|Socket|Thread|Core" | head -5
( time timeout 300 python bench.py --cpu-baseline-only --quick-cpu-baseline --config C1 ) > gpurun_out/cpu_baseline_c1.log 2>&1; tail -6 gpurun_out/cpu_baseline_c1.log | cut -c1-900
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -m gpu -q -x > gpurun_out/test_gpu_mfma.log 2>&1; tail -4 gpurun_out/test_gpu_mfma.log | cut -c1-300
bash tools/gpu_kprof.sh "" "-DGSR_BWD_BUTTERFLY" 2>&1 | grep -E "FLAGS|blend|preprocess_bwd"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate && timeout 200 /tmp/valu_rate > gpurun_out/valu_rate2.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/valu_rate2.json"))
for k, v in d["ops"].items():
    print(f"{k:46s}", " ".join(f"{w}:{x['event_cycles']:7.3f}" for w, x in v.items()))
PY
