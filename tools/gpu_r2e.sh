#!/bin/bash
# round-2 session E: kernel trace of the full (fused, stationary) step; CPU baseline at C3; whole gpu suite
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
rm -rf /tmp/kp; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o kp -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --median-steps 0 > /tmp/kp.log 2>&1)
cp $(find /tmp/kp -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_full_step.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/kernel_stats_full_step.csv")))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f'{r["Name"][:64]:64s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:8.1f} us  {float(r["Percentage"]):5.2f} %')
PY
( time timeout 600 python bench.py --cpu-baseline-only --quick-cpu-baseline --config C3 ) > gpurun_out/cpu_baseline_c3.log 2>&1; tail -5 gpurun_out/cpu_baseline_c3.log | cut -c1-1500
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/test_gpu.log 2>&1; tail -4 gpurun_out/test_gpu.log | cut -c1-300
