#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1
rm -rf /tmp/kp; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o kp -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/kp.log 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/kp/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step (ms):", tot/13/1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} us  per-step {float(r["TotalDurationNs"])/13/1e3:8.1f} us')
PY
cp /tmp/kp/*/*kernel_stats.csv gpurun_out/full_kernel_stats.csv 2>/dev/null || cp $(find /tmp/kp -name "*kernel_stats.csv" | head -1) gpurun_out/full_kernel_stats.csv
